"""Loss hook with the reference's signature (``step/step_loss/step_loss.py:5-16`` +
``basicts/metrics/mae.py:5-28``): ``step_loss(prediction, real_value, theta, priori_adj,
gsl_coefficient, null_val)``.  The reference runner calls it on RESCALED predictions
(``base_tsf_runner.py:240-250``); it is caller-side code, a handful of element-wise torch ops on
``[B,12,N,1]`` / ``[B,N,N]`` tensors whose autograd feeds ``dpred`` / ``dtheta`` into the native backward.
"""
import ctypes

import numpy as np
import torch


def masked_mae(preds, labels, null_val=np.nan):
    if np.isnan(null_val):
        mask = ~torch.isnan(labels)
    else:
        mask = (labels - null_val).abs() > 5e-5          # ~isclose(labels, null_val, atol=5e-5, rtol=0)
    mask = mask.float()
    mask = mask / torch.mean(mask)
    mask = torch.nan_to_num(mask, nan=0.0)
    loss = torch.abs(preds - labels) * mask
    return torch.mean(torch.nan_to_num(loss, nan=0.0))


def _flat_stride(x):
    """element stride s such that the logical elements of x, in C order, sit at x.data_ptr() + 4 * s * i (x contiguous: 1; one
    feature of a contiguous [..., C] tensor, ``y[..., :1]``: C); None when x has no such stride."""
    if x.is_contiguous():
        return 1
    if x.dim() < 2 or x.shape[-1] != 1:
        return None
    s = x.stride(-2)
    exp = s
    for d in range(x.dim() - 2, -1, -1):
        if x.shape[d] != 1 and x.stride(d) != exp:
            return None
        exp *= x.shape[d]
    return s


class _NativeStepLoss(torch.autograd.Function):
    """Loss value + both gradients from two HIP launches (libstep_hip step_loss_scaled_fwd_bwd); the backward multiplies both by the
    incoming gradient in one more."""

    @staticmethod
    def forward(ctx, prediction, real_value, theta, priori_adj, coef, null_val, scale, shift):
        from . import _lib
        p = prediction.contiguous().float()
        r = real_value.float()
        rs = _flat_stride(r)
        if rs is None:
            r, rs = r.contiguous(), 1
        t = theta.contiguous().float()
        a = priori_adj.contiguous().float()
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        dp, dt = torch.empty_like(p), torch.empty_like(t)
        work = torch.empty(3, device=p.device, dtype=torch.float64)
        assert r.is_cuda and r.dtype == torch.float32
        if torch.is_tensor(coef):
            # replayed (graph-captured) step: `coef` is the device StepDynState whose gsl_coef field the kernel reads (step_amd/graphed.py)
            _lib.call("step_loss_scaled_fwd_bwd_dyn", _lib.ptr(p), ctypes.c_void_p(r.data_ptr()), p.numel(), rs, float(scale), float(shift), _lib.ptr(t),
                      _lib.ptr(a), t.numel(), float(null_val), _lib.ptr(work), _lib.ptr(loss), _lib.ptr(dp), _lib.ptr(dt), _lib.ptr(coef), _lib.stream())
        else:
            _lib.call("step_loss_scaled_fwd_bwd", _lib.ptr(p), ctypes.c_void_p(r.data_ptr()), p.numel(), rs, float(scale), float(shift), _lib.ptr(t),
                      _lib.ptr(a), t.numel(), float(null_val), float(coef), _lib.ptr(work), _lib.ptr(loss), _lib.ptr(dp), _lib.ptr(dt), _lib.stream())
        ctx.save_for_backward(dp, dt)
        ctx.shapes = (prediction.shape, theta.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        dp, dt = ctx.saved_tensors
        if g.is_cuda and g.dtype == torch.float32 and g.numel() == 1:
            op, ot = torch.empty_like(dp), torch.empty_like(dt)
            _lib.call("step_scale2", _lib.ptr(dp), dp.numel(), _lib.ptr(dt), dt.numel(), _lib.ptr(g.contiguous()), _lib.ptr(op), _lib.ptr(ot),
                      _lib.stream())
        else:
            op, ot = dp * g, dt * g
        return op.view(ctx.shapes[0]), None, ot.view(ctx.shapes[1]), None, None, None, None, None


def step_loss_native(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val=0.0, rescale=None):
    """Same signature and value as ``step_loss`` (finite ``null_val``), computed by libstep_hip.
    ``rescale=(mean, std)``: ``prediction`` / ``real_value`` are the NORMALISED tensors and the loss is taken on ``x * std + mean`` --
    the runner's inverse scaling (base_tsf_runner.py:240-250, step_runner.py:86-92) done inside the loss kernels instead of four
    element-wise launches before them and their autograd nodes after; ``real_value`` may be one feature of the batch tensor
    (``future[..., :1]``), it is read in place."""
    mean, std = (0.0, 1.0) if rescale is None else rescale
    return _NativeStepLoss.apply(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val, std, mean)


def step_loss(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val=np.nan):
    B, N, _ = theta.shape
    t = theta.reshape(B, N * N)
    y = priori_adj.reshape(B, N * N)
    loss_graph = torch.nn.functional.binary_cross_entropy(t, y)
    loss_pred = masked_mae(prediction, real_value, null_val)
    return loss_pred + loss_graph * gsl_coefficient


_DUMMY = {}


def masked_mae_native(preds, labels, null_val=0.0, rescale=None):
    """``masked_mae(preds * std + mean, labels * std + mean, null_val)`` (basicts/metrics/mae.py:5-28 behind the runner's inverse scaling,
    base_tsf_runner.py:240-250) with value and gradient from the two launches of ``step_loss_native`` -- the TSFormer pre-training loss
    (step/TSFormer_*.py: ``CFG.TRAIN.LOSS = masked_mae``).  Element order does not matter to a mean: pass the tensors in whatever
    contiguous layout they have (the pre-training module's outputs are transposed views of contiguous tensors: ``recon.transpose(1, 2)``)."""
    dev = preds.device
    d = _DUMMY.get(dev)
    if d is None:      # the graph term of step_loss with coefficient 0: one edge with theta = prior = 1/2
        d = _DUMMY[dev] = (torch.full((1, 1, 1), 0.5, device=dev), torch.full((1, 1, 1), 0.5, device=dev))
    return step_loss_native(preds, labels, d[0], d[1], 0.0, null_val=null_val, rescale=rescale)


_METRIC_WORK = {}


def masked_metrics_native(preds, labels, null_val=0.0):
    """-> f32 cuda tensor [3] = (masked_mae, masked_rmse, masked_mape) of ``basicts/metrics/{mae,rmse,mape}.py`` for a finite ``null_val``
    (MAPE's own null value is 0, mape.py:21), from ONE launch of libstep_hip instead of ~30 element-wise ones -- the three numbers the
    reference's runner evaluates every training iteration (base_tsf_runner.py:252-254).  No gradient (they feed the epoch meters)."""
    from . import _lib
    p, r = preds.detach(), labels.detach()
    if p.dtype != torch.float32 or r.dtype != torch.float32 or not p.is_cuda or p.numel() != r.numel():
        raise ValueError("masked_metrics_native: f32 cuda tensors of one size")
    ps, rs = _flat_stride(p), _flat_stride(r)
    if ps is None:
        p, ps = p.contiguous(), 1
    if rs is None:
        r, rs = r.contiguous(), 1
    work = _METRIC_WORK.get(p.device)
    if work is None:
        work = _METRIC_WORK[p.device] = torch.zeros(6, dtype=torch.float64, device=p.device)
    out = torch.empty(3, dtype=torch.float32, device=p.device)
    _lib.call("step_masked_metrics", ctypes.c_void_p(p.data_ptr()), ps, ctypes.c_void_p(r.data_ptr()), rs, p.numel(), float(null_val),
              _lib.ptr(work), _lib.ptr(out), _lib.stream())
    return out
