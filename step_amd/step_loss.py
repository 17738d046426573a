"""Loss hook with the reference's signature (``step/step_loss/step_loss.py:5-16`` +
``basicts/metrics/mae.py:5-28``): ``step_loss(prediction, real_value, theta, priori_adj,
gsl_coefficient, null_val)``.  The reference runner calls it on RESCALED predictions
(``base_tsf_runner.py:240-250``); it is caller-side code, a handful of element-wise torch ops on
``[B,12,N,1]`` / ``[B,N,N]`` tensors whose autograd feeds ``dpred`` / ``dtheta`` into the native backward.
"""
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


class _NativeStepLoss(torch.autograd.Function):
    """Loss value + both gradients from two HIP launches (libstep_hip step_loss_fwd_bwd)."""

    @staticmethod
    def forward(ctx, prediction, real_value, theta, priori_adj, coef, null_val):
        from . import _lib
        p = prediction.contiguous().float()
        r = real_value.contiguous().float()
        t = theta.contiguous().float()
        a = priori_adj.contiguous().float()
        loss = torch.empty((), device=p.device, dtype=torch.float32)
        dp, dt = torch.empty_like(p), torch.empty_like(t)
        work = torch.empty(3, device=p.device, dtype=torch.float64)
        _lib.call("step_loss_fwd_bwd", _lib.ptr(p), _lib.ptr(r), p.numel(), _lib.ptr(t), _lib.ptr(a), t.numel(), float(null_val),
                  float(coef), _lib.ptr(work), _lib.ptr(loss), _lib.ptr(dp), _lib.ptr(dt), _lib.stream())
        ctx.save_for_backward(dp, dt)
        ctx.shapes = (prediction.shape, theta.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        dp, dt = ctx.saved_tensors
        return (dp * g).view(ctx.shapes[0]), None, (dt * g).view(ctx.shapes[1]), None, None, None


def step_loss_native(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val=0.0):
    """Same signature and value as ``step_loss`` (finite ``null_val``), computed by libstep_hip."""
    return _NativeStepLoss.apply(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val)


def step_loss(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val=np.nan):
    B, N, _ = theta.shape
    t = theta.reshape(B, N * N)
    y = priori_adj.reshape(B, N * N)
    loss_graph = torch.nn.functional.binary_cross_entropy(t, y)
    loss_pred = masked_mae(prediction, real_value, null_val)
    return loss_pred + loss_graph * gsl_coefficient
