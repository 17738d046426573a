"""GraphWaveNet backbone with the reference's module surface (parameter holder + native calls).

Mirrors ``step/step_arch/graphwavenet/model.py:50-224``: same constructor keywords and the same
``state_dict`` (``filter_convs.{i}``, ``gate_convs.{i}``, ``residual_convs.{i}`` -- present but
never used when ``gcn_bool`` --, ``skip_convs.{i}``, ``bn.{i}``, ``gconv.{i}.mlp.mlp``,
``fc_his.{0,2}``, ``start_conv``, ``end_conv_1/2``, ``nodevec1/2``).  Arithmetic lives in
``libstep_hip`` (``step_gwnet_forward`` / ``step_gwnet_backward``), driven by ``step.py``.
"""
import ctypes

import torch
from torch import nn

from .. import _lib


class _Linear1x1(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.mlp = nn.Conv2d(c_in, c_out, kernel_size=(1, 1), padding=(0, 0), stride=(1, 1), bias=True)


class _Gcn(nn.Module):
    def __init__(self, c_in, c_out, support_len=3, order=2):
        super().__init__()
        self.mlp = _Linear1x1((order * support_len + 1) * c_in, c_out)


class GraphWaveNet(nn.Module):
    def __init__(self, num_nodes, support_len, dropout=0.3, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=2,
                 out_dim=12, residual_channels=32, dilation_channels=32, skip_channels=256, end_channels=512,
                 kernel_size=2, blocks=4, layers=2, **kwargs):
        super().__init__()
        fixed = dict(gcn_bool=True, addaptadj=True, aptinit=None, in_dim=2, out_dim=12, residual_channels=32,
                     dilation_channels=32, skip_channels=256, end_channels=512, kernel_size=2, blocks=4, layers=2,
                     support_len=2)
        given = dict(gcn_bool=gcn_bool, addaptadj=addaptadj, aptinit=aptinit, in_dim=in_dim, out_dim=out_dim,
                     residual_channels=residual_channels, dilation_channels=dilation_channels,
                     skip_channels=skip_channels, end_channels=end_channels, kernel_size=kernel_size, blocks=blocks,
                     layers=layers, support_len=support_len)
        if given != fixed:
            raise NotImplementedError(f"the HIP backbone is specialised to the STEP configs {fixed}; got {given}")
        self.num_nodes = num_nodes
        self.dropout = float(dropout)
        self.blocks, self.layers = blocks, layers
        self.gcn_bool, self.addaptadj = gcn_bool, addaptadj
        self.filter_convs = nn.ModuleList()
        self.gate_convs = nn.ModuleList()
        self.residual_convs = nn.ModuleList()
        self.skip_convs = nn.ModuleList()
        self.bn = nn.ModuleList()
        self.gconv = nn.ModuleList()
        self.fc_his = nn.Sequential(nn.Linear(96, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU())
        self.start_conv = nn.Conv2d(in_dim, residual_channels, kernel_size=(1, 1))
        self.nodevec1 = nn.Parameter(torch.randn(num_nodes, 10))
        self.nodevec2 = nn.Parameter(torch.randn(10, num_nodes))
        self.supports_len = support_len + 1
        for _b in range(blocks):
            dil = 1
            for _l in range(layers):
                self.filter_convs.append(nn.Conv2d(residual_channels, dilation_channels, kernel_size=(1, kernel_size), dilation=dil))
                self.gate_convs.append(nn.Conv2d(residual_channels, dilation_channels, kernel_size=(1, kernel_size), dilation=dil))
                self.residual_convs.append(nn.Conv2d(dilation_channels, residual_channels, kernel_size=(1, 1)))
                self.skip_convs.append(nn.Conv2d(dilation_channels, skip_channels, kernel_size=(1, 1)))
                self.bn.append(nn.BatchNorm2d(residual_channels))
                self.gconv.append(_Gcn(dilation_channels, residual_channels, support_len=self.supports_len))
                dil *= 2
        self.end_conv_1 = nn.Conv2d(skip_channels, end_channels, kernel_size=(1, 1), bias=True)
        self.end_conv_2 = nn.Conv2d(end_channels, out_dim, kernel_size=(1, 1), bias=True)
        self.receptive_field = 13

    # names of the tensors the native forward reads, in C-struct order
    def _apply(self, fn, recurse=True):
        self._nt_cache = None                # .to() / .float() replace the BatchNorm buffers by new tensors
        return super()._apply(fn, recurse)

    def native_tensors(self):
        """name -> tensor, in C-struct order.  The dict is built once (it is a walk over ~110 module attributes, 0.15 ms of host
        time per call and two calls per step) and dropped whenever the module is converted (``_apply``); the pointers are read from
        the tensors at every call of ``fill_gwnet_struct``, so in-place updates, ``load_state_dict`` and re-homing into the flat
        parameter buffer need no invalidation."""
        c = getattr(self, "_nt_cache", None)
        if c is not None:
            return c
        t = {"nodevec1": self.nodevec1, "nodevec2": self.nodevec2,
             "start_w": self.start_conv.weight, "start_b": self.start_conv.bias,
             "fc_his0_w": self.fc_his[0].weight, "fc_his0_b": self.fc_his[0].bias,
             "fc_his2_w": self.fc_his[2].weight, "fc_his2_b": self.fc_his[2].bias,
             "end1_w": self.end_conv_1.weight, "end1_b": self.end_conv_1.bias,
             "end2_w": self.end_conv_2.weight, "end2_b": self.end_conv_2.bias}
        for i in range(8):
            t[f"filter_w.{i}"] = self.filter_convs[i].weight
            t[f"filter_b.{i}"] = self.filter_convs[i].bias
            t[f"gate_w.{i}"] = self.gate_convs[i].weight
            t[f"gate_b.{i}"] = self.gate_convs[i].bias
            t[f"skip_w.{i}"] = self.skip_convs[i].weight
            t[f"skip_b.{i}"] = self.skip_convs[i].bias
            t[f"bn_w.{i}"] = self.bn[i].weight
            t[f"bn_b.{i}"] = self.bn[i].bias
            t[f"bn_rm.{i}"] = self.bn[i].running_mean
            t[f"bn_rv.{i}"] = self.bn[i].running_var
            t[f"gconv_w.{i}"] = self.gconv[i].mlp.mlp.weight
            t[f"gconv_b.{i}"] = self.gconv[i].mlp.mlp.bias
        self._nt_cache = t
        return t

    # parameters that receive a gradient (the reference leaves the others at grad=None:
    # residual_convs.*, gconv.7, bn.7 -- SURVEY.md section 5)
    def trainable_native(self):
        c = getattr(self, "_tn_cache", None)
        if c is not None and c[0] is getattr(self, "_nt_cache", None):
            return c[1]
        t = self.native_tensors()
        keep = {}
        for k, v in t.items():
            if "_rm" in k or "_rv" in k:
                continue
            if k.split(".")[0] in ("gconv_w", "gconv_b", "bn_w", "bn_b") and k.endswith(".7"):
                continue
            keep[k] = v
        self._tn_cache = (t, keep)
        return keep

    def forward(self, input, hidden_states, sampled_adj):
        """Standalone call with the reference's signature (graphwavenet/model.py:132-224): input [B, 12, N, C >= 2] short history,
        hidden_states [B, N, 96] = TSFormer state of the last patch, sampled_adj [B, N, N] -> prediction [B, N, 12].  One native
        launch sequence (``step_gwnet_forward``; train mode: batch-statistic BatchNorm with running-stat update and the gcn dropout,
        like the reference module in train mode).  Forward only: gradients flow through ``step_amd.STEP`` (whose autograd function
        drives the same kernels plus ``step_gwnet_backward``), so this call must run under ``torch.no_grad()``."""
        if not input.is_cuda:
            raise RuntimeError("step_amd.GraphWaveNet runs only on an AMD GPU: libstep_hip has no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("step_amd.GraphWaveNet.forward is forward-only (wrap the call in torch.no_grad()); training goes through step_amd.STEP")
        L = _lib
        B, T, N, Cin = input.shape
        assert T == 12 and N == self.num_nodes and tuple(hidden_states.shape) == (B, N, 96) and tuple(sampled_adj.shape) == (B, N, N)
        dev = input.device
        drop = self.dropout if self.training else 0.0
        bf = int(getattr(self, "matmul_precision", "f32") == "bf16")
        struct = fill_gwnet_struct(self.native_tensors(), bf)
        saved = torch.empty(int(L.lib().step_gwnet_saved_floats(B, N, int(drop > 0))), device=dev)
        work = torch.empty(int(L.lib().step_gwnet_work_floats(B, N, 0)), device=dev)
        pred = torch.empty(B, 12, N, device=dev)
        self._seed_ctr = getattr(self, "_seed_ctr", 0) + 1
        seed = (torch.initial_seed() * 2862933555777941757 + self._seed_ctr * 3037000493) & ((1 << 63) - 1)
        L.call("step_gwnet_forward", L.ptr(input.contiguous().float()), B, N, Cin, L.ptr(hidden_states.contiguous().float()),
               L.ptr(sampled_adj.contiguous().float()), ctypes.byref(struct), int(self.training), float(drop), seed, 0.1,
               L.ptr(saved), L.ptr(work), L.ptr(pred), L.stream())
        if self.training:
            for m in list(self.bn)[:7]:
                m.num_batches_tracked += 1
        return pred.transpose(1, 2)               # [B, N, 12] (model.py:222-224)


def fill_gwnet_struct(tensors, gemm_bf16=False):
    """tensors: dict name -> device tensor (names as in GraphWaveNet.native_tensors)."""
    s = _lib.StepGwnetParams()
    s.gemm_bf16 = int(gemm_bf16)
    for k, v in tensors.items():
        if v is None:
            continue
        assert v.is_cuda and v.is_contiguous() and v.dtype == torch.float32, k
        if "." in k:
            name, idx = k.split(".")
            getattr(s, name)[int(idx)] = v.data_ptr()
        else:
            setattr(s, k, v.data_ptr())
    return s
