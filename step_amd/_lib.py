"""ctypes binding of libstep_hip.so (the C ABI declared in include/step_hip.h).

The library is the product path: if it is missing or a call fails, this module raises --
there is no eager/PyTorch fallback anywhere in step_amd.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEP_HIP_LIB") or os.path.join(_HERE, "libstep_hip.so")     # override: A/B builds of the same ABI

_vp, _i, _l, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64


class StepGemm(ctypes.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i), ("batch", _i),
                ("A", _vp), ("sam", _l), ("sak", _l), ("sab", _l), ("a_bf16", _i),
                ("B", _vp), ("sbk", _l), ("sbn", _l), ("sbb", _l), ("b_bf16", _i),
                ("C", _vp), ("ldc", _l), ("scn", _l), ("scb", _l),
                ("alpha", _f), ("accumulate", _i), ("bias", _vp), ("relu", _i), ("splitk", _i),
                ("a_kblk", _i), ("a_kstride", _l), ("b_kblk", _i), ("b_kstride", _l),
                ("b_nblk", _i), ("b_nstride", _l), ("c_nblk", _i), ("c_nstride", _l),
                ("a_kscale", _vp), ("a_kshift", _vp), ("a_kperiod", _i),
                ("batch0", _i), ("sab1", _l), ("sbb1", _l), ("scb1", _l), ("compute_bf16", _i), ("a_rowsum", _vp), ("c_nscale", _vp), ("c_nshift", _vp), ("c_mvec", _vp),
                ("c_nperiod", _i), ("splitk_ws", _vp), ("splitk_ws_floats", _l)]


class StepDglParams(ctypes.Structure):
    NAMES = ["conv1_w", "conv1_b", "conv2_w", "conv2_b", "fc_w", "fc_b",
             "bn1_w", "bn1_b", "bn1_rm", "bn1_rv", "bn2_w", "bn2_b", "bn2_rm", "bn2_rv",
             "bn3_w", "bn3_b", "bn3_rm", "bn3_rv", "fc_out_w", "fc_out_b", "fc_cat_w", "fc_cat_b"]
    _fields_ = [(n, _vp) for n in NAMES] + [("gemm_bf16", _i)]


class StepGwnetParams(ctypes.Structure):
    _fields_ = [("nodevec1", _vp), ("nodevec2", _vp), ("start_w", _vp), ("start_b", _vp),
                ("filter_w", _vp * 8), ("filter_b", _vp * 8), ("gate_w", _vp * 8), ("gate_b", _vp * 8),
                ("skip_w", _vp * 8), ("skip_b", _vp * 8),
                ("bn_w", _vp * 8), ("bn_b", _vp * 8), ("bn_rm", _vp * 8), ("bn_rv", _vp * 8),
                ("gconv_w", _vp * 8), ("gconv_b", _vp * 8),
                ("fc_his0_w", _vp), ("fc_his0_b", _vp), ("fc_his2_w", _vp), ("fc_his2_b", _vp),
                ("end1_w", _vp), ("end1_b", _vp), ("end2_w", _vp), ("end2_b", _vp), ("gemm_bf16", _i)]


_PD = ctypes.POINTER(StepDglParams)


class StepDynState(ctypes.Structure):
    """host mirror of the device-resident per-step scalars of a replayed training step (include/step_hip.h)"""
    _fields_ = [("seed_xor", ctypes.c_uint64), ("adam_step", ctypes.c_int32), ("lr", _f), ("gsl_coef", _f), ("reserved", _f)]


class StepDglShard(ctypes.Structure):
    _fields_ = [("own1", ctypes.c_int), ("count1", ctypes.c_double), ("count2", ctypes.c_double)]


_PS = ctypes.POINTER(StepDglShard)
_PG = ctypes.POINTER(StepGwnetParams)

_SIGS = {
    "step_last_error": (ctypes.c_char_p, []),
    "step_abi_version": (_i, []),
    "step_gemm": (_i, [ctypes.POINTER(StepGemm), _vp]),
    "step_tsformer_encode": (_i, [_vp, _i, _i, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _l, _u64, _vp, _vp]),
    "step_dropout_pool_fill": (_i, [_vp, _l, _f, _u64, _vp]),
    "step_tsformer_dropout_words": (_l, [_i, _i]),
    "step_gather_windows": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "step_pack_long_history": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "step_knn_workspace_bytes": (_l, [_i, _i, _i]),
    "step_knn_graph": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _l, _vp]),
    "step_topk_mask": (_i, [_vp, _i, _i, _i, _vp, _vp, _l, _vp]),
    "step_selftest_mfma": (_i, [_vp, _vp]),
    "step_streams_concurrent": (_i, [_vp, _vp, ctypes.POINTER(_i)]),
    "step_dgl_global_saved_floats": (_l, [_i, _i]),
    "step_dgl_global_work_floats": (_l, [_i, _i, _i]),
    "step_dgl_global_forward": (_i, [_vp, _i, _i, _PD, _i, _f, _vp, _vp, _vp, _vp]),
    "step_dgl_global_backward": (_i, [_vp, _i, _i, _PD, _vp, _vp, _vp, _PD, _vp]),
    "step_dgl_global_backward_phase": (_i, [_vp, _i, _i, _PD, _vp, _vp, _vp, _PD, _i, _vp]),
    "step_dgl_global_forward_shard": (_i, [_vp, _i, _i, _PD, _i, _f, _vp, _vp, _vp, _vp, _PS, _i, _vp]),
    "step_dgl_global_backward_shard": (_i, [_vp, _i, _i, _PD, _vp, _vp, _vp, _PD, _PS, _i, _vp]),
    "step_dgl_global_offset": (_l, [_i, _i, _i]),
    "step_dgl_edges_saved_floats": (_l, [_i, _i]),
    "step_dgl_edges_work_floats": (_l, [_i]),
    "step_dgl_edges_theta_offset": (_l, [_i]),
    "step_dgl_edges_forward": (_i, [_vp, _i, _i, _PD, _vp, _u64, _f, _vp, _vp, _vp, _vp]),
    "step_dgl_edges_backward": (_i, [_vp, _i, _i, _PD, _vp, _vp, _vp, _f, _vp, _PD, _vp, _vp, _vp]),
    "step_gwnet_saved_floats": (_l, [_i, _i, _i]),
    "step_gwnet_work_floats": (_l, [_i, _i, _i]),
    "step_gwnet_saved_offset": (_l, [_i, _i, _i, _i, _i]),
    "step_gwnet_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _PG, _i, _f, _u64, _f, _vp, _vp, _vp, _vp]),
    "step_gwnet_forward_phase": (_i, [_vp, _i, _i, _i, _vp, _vp, _PG, _i, _f, _u64, _f, _vp, _vp, _vp, _i, _vp]),
    "step_gwnet_backward": (_i, [_vp, _i, _i, _i, _vp, _PG, _vp, _vp, _vp, _PG, _vp, _i, _vp, _vp, _vp]),
    "step_pt_dropout": (_i, [_vp, _vp, _l, _f, _u64, ctypes.c_uint32, _vp]),
    "step_pt_ffn_hidden_fwd": (_i, [_vp, _vp, _vp, _l, _f, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_ffn_hidden_bwd": (_i, [_vp, _vp, _vp, _l, _f, _vp, _vp]),
    "step_pt_colsum_bf16": (_i, [_vp, _l, _i, _vp, _vp]),
    "step_pt_rows_linear_pack_bytes": (_l, [_i, _i]),
    "step_pt_rows_linear_pack": (_i, [_vp, _l, _l, _i, _i, _vp, _vp, _vp]),
    "step_pt_rows_linear": (_i, [_vp, _i, _l, _vp, _i, _i, _vp, _i, _i, _vp]),
    "step_pt_layer_pack": (_i, [_vp] * 14),
    "step_pt_proj_wgrad_ws_floats": (_l, [_l]),
    "step_pt_proj_wgrad": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_ffn_fused_fwd_ln": (_i, [_vp, _l, _vp, _f, _vp, _l, _u64, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_rows_linear_ln": (_i, [_vp, _l, _vp, _vp, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_embed_unmasked_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_embed_unmasked_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp]),
    "step_pt_dec_input_bwd_sums": (_i, [_vp, _l, _i, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_ffn_pack_bytes": (_l, []),
    "step_pt_ffn_wgrad_workgroups": (_l, [_l]),
    "step_pt_ffn_wgrad_ws_floats": (_l, [_l]),
    "step_pt_ffn_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_ffn_fused_fwd": (_i, [_vp, _l, _vp, _f, _vp, _l, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_ffn_fused_bwd_data": (_i, [_vp, _vp, _l, _vp, _f, _vp, _l, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_ffn_fused_bwd_weights": (_i, [_vp, _vp, _l, _vp, _vp, _f, _vp, _l, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_add_layernorm_fwd": (_i, [_vp, _vp, _l, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_layernorm_bwd_dropout": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _vp, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp]),
    "step_pt_dropout_relu_mask": (_i, [_vp, _vp, _l, _f, _u64, ctypes.c_uint32, _vp]),
    "step_pt_add_dropout": (_i, [_vp, _vp, _vp, _l, _f, _u64, ctypes.c_uint32, _vp]),
    "step_pt_add_rows": (_i, [_vp, _l, _i, _vp, _vp, _vp]),
    "step_pt_sum_over_seq": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _vp]),
    "step_pt_token_gather": (_i, [_vp, _l, _i, _vp, _i, _f, _vp, _vp]),
    "step_pt_token_scatter": (_i, [_vp, _l, _i, _vp, _i, _f, _vp, _vp]),
    "step_pt_dec_input": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _f, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_dec_input_bwd": (_i, [_vp, _l, _i, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp]),
    "step_pt_layernorm_fwd": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_layernorm_bwd": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp]),
    "step_pt_attention_fwd": (_i, [_vp, _l, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp]),
    "step_pt_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _l, _i, _f, _u64, ctypes.c_uint32, _vp, _vp]),
    "step_pt_attention_fwd_bf16": (_i, [_vp, _l, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp, _vp, _l, _vp]),
    "step_pt_attention_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _l, _i, _f, _u64, ctypes.c_uint32, _vp, _vp, _vp]),
    "step_pt_linear_bf16out": (_i, [_vp, _vp, _l, _l, _vp, _l, _i, _i, _vp, _vp]),
    "step_pt_relu_mask": (_i, [_vp, _vp, _l, _vp]),
    "step_colsum": (_i, [_vp, _l, _i, _l, _vp, _vp]),
    "step_loss_fwd_bwd": (_i, [_vp, _vp, _l, _vp, _vp, _l, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "step_loss_scaled_fwd_bwd": (_i, [_vp, _vp, _l, _l, _f, _f, _vp, _vp, _l, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "step_scale2": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp]),
    "step_masked_metrics": (_i, [_vp, _l, _vp, _l, _l, _f, _vp, _vp, _vp]),
    "step_adam_work_floats": (_l, []),
    "step_adam_clip": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp]),
    "step_adam_clip_sharded": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp, _vp]),
    # data-parallel collectives on RCCL's C API (csrc/comm.cpp)
    "step_comm_available": (_i, []),
    "step_comm_version": (_i, []),
    "step_comm_unique_id": (_i, [_vp]),
    "step_comm_init_rank": (_i, [_vp, _i, _i, ctypes.POINTER(_vp)]),
    "step_comm_destroy": (_i, [_vp]),
    "step_comm_set_side_stream": (_i, [_vp, _vp]),
    "step_comm_allreduce": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "step_comm_broadcast": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "step_grad_allreduce": (_i, [_vp, _vp, _l, _vp]),
    "step_grad_allreduce_begin": (_i, [_vp, _vp, _l, _vp]),
    "step_grad_allreduce_join": (_i, [_vp, _vp]),
    # replayable steps: per-step scalars in a device-resident StepDynState (include/step_hip.h)
    "step_dyn_advance": (_i, [_vp, _vp]),
    "step_dropout_pool_fill_dyn": (_i, [_vp, _l, _f, _u64, _vp, _vp]),
    "step_dgl_edges_forward_dyn": (_i, [_vp, _i, _i, _PD, _vp, _u64, _f, _vp, _vp, _vp, _vp, _vp]),
    "step_gwnet_forward_phase_dyn": (_i, [_vp, _i, _i, _i, _vp, _vp, _PG, _i, _f, _u64, _f, _vp, _vp, _vp, _i, _vp, _vp]),
    "step_adam_clip_dyn": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "step_loss_scaled_fwd_bwd_dyn": (_i, [_vp, _vp, _l, _l, _f, _f, _vp, _vp, _l, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


ABI_VERSION = 10
ENC_F16, ENC_ALWAYS_RESHIFT, ENC_RANGE_FLAG = 1, 2, 4          # step_tsformer_encode flags (include/step_hip.h)
COMM_F32, COMM_F64, COMM_U8, COMM_ID_BYTES = 0, 1, 2, 128


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libstep_hip.so not found at {LIB_PATH}: build it with `python -m step_amd.build` "
                "(step_amd has no fallback path)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if _lib.step_abi_version() != ABI_VERSION:      # struct layouts below mirror include/step_hip.h of this version
            v = _lib.step_abi_version()
            _lib = None
            raise RuntimeError(f"libstep_hip.so has ABI version {v}, this package needs {ABI_VERSION}: rebuild with `python -m step_amd.build`")
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = lib().step_last_error().decode(errors="replace")
        raise RuntimeError(f"libstep_hip {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libstep_hip takes contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def gemm(a, b, c, M, N, K, sam, sak, sbk, sbn, ldc, batch=1, sab=0, sbb=0, scb=0, scn=1, alpha=1.0,
         accumulate=0, bias=None, relu=False, splitk=1, a_off=0, b_off=0, c_off=0, a_k=(0, 0), b_k=(0, 0),
         b_n=(0, 0), c_n=(0, 0), a_kscale=None, a_kshift=None, a_kperiod=0, batch0=0, sab1=0, sbb1=0, scb1=0, compute_bf16=False, a_rowsum=None, c_nscale=None, c_nshift=None, c_mvec=None,
         c_nperiod=0, splitk_ws=None):
    """Thin descriptor builder around step_gemm; a/b/c are device tensors (f32 or bf16 for a, b),
    offsets are in elements."""
    g = StepGemm()
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.A = a.data_ptr() + a_off * a.element_size()
    g.sam, g.sak, g.sab, g.a_bf16 = sam, sak, sab, int(a.dtype == torch.bfloat16)
    g.B = b.data_ptr() + b_off * b.element_size()
    g.sbk, g.sbn, g.sbb, g.b_bf16 = sbk, sbn, sbb, int(b.dtype == torch.bfloat16)
    g.C = c.data_ptr() + c_off * 4
    g.ldc, g.scn, g.scb = ldc, scn, scb
    g.alpha, g.accumulate = alpha, accumulate
    g.bias = bias.data_ptr() if bias is not None else None
    g.relu, g.splitk = int(relu), splitk
    g.a_kblk, g.a_kstride = a_k
    g.b_kblk, g.b_kstride = b_k
    g.b_nblk, g.b_nstride = b_n
    g.c_nblk, g.c_nstride = c_n
    g.batch0, g.sab1, g.sbb1, g.scb1 = batch0, sab1, sbb1, scb1
    g.compute_bf16 = int(compute_bf16)
    g.a_rowsum = ptr(a_rowsum) if a_rowsum is not None else None
    g.c_nscale, g.c_nshift, g.c_mvec = (ptr(c_nscale) if c_nscale is not None else None, ptr(c_nshift) if c_nshift is not None else None,
                                         ptr(c_mvec) if c_mvec is not None else None)
    g.c_nperiod = c_nperiod
    if splitk_ws is not None:           # scratch for the partial tiles of a split-K product (summed by a second launch instead of atomics)
        g.splitk_ws, g.splitk_ws_floats = ptr(splitk_ws), splitk_ws.numel()
    if a_kscale is not None:
        g.a_kscale, g.a_kshift, g.a_kperiod = a_kscale.data_ptr(), a_kshift.data_ptr(), a_kperiod
    check(lib().step_gemm(ctypes.byref(g), stream()), "step_gemm")
