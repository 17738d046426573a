"""STEP with the reference's module surface; forward and backward run in libstep_hip.

Mirrors ``step/step_arch/step.py:9-72``: ``STEP(dataset_name, pre_trained_tsformer_path,
tsformer_args, backend_args, dgl_args)``, called by keyword from ``STEPRunner.forward``
(``step_runner.py:66``) and returning ``(prediction [B,12,N,1], theta [B,N,N], adj_knn [B,N,N],
gsl_coefficient)``.  ``prediction`` and ``theta`` are autograd-connected to the trainable
parameters through one ``torch.autograd.Function`` whose forward/backward only launch HIP
kernels through the C ABI (PyTorch supplies device memory, the stream and ``.grad`` plumbing).
"""
import ctypes
import os

import torch
from torch import nn

from .. import _lib
from .tsformer import TSFormer
from .graphwavenet import GraphWaveNet, fill_gwnet_struct
from .discrete_graph_learning import DiscreteGraphLearning, fill_dgl_struct

TEMPERATURE = 0.5        # discrete_graph_learning.py:157
BN_MOMENTUM = 0.1


def _f32(n, device):
    return torch.empty(int(n), device=device, dtype=torch.float32)


class LongHistoryRef:
    """Stand-in for the ``long_history_data`` tensor [B, L, N, C] of a batch whose series lives on the device
    (SURVEY 8f-1): ``data`` f32 [T, N, C] resident in HBM, ``t0`` int64 [B] forecast origins, ``length`` = L.  STEP.forward
    accepts it in place of the tensor and gathers the encoder input directly (no 14.9 MB/window copy).
    It answers the three things the reference's runner does to the batch tensor before the model sees it
    (step/step_runner/step_runner.py:58-63): ``.to(device)`` / ``.cuda()`` (already there: itself), ``.shape``, and the feature
    selection ``data[:, :, :, forward_features]`` (a reference to the same windows exposing those channels)."""

    def __init__(self, data, t0, length, channels=None):
        assert data.is_cuda and data.dtype == torch.float32 and data.dim() == 3 and data.is_contiguous()
        assert t0.is_cuda and t0.dtype == torch.int64 and t0.dim() == 1
        self.data, self.t0, self.length = data, t0, int(length)
        self.channels = list(range(data.shape[2])) if channels is None else [int(c) for c in channels]      # data channels this reference exposes, in order

    @property
    def shape(self):
        return (self.t0.shape[0], self.length, self.data.shape[1], len(self.channels))

    @property
    def device(self):
        return self.data.device

    is_cuda = True

    def to(self, *args, **kwargs):
        return self

    def cuda(self, *args, **kwargs):
        return self

    def __getitem__(self, idx):
        full = slice(None)
        if isinstance(idx, tuple) and len(idx) == 4 and all(i == full for i in idx[:3] if isinstance(i, slice)) and all(isinstance(i, slice) for i in idx[:3]):
            sel = idx[3]
            if isinstance(sel, slice):
                sel = list(range(len(self.channels)))[sel]
            elif torch.is_tensor(sel):
                sel = sel.tolist()
            return LongHistoryRef(self.data, self.t0, self.length, [self.channels[int(c)] for c in sel])
        raise IndexError("LongHistoryRef supports the runner's feature selection only: ref[:, :, :, features]")


class DeviceWindowLoader:
    """Index-only loader over a device-resident series: yields (history_data [B,12,N,C], LongHistoryRef, future_data [B,12,N,C])
    for a batch of forecast origins; one gather launch, nothing crosses PCIe."""

    def __init__(self, data, long_len, horizon=12):
        self.data = data.contiguous().float()
        assert self.data.is_cuda
        self.long_len, self.horizon = int(long_len), int(horizon)

    def batch(self, t0):
        t0 = torch.as_tensor(t0, dtype=torch.int64, device=self.data.device).contiguous()
        B, (T, N, C), H = t0.shape[0], self.data.shape, self.horizon
        hist = torch.empty(B, H, N, C, device=self.data.device)
        fut = torch.empty(B, H, N, C, device=self.data.device)
        _lib.call("step_gather_windows", _lib.ptr(self.data), T, N, C, 0, _lib.ptr(t0), B, 0, H, None, _lib.ptr(hist), _lib.ptr(fut),
                  _lib.stream())
        return hist, LongHistoryRef(self.data, t0, self.long_len), fut


class _StepFunction(torch.autograd.Function):
    """Inputs: (model, hist [B,12,N,C], long_hist [B,L,N,C], u or None, *trainable tensors)."""

    @staticmethod
    def forward(ctx, model, hist, long_hist, u, *params):
        L = _lib
        dev = hist.device
        B, T12, N, Cin = hist.shape
        Lh = long_hist.shape[1]
        training = model.training
        st = L.stream()
        hist = hist.contiguous().float()
        frozen = model._take_prefetched(long_hist)      # the frozen branch of this batch, if STEP.prefetch() already queued it
        # Everything up to the GraphWaveNet head is independent of the (frozen) TSFormer: the DGL's global feature, the edge
        # logits, the Gumbel sample and the 8 WaveNet layers only need the train series, the short history and the weights.
        # They are queued on a second stream next to the encoder: its workgroups keep the compute units busy while that chain of
        # small, latency-bound kernels advances.  Buffers are allocated on the main stream and outlive the join below.
        dgl, be = model.discrete_graph_learning, model.backend
        sh = dgl._shard                      # time slice of the global branch on this data-parallel rank (or None)
        if sh is not None and not training:
            # Evaluation of a sharded model: the reference's validate() / test() are @master_only (base_tsf_runner.py:276,320), so
            # the other ranks are not here to answer the slices' collectives.  The whole graph learner is evaluated on this rank
            # from the gathered fc.weight -- no collective -- and refuses to run on stale weights.
            if dgl._slice_dirty:
                raise RuntimeError("evaluation of a time-sliced graph learner needs the gathered fc.weight: call "
                                   "model.discrete_graph_learning.gather_fc_weight() on ALL ranks after the training epoch "
                                   "(the runner's on_epoch_end) before the master-only validation")
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                raise RuntimeError("a time-sliced graph learner evaluates without gradients only (wrap the call in torch.no_grad())")
            sh = None
        bf = {"f32": 0, "bf16": 1}[model.matmul_precision]
        dstruct, bstruct = model._param_structs(bf, full=sh is None)
        Ttr = dgl.train_length if sh is None else sh["Ts"]
        series_nt = dgl._series_nt if sh is None else dgl._series_slice
        drop = be.dropout if training else 0.0
        gsaved = _f32(L.lib().step_dgl_global_saved_floats(N, Ttr), dev)
        gwork = _f32(L.lib().step_dgl_global_work_floats(N, Ttr, 0), dev)
        g = _f32(N * 100, dev).view(N, 100)
        esaved = _f32(L.lib().step_dgl_edges_saved_floats(B, N), dev)
        # theta is one of the tensors the edge backward reads from `esaved`: returned as a view of it, written once (a separate output
        # tensor cost a 3 MB device copy on the second stream's chain, 217 us next to the encoder)
        to = int(L.lib().step_dgl_edges_theta_offset(N))
        theta = esaved[to:to + B * N * N].view(B, N, N)
        adj = _f32(B * N * N, dev).view(B, N, N)
        wsaved = _f32(L.lib().step_gwnet_saved_floats(B, N, int(drop > 0)), dev)
        wwork = _f32(L.lib().step_gwnet_work_floats(B, N, 0), dev)
        pred = _f32(B * 12 * N, dev).view(B, 12, N)
        seed, seed_gw = model._next_seed(), model._next_seed()
        side = model._side_stream(dev) if model.overlap_streams else None
        main = torch.cuda.current_stream()
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(main)              # inputs, weights (the previous optimizer step) and the noise are ordered before this point
        P = Lh // 12
        # What the WaveNet needs before its first diffusion hop and that does not read the sampled adjacency -- start convolution, adaptive
        # support, weight packing, layer 0's gated TCN (phase 5, ~60 us of small dependent kernels) -- runs on a third stream NEXT TO the
        # graph learner instead of behind it; the layers (phase 6) wait for both
        prep_done = None
        if side is not None and model.split_wavenet_prep and model._dyn is None:
            prep = model._side_stream(dev, "aux")
            prep.wait_event(ready)
            with torch.cuda.stream(prep):
                hist.record_stream(prep)
                L.call("step_gwnet_forward_phase_dyn", L.ptr(hist), B, N, Cin, None, None, ctypes.byref(bstruct),
                       int(training) | (2 if (training and model.track_dead_bn7) else 0),
                       float(drop), seed_gw, BN_MOMENTUM, L.ptr(wsaved), L.ptr(wwork), None, 5, None, L.stream())
                prep_done = torch.cuda.Event()
                prep_done.record(prep)

        def graph_and_layers(sst):
            if sh is None:
                L.call("step_dgl_global_forward", L.ptr(series_nt), N, Ttr, ctypes.byref(dstruct), int(training), BN_MOMENTUM,
                       L.ptr(gsaved), L.ptr(gwork), L.ptr(g), sst)
            else:
                # the slices meet in three small sums over the ranks: BatchNorm1 sums, BatchNorm2 sums, the [N,100] partial fc product
                sstruct = dgl.shard_struct()
                sums = torch.zeros(48, dtype=torch.float64, device=dev)
                go = L.lib().step_dgl_global_offset(N, Ttr, 2)
                exchange = {1: sums[:16], 2: sums[16:], 3: gsaved[go:go + N * 100]}
                for phase in (1, 2, 3, 4):
                    L.call("step_dgl_global_forward_shard", L.ptr(series_nt), N, Ttr, ctypes.byref(dstruct), int(training), BN_MOMENTUM,
                           L.ptr(gsaved), L.ptr(gwork), L.ptr(sums), L.ptr(g), ctypes.byref(sstruct), phase, sst)
                    if phase in exchange:
                        model._sum_over_ranks(exchange[phase])
            L.call("step_dgl_edges_forward_dyn", L.ptr(g), N, B, ctypes.byref(dstruct), L.ptr(u) if u is not None else None, seed,
                   TEMPERATURE, L.ptr(esaved), L.ptr(theta), L.ptr(adj), L.ptr(model._dyn), sst)
            if prep_done is not None:
                torch.cuda.current_stream().wait_event(prep_done)
            L.call("step_gwnet_forward_phase_dyn", L.ptr(hist), B, N, Cin, None, L.ptr(adj), ctypes.byref(bstruct),
                   int(training) | (2 if (training and model.track_dead_bn7) else 0),
                   float(drop), seed_gw, BN_MOMENTUM, L.ptr(wsaved), L.ptr(wwork), None, 6 if prep_done is not None else 1, L.ptr(model._dyn), sst)
            if training:
                with torch.no_grad():
                    # the BatchNorm step counters: one multi-tensor launch instead of ten scalar ones, on this (the second) stream -- on the
                    # main stream it sat between the head and the loss
                    torch._foreach_add_([m.num_batches_tracked for m in (dgl.bn1, dgl.bn2, dgl.bn3, *list(be.bn)[:8 if model.track_dead_bn7 else 7])], 1)

        # ---- TSFormer (frozen) and the kNN prior graph (no grad): on the main stream, or already in flight on the prefetch stream
        if frozen is None:
            # (the kNN prior is only needed by the loss; running its Gram product and top-k selection on a third stream next to the
            #  GraphWaveNet head was measured: 4.72 vs 4.69 ms, the Gram product takes the compute units from the head's small kernels --
            #  `knn_stream` of _frozen_branch stays available, STEP_KNN_STREAM=1)
            ks = model._side_stream(dev, "knn") if (side is not None and os.environ.get("STEP_KNN_STREAM", "0") == "1") else None
            frozen = model._frozen_branch(long_hist, B, N, knn_stream=ks)
        if side is not None:
            side.wait_event(ready)
            try:
                with torch.cuda.stream(side):
                    hist.record_stream(side)
                    if u is not None:
                        u.record_stream(side)
                    graph_and_layers(L.stream())
            except BaseException:
                # the join is queued even when a launch above raised: buffers handed back to the allocator must not be
                # re-used by main-stream work while side-stream kernels may still write them
                main.wait_stream(side)
                raise
        else:
            graph_and_layers(st)
        if frozen["done"] is not None:
            main.wait_event(frozen["done"])
        enc, sim, adj_knn = frozen["enc"], frozen["sim"], frozen["adj_knn"]
        # ---- GraphWaveNet head.  The fc_his branch is the only consumer of the TSFormer's last hidden state and needs nothing else: it is
        # queued behind the encoder (and the kNN prior) BEFORE the main stream waits for the layers on the second stream
        try:
            L.call("step_gwnet_forward_phase", L.ptr(hist), B, N, Cin, L.ptr(enc["last"]), None, ctypes.byref(bstruct), int(training),
                   float(drop), seed_gw, BN_MOMENTUM, L.ptr(wsaved), L.ptr(wwork), None, 3, st)
        finally:
            if side is not None:
                main.wait_stream(side)
        L.call("step_gwnet_forward_phase", L.ptr(hist), B, N, Cin, None, None, ctypes.byref(bstruct), int(training),
               float(drop), seed_gw, BN_MOMENTUM, L.ptr(wsaved), L.ptr(wwork), L.ptr(pred), 4, st)
        if frozen.get("knn_done") is not None:
            main.wait_event(frozen["knn_done"])          # adj_knn / sim are handed to the caller on the current stream
        ctx.model = model
        ctx.dims = (B, N, Cin, Ttr, float(drop))
        ctx.held = (hist, enc["last"], g, gsaved, esaved, wsaved)
        ctx.esaved_version = esaved._version
        ctx.mark_non_differentiable(adj_knn)
        model._last = {"sampled_adj": adj, "sim": sim, "hidden_bf16": enc["hidden_bf16"], "saved_gwnet": wsaved, "g": g,
                       "hidden_last": enc["last"]}
        return pred.unsqueeze(-1), theta, adj_knn

    @staticmethod
    def backward(ctx, dpred, dtheta, _dknn):
        L = _lib
        model = ctx.model
        B, N, Cin, Ttr, drop = ctx.dims
        hist, last, g, gsaved, esaved, wsaved = ctx.held
        dev = hist.device
        st = L.stream()
        be, dgl = model.backend, model.discrete_graph_learning
        layout = model._grad_layout()
        # the flat gradient buffer: everything in front of the fc weight (gradients accumulate into zeros, 18 MB at PEMS04) is cleared; the fc
        # weight's 87 MB are STORED by the graph learner's backward (STEP_DGL_FRESH_FC_GRAD), neither cleared here nor read there
        fo, fn, _ = layout["items"]["dgl.fc_w"]
        flat, views, gw_grads, dg_grads = model._grad_buffers(layout, dev)
        flat[:fo].zero_()
        if fo + fn < layout["total"]:
            flat[fo + fn:].zero_()
        bf = {"f32": 0, "bf16": 1}[model.matmul_precision]
        dstruct, bstruct = model._param_structs(bf, full=False)
        dpred = dpred.contiguous().float().view(B, 12, N) if dpred is not None else torch.zeros(B, 12, N, device=dev)
        dadj = _f32(B * N * N, dev)
        wwork = _f32(L.lib().step_gwnet_work_floats(B, N, 1), dev)
        # the weight / bias gradients of the WaveNet are leaves of the backward: the library forks them onto the second stream
        use_aux = model.overlap_streams and os.environ.get("STEP_NO_AUX", "0") != "1"
        aux = ctypes.c_void_p(model._side_stream(dev, "aux").cuda_stream) if use_aux else None
        # (a stream of their own for the pure leaves was measured and changes nothing -- 4.51 vs 4.51 ms at PEMS04, profiles/r03_ac_* -- so
        #  the library's leaf_stream stays NULL and they share "aux": one hardware queue fewer in use; STEP_LEAF_STREAM=1 turns it on)
        leaf = ctypes.c_void_p(model._side_stream(dev, "leaf").cuda_stream) if (use_aux and os.environ.get("STEP_LEAF_STREAM", "0") == "1") else None
        joined = [False]

        def join_aux():
            # the leaves the two library calls below leave on the auxiliary stream (parameter gradients only) must be finished before the
            # first reader of the flat gradient buffer: the all-reduce of everything in front of fc_w, or the optimizer
            if aux is not None:
                torch.cuda.current_stream().wait_stream(model._side_stream(dev, "aux"))
            if leaf is not None:
                torch.cuda.current_stream().wait_stream(model._side_stream(dev, "leaf"))
            joined[0] = True
        if esaved._version != ctx.esaved_version:
            # theta is returned as a view of this buffer (forward); the edge backward reads it -- an in-place edit by user code (e.g.
            # theta.clamp_() in a custom loss) would silently corrupt the gradients, and autograd cannot see it (ctx.held, not save_for_backward)
            raise RuntimeError("step_amd.STEP: the returned edge probabilities (theta) were modified in place after forward(); the native "
                               "backward reads that buffer -- use an out-of-place op (theta.clamp(...)) or theta.clone()")
        FRESH = 16          # include/step_hip.h STEP_DGL_FRESH_FC_GRAD
        try:
            L.call("step_gwnet_backward", L.ptr(hist), B, N, Cin, L.ptr(last), ctypes.byref(bstruct), L.ptr(wsaved), L.ptr(wwork),
                   L.ptr(dpred), ctypes.byref(gw_grads), L.ptr(dadj), int(drop > 0), aux, leaf, st)
            # (wwork / ework stay referenced until the auxiliary stream is joined below: its last leaves still read them)
            ework = _f32(L.lib().step_dgl_edges_work_floats(N), dev)
            dgv = _f32(N * 100, dev)
            dth = dtheta.contiguous().float() if dtheta is not None else None
            L.call("step_dgl_edges_backward", L.ptr(g), N, B, ctypes.byref(dstruct), L.ptr(esaved), L.ptr(dth) if dth is not None else None,
                   L.ptr(dadj), TEMPERATURE, L.ptr(ework), ctypes.byref(dg_grads), L.ptr(dgv), leaf if leaf is not None else aux, st)
            gwork = _f32(L.lib().step_dgl_global_work_floats(N, Ttr, 1), dev)
            fo, fn, _ = layout["items"]["dgl.fc_w"]
            assert fo + fn == layout["total"] or fo + ((fn + 3) & ~3) == layout["total"]
            assert layout["norm_slot"] == fo - 4
            sh = dgl._shard
            if sh is None:
                L.call("step_dgl_global_backward_phase", L.ptr(dgl._series_nt), N, Ttr, ctypes.byref(dstruct), L.ptr(gsaved), L.ptr(dgv),
                       L.ptr(gwork), ctypes.byref(dg_grads), 1 | FRESH, st)
                pending = model._reduce_begin(flat[fo:fo + fn])          # overlaps with the conv / BatchNorm backward below
                L.call("step_dgl_global_backward_phase", L.ptr(dgl._series_nt), N, Ttr, ctypes.byref(dstruct), L.ptr(gsaved), L.ptr(dgv),
                       L.ptr(gwork), ctypes.byref(dg_grads), 2, st)
                join_aux()
                pending += model._reduce_begin(flat[:fo])
                model._reduce_finish(flat, pending, flat)
            else:
                # time slices: the gradient of g is averaged over the ranks first, every rank then back-propagates its slice; the slices
                # meet in two small sums (BatchNorm2's 32 "dots", the 1296 raw conv2 weight-gradient sums).  The fc weight slice's
                # gradient stays on its rank -- 98 % of the gradient bytes never enter a collective.
                world = sh["world"]
                model._sum_over_ranks(dgv)
                dgv.mul_(1.0 / world)
                dgl._slice_dirty = True          # the optimizer is about to change the slices: fc.weight / state_dict() are stale until a gather
                sstruct = dgl.shard_struct()
                o_dots, o_graw = L.lib().step_dgl_global_offset(N, Ttr, 10), L.lib().step_dgl_global_offset(N, Ttr, 11)
                exchange = {1: gwork[o_dots:o_dots + 32], 3: gwork[o_graw:o_graw + 1296]}
                for phase in (1, 3, 4):
                    L.call("step_dgl_global_backward_shard", L.ptr(dgl._series_slice), N, Ttr, ctypes.byref(dstruct), L.ptr(gsaved), L.ptr(dgv),
                           L.ptr(gwork), ctypes.byref(dg_grads), ctypes.byref(sstruct), phase | FRESH, st)
                    if phase in exchange:
                        model._sum_over_ranks(exchange[phase])
                join_aux()
                # conv1's gradients are per-slice partial sums: summed, not averaged, by the mean all-reduce below
                views["dgl.conv1_w"].mul_(float(world))
                views["dgl.conv1_b"].mul_(float(world))
                # the squared norm of this rank's fc-slice gradient rides in the spare slot (x world: the reduction below takes the mean)
                ns = layout["norm_slot"]
                own = torch.linalg.vector_norm(flat[fo:fo + fn]).reshape(1)
                torch.mul(own.square(), float(world), out=flat[ns:ns + 1])
                model._reduce_finish(flat, model._reduce_begin(flat[:fo]), flat[:fo])
                # the slot now holds sum_r |g_fc slice of rank r|^2 (mean of world * own_r^2): keep the other ranks' part for the clip norm
                # (FusedAdamClip) and clear the slot, so that the flat buffer holds gradients only
                model._other_slices_sumsq = torch.addcmul(flat[ns:ns + 1], own, own, value=-1.0)
                slices = flat[ns:ns + 1].clone()
                flat[ns:ns + 1].zero_()
                # ... and the squared norm of the WHOLE gradient, formed the same way on every rank: replicated part (identical everywhere after
                # the all-reduce) + the reduced slot.  "own buffer + other slices" is added in a different order on every rank, differs in the
                # last bit, and the clip factor then lets the replicated parameters drift apart (tests/test_gpu_comm_two_ranks.py found it)
                rep = torch.linalg.vector_norm(flat[:fo])
                model._total_sumsq = torch.addcmul(slices, rep, rep)
        finally:
            # also on the error path: the unjoined leaves still read wwork / ework / wsaved / g / hidden_last and write `flat`; once this frame's
            # locals go back to the caching allocator, main-stream work must be ordered behind them (mirrors the forward's guard)
            if not joined[0]:
                join_aux()
        del wwork, ework
        model._flat_grad = flat
        model._backward_count = getattr(model, "_backward_count", 0) + 1
        ctx.held = None
        del views, gw_grads, dg_grads
        if model.flat_gradients_only:
            # the caller reads model._flat_grad (step_amd.optim.FusedAdamClip does): no per-parameter .grad tensors, no AccumulateGrad work
            return (None, None, None, None) + (None,) * len(layout["order"])
        # FRESH views with no other owner, built on every backward: autograd's AccumulateGrad adopts such a tensor as .grad (an alias of
        # the flat buffer) but CLONES one that somebody else still references -- a cached tuple made every step copy 105 MB (the fc
        # weight gradient alone is 87 MB) and left .grad pointing away from model._flat_grad, which a DistributedDataParallel wrap or
        # FusedAdamClip(param_grads=True) rely on (tests/test_gpu_step.py::test_param_grads_alias_flat_buffer_every_step)
        return (None, None, None, None) + tuple(flat[o:o + n].view(shape) for o, n, shape in (layout["items"][k] for k in layout["order"]))


_STREAMS = {}        # (name, device type, device index) -> torch.cuda.Stream, see STEP._side_stream


def _concurrent_stream(dev, tries=8, priority=0):
    """A new stream that RUNS CONCURRENTLY with the current stream.  The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues by
    least use, and a stream that lands on the main stream's queue serialises behind it -- silently: with a process group in the process
    (RCCL, torch.distributed and the native communicator create streams of their own) the step's second stream did, and the graph
    learner's whole forward chain waited for the encoder (+0.5 ms per step at PEMS04, profiles/r05_h_dp_one_rank.log).  So candidates are
    created until the probe of libstep_hip (step_streams_concurrent: a 200 us spin against an empty kernel) sees one overlap with the
    current stream; the rejected ones are dropped, which frees their queue slots.  STEP_STREAM_PROBE=0 takes the first stream."""
    if os.environ.get("STEP_STREAM_PROBE", "1") == "0":
        return torch.cuda.Stream(device=dev, priority=priority)
    main = torch.cuda.current_stream(dev)
    mine = [(n, v) for (n, t, i), v in _STREAMS.items() if t == dev.type and i == dev.index]      # the step's other streams on this device
    kept, flag = [], ctypes.c_int(0)

    def overlaps(a, b):
        _lib.call("step_streams_concurrent", ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream), ctypes.byref(flag))
        return bool(flag.value)
    best, best_score = None, -1
    full = 100 + 10 * sum(1 for n, _ in mine if n == "prefetch") + sum(1 for n, _ in mine if n != "prefetch")
    for _ in range(tries):
        s = torch.cuda.Stream(device=dev, priority=priority)
        kept.append(s)          # kept alive until the search ends, so that the next candidate gets another queue
        # what matters, in this order: the main stream (everything forks from it and joins it); the prefetch stream (it carries the
        # persistent encoder for ~3 ms: anything queued behind it waits that long); then a queue of its own against the rest, when the
        # runtime has one left (four hardware queues by default)
        score = 0
        if overlaps(main, s):
            score = 100
            for n, o in mine:
                if overlaps(o, s):
                    score += 10 if n == "prefetch" else 1
        if score > best_score:
            best, best_score = s, score
        if score >= full:
            break
    return best


class STEP(nn.Module):
    """Pre-training Enhanced Spatial-temporal Graph Neural Network -- MI355X-native drop-in."""

    def __init__(self, dataset_name, pre_trained_tsformer_path, tsformer_args, backend_args, dgl_args):
        super().__init__()
        self.dataset_name = dataset_name
        self.pre_trained_tsformer_path = pre_trained_tsformer_path
        self.tsformer = TSFormer(**tsformer_args)
        self.backend = GraphWaveNet(**backend_args)
        self.load_pre_trained_model()
        self.discrete_graph_learning = DiscreteGraphLearning(**dgl_args)
        self.gumbel_noise = "device"        # "torch_cpu": draw torch.rand on the host like the reference (:12)
        # "f32": every contraction outside the TSFormer on the exact-f32 matrix cores (tight parity with the oracle);
        # "bf16": every GEMM-shaped contraction of the GraphWaveNet and the DGL (hops, 1x1 / gate / mix / head layers and their
        # weight gradients, DGL conv and fc) runs on the bf16 matrix cores with f32 accumulation (BASELINE config "bf16")
        self.matmul_precision = "f32"
        # The last WaveNet layer's gcn + BatchNorm output is dead code in the reference (model.py:202-213), but its forward still runs
        # there and moves bn.7's running statistics.  True: evaluate that layer's gcn for its batch statistics as well (two hop
        # launches, the mix kernel and a finalize: ~40 us per step) so that a checkpoint written after training round-trips all of
        # the reference's buffers; False (default): skip the dead work, bn.7's three buffers stay at their initial values.
        self.track_dead_bn7 = False
        self._noise_override = None         # tests: explicit uniform noise [B, N*N, 2]
        self._seed_ctr = 0
        self._dyn = None                    # device StepDynState of a replayed (graph-captured) step (step_amd/graphed.py); None: eager
        self._process_group = None
        self._struct_cache = {}             # (bf16 flag, full) -> (address key, StepDglParams, StepGwnetParams), see _param_structs
        self._bwd_cache = None              # the reused flat gradient buffer with its views and pointer structs, see _grad_buffers
        self._trainable_cache = None
        # True (set by FusedAdamClip(..., param_grads=False)): backward leaves the gradients in model._flat_grad ONLY -- no per-parameter
        # .grad tensors are handed to autograd (that is ~100 AccumulateGrad nodes, 0.3-0.5 ms of host time per step)
        self.flat_gradients_only = False
        self._comm = None                   # step_amd.comm.NativeComm when the collectives are RCCL C-API calls (enable_native_data_parallel)
        self._min_world = 1                 # collectives are issued for groups larger than this (0: also for a single rank)
        self._layout = None
        self._zg_params = None
        self._zg_key = None
        self._last = {}
        self._flat_param = None
        self._flat_grad = None
        self._backward_count = 0
        self.overlap_streams = os.environ.get("STEP_NO_OVERLAP", "0") != "1"      # graph learner + WaveNet layers next to the encoder
        # the adjacency-independent start of the WaveNet (library phases 5 / 6) on a third stream next to the graph learner: measured, no gain
        # (3.496 / 3.495 vs 3.476 / 3.491 ms at PEMS04, profiles/r06_e_*: the step is bound by the prefetched branch, not by this chain) -- off
        self.split_wavenet_prep = os.environ.get("STEP_PREP_SPLIT", "0") == "1"
        # prefetch(): the kNN prior (Gram product + top-k) of the announced batch on a stream of its own instead of behind its encoder on the
        # prefetch stream -- only the loss needs it, half a step later.  Pays where the branch is announced at the START of the step
        # (PEMS04: 3.66 -> 3.55 ms on one box) and costs where it is announced late and the prior is large (PEMS07, 883 nodes: 5.57 -> 6.68 ms,
        # profiles/r06_j_knn_stream_C4.log): off here, bench.py and step_amd.runner switch it on with the early announcement
        self.prefetch_knn_stream = os.environ.get("STEP_PREFETCH_KNN_STREAM", "0") == "1"
        self._prefetched = None             # FIFO (list) of the frozen branches queued by prefetch() for upcoming batches
        self._precision_override = None
        # announced branches kept at once: 2 = batch i + 1 announced before forward(i) consumed batch i's; 3 = batch i + 2 announced at the
        # start of step i -- the encoders of successive batches then run back to back on the prefetch stream (see bench.py --prefetch-ahead)
        self.prefetch_fifo = 2
        self._alias = {}                    # batch key of a derived tensor -> key of the announced batch it was made from (alias_batch)
        self.prefetch_enabled = os.environ.get("STEP_NO_PREFETCH", "0") != "1"
        self._reduce_wait_ms = None         # bench.py: list that collect_reduce_waits() fills
        self._reduce_events = []
        self._small_events = []             # (start, end, bytes) of the time-sliced graph learner's blocking sums (bench.py)

    def load_pre_trained_model(self):
        """step.py:27-35: load {"model_state_dict": ...} and freeze."""
        if self.pre_trained_tsformer_path is not None:
            ckpt = torch.load(self.pre_trained_tsformer_path, map_location="cpu")
            self.tsformer.load_state_dict(ckpt["model_state_dict"])
        for p in self.tsformer.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ helpers
    def _side_stream(self, dev, name="side"):
        """The model's extra streams: "side" (forward: graph learner + WaveNet layers next to the encoder), "aux" (backward: the leaves).
        Stream priorities were measured and change nothing here (the device offers two levels; 4.683 vs 4.686 ms with the side chain at
        high priority, profiles/r03_g_stream_priority_ab.log): next to the encoder a small kernel still waits ~150 us for a compute
        unit, because a freed unit goes back to the encoder's queue as often as to the other one."""
        # One set of streams per PROCESS and device, shared by every model: the runtime multiplexes streams onto a handful of hardware
        # queues (four by default), and two streams that land on one queue serialise.  With per-model streams a second model built in the
        # same process (bench.py's other configs) ran its "side" / "aux" work on queues its own main stream was using: PEMS07 6.83 ms
        # instead of 6.07 ms (profiles/r03_ag_*, r03_ah_*).
        key = (name, dev.type, dev.index)
        if key not in _STREAMS:
            _STREAMS[key] = _concurrent_stream(dev, priority=int(os.environ.get("STEP_PRIORITY_" + name.upper(), "0")))
        return _STREAMS[key]

    # ------------------------------------------------------------------ the frozen branch (TSFormer + kNN prior)
    def _frozen_branch(self, long_hist, B, N, knn_stream=None, channel=0):
        """[B,L,N,C] long history (or a LongHistoryRef) -> TSFormer hidden states -> cosine kNN prior graph, queued on the CURRENT
        stream.  Depends only on the input and the frozen TSFormer (step.py:34-35, discrete_graph_learning.py:139,164-166):
        no parameter the optimizer touches is read, which is what lets prefetch() run it a step ahead."""
        L = _lib
        st = L.stream()
        dev = self.backend.nodevec1.device
        Lh = long_hist.shape[1]
        series = _f32(B * N * Lh, dev).view(B * N, Lh)
        if isinstance(long_hist, LongHistoryRef):          # index-only loader: gather straight from the resident series
            d = long_hist.data
            L.call("step_gather_windows", L.ptr(d), d.shape[0], d.shape[1], d.shape[2], long_hist.channels[channel], L.ptr(long_hist.t0), B, Lh, 12,
                   L.ptr(series), None, None, st)
        else:
            long_hist = long_hist.contiguous().float()
            L.call("step_pack_long_history", L.ptr(long_hist), B, Lh, N, long_hist.shape[3], int(channel), L.ptr(series), st)
        P = Lh // 12
        enc = self.tsformer.encode_series(series)
        sim = _f32(B * N * N, dev).view(B, N, N)
        adj_knn = _f32(B * N * N, dev).view(B, N, N)
        kwork = torch.empty(L.lib().step_knn_workspace_bytes(B, N, P * 96), dtype=torch.uint8, device=dev)
        knn_done = None
        if knn_stream is not None:
            enc_done = torch.cuda.Event()
            enc_done.record()
            knn_stream.wait_event(enc_done)
            with torch.cuda.stream(knn_stream):
                for t in (enc["hidden_bf16"], enc["sqnorm"], sim, adj_knn, kwork):
                    t.record_stream(knn_stream)
                L.call("step_knn_graph", L.ptr(enc["hidden_bf16"]), L.ptr(enc["sqnorm"]), B, N, P * 96, self.discrete_graph_learning.k * N,
                       L.ptr(sim), L.ptr(adj_knn), L.ptr(kwork), kwork.numel(), L.stream())
                knn_done = torch.cuda.Event()
                knn_done.record()
        else:
            L.call("step_knn_graph", L.ptr(enc["hidden_bf16"]), L.ptr(enc["sqnorm"]), B, N, P * 96, self.discrete_graph_learning.k * N,
                   L.ptr(sim), L.ptr(adj_knn), L.ptr(kwork), kwork.numel(), st)
        return {"enc": enc, "sim": sim, "adj_knn": adj_knn, "held": (series, kwork, long_hist), "done": None, "knn_done": knn_done}

    @staticmethod
    def _batch_key(long_hist, channel=0):
        """identity of (a batch, the channel of it the TSFormer reads): storage, leading shape, version, data channel"""
        if isinstance(long_hist, LongHistoryRef):
            t = long_hist.t0
            return (t.data_ptr(), tuple(long_hist.shape[:3]), t._version, True, long_hist.channels[channel])
        return (long_hist.data_ptr(), tuple(long_hist.shape[:3]), long_hist._version, False, int(channel))

    def alias_batch(self, source, derived, channel=0):
        """``derived`` (what forward() is about to be called with) carries, in its channel 0, the values of channel ``channel`` of
        ``source`` -- the batch tensor the loader announced with ``prefetch(source, channel)``.  The reference's runner does exactly
        this between its loader and the model (``data[:, :, :, forward_features]``, step_runner.py:25-26,61-63: advanced indexing,
        a copy); step_amd.runner registers the pair so that forward() still finds the prefetched branch."""
        if len(self._alias) > 8:
            self._alias.clear()
        self._alias[self._batch_key(derived, 0)] = self._batch_key(source, channel)

    def prefetch(self, long_history_data, channel=0):
        """Queue the frozen branch (TSFormer encoder + kNN prior) of an UPCOMING batch on its own stream, so that it runs next to
        the backward pass and optimizer step of the current one.  Legal because the TSFormer is frozen in this stage
        (step.py:34-35): nothing the optimizer updates is read, and the branch's outputs are bit-identical to computing them inside
        forward() -- dropout seeds are drawn per encoder launch in the same order either way.  forward() recognises the batch by
        the identity of ``long_history_data`` (storage, shape, version) and falls back to the inline path for any other input.
        Training-loop use:  ``out = model(batch_i); model.prefetch(long_history_of_batch_i+1); loss.backward(); opt.step()`` -- or, with
        ``tsformer.encoder_workgroups`` set, ``model.prefetch(long_history_of_batch_i+1)`` BEFORE ``model(batch_i)`` (up to two batches may be
        queued): the next batch's encoder then shares the chip with the whole of step i ("Step schedule")."""
        if not self.prefetch_enabled:
            return
        if isinstance(long_history_data, LongHistoryRef):
            dev = long_history_data.data.device
        else:
            if not long_history_data.is_cuda:
                raise RuntimeError("step_amd.STEP runs only on an AMD GPU: libstep_hip has no CPU fallback")
            dev = long_history_data.device
        B, _, N, _ = long_history_data.shape
        main = torch.cuda.current_stream()
        ps = self._prefetch_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)                  # the batch (a gather on the main stream) and any inline encoder launch come first
        ps.wait_event(ready)
        mode = self.training
        # (never under a process group: RCCL's and the communicator's streams already compete for the four hardware queues, and a fifth stream
        #  of the step then shares a queue with one that must not wait -- 4.89 instead of 3.57 ms at PEMS04, profiles/r06_l_knn_stream_process_group.log)
        import torch.distributed as dist
        grouped = self._process_group is not None or (dist.is_available() and dist.is_initialized())
        ks = self._side_stream(dev, "knn") if (self.prefetch_knn_stream and not grouped) else None
        with torch.cuda.stream(ps):
            rec = self._frozen_branch(long_history_data, B, N, knn_stream=ks, channel=channel)
            rec["done"] = torch.cuda.Event()          # (with a kNN stream: the encoder; the prior graph has its own event, knn_done)
            rec["done"].record(ps)
        # everything the branch allocated is consumed on the main stream after the `done` event
        for t in (rec["enc"]["hidden_bf16"], rec["enc"]["last"], rec["enc"]["sqnorm"], rec["sim"], rec["adj_knn"]):
            if t is not None:
                t.record_stream(main)
        rec["key"], rec["training"] = self._batch_key(long_history_data, channel), mode
        # a short FIFO: the branch of batch i + 1 may be queued before forward() has consumed the one of batch i (the encoder then runs
        # next to the WHOLE of step i, not only its backward)
        q = self._prefetched if isinstance(self._prefetched, list) else []
        q.append(rec)
        while len(q) > self.prefetch_fifo:   # nobody came for the oldest: keep the stream order, drop the record
            self._wait_record(main, q.pop(0))
        self._prefetched = q

    def cancel_prefetch(self):
        """Drop a frozen branch queued by ``prefetch()`` that will not be consumed.  The prefetch stream's encoder reads the TSFormer's
        keep-mask pool, which the next encoder launch refills: the CURRENT stream is made to wait for the queued branch first, so no
        later launch -- a forward() of another batch, a direct ``model.tsformer(...)`` / ``DiscreteGraphLearning.forward`` call -- can
        refill the pool under a running kernel (that would make the dropout masks irreproducible)."""
        q, self._prefetched = self._prefetched, None
        for rec in (q or []):
            self._wait_record(torch.cuda.current_stream(), rec)

    @staticmethod
    def _wait_record(stream, rec):
        """order `stream` behind everything a queued frozen branch launched (its encoder and, on its own stream, its kNN prior)"""
        for k in ("done", "knn_done"):
            if rec.get(k) is not None:
                stream.wait_event(rec[k])

    def _take_prefetched(self, long_hist):
        q = self._prefetched
        if not q:
            return None
        key = self._batch_key(long_hist)
        if self._alias:
            key = self._alias.pop(key, key)
            self._alias.clear()
        main = torch.cuda.current_stream()
        stale = [rec for rec in q if rec["training"] != self.training]
        if stale:
            # queued in the other mode (a branch announced at the end of a training epoch, then validation): it would pin its encoder
            # output and kNN buffers (160 MB and more each at PEMS04) and add an event wait to every forward of the whole eval loop
            for rec in stale:
                self._wait_record(main, rec)
            q[:] = [rec for rec in q if rec["training"] == self.training]
            if not q:
                self._prefetched = None
                return None
        for idx, rec in enumerate(q):
            if rec["key"] == key and rec["training"] == self.training:
                for skipped in q[:idx]:              # announced but never consumed: keep the stream order, drop them
                    self._wait_record(main, skipped)
                del q[:idx + 1]
                if not q:
                    self._prefetched = None
                return rec
        # not an announced batch (e.g. the very first step): the inline encoder launch refills the shared keep-mask pool, so it is ordered
        # behind the queued branches -- which stay queued for the batches they were announced for
        for rec in q:
            self._wait_record(main, rec)
        return None

    def _prefetch_stream(self, dev):
        return self._side_stream(dev, "prefetch")

    def _next_seed(self):
        self._seed_ctr += 1
        return (torch.initial_seed() * 2654435761 + self._seed_ctr * 40503) & ((1 << 63) - 1)

    def _trainable(self):
        items = [("be." + k, v) for k, v in self.backend.trainable_native().items()]
        items += [("dgl." + k, v) for k, v in self.discrete_graph_learning.trainable_native().items()]
        # the DGL fc weight (87 MB at PEMS04, 98 % of the gradient bytes) goes last: its gradient is finished early in the
        # backward and is all-reduced as its own chunk while the rest of the backward runs
        items.sort(key=lambda kv: kv[0] == "dgl.fc_w")
        return items

    def _param_structs(self, bf, full):
        """(StepDglParams, StepGwnetParams) of the CURRENT parameter / buffer storage for the library calls.  Filling them is a walk over
        ~130 tensors (0.1-0.2 ms of host time, twice per step); once the parameters live in the flat buffer (flatten_parameters(), which
        FusedAdamClip calls) their addresses do not move, so the structs are kept and re-checked against a few addresses per call --
        .to() / .cuda() / load of another flat buffer drop them (_apply, flatten_parameters)."""
        dgl, be = self.discrete_graph_learning, self.backend
        ov = self._precision_override          # tools/precision_split.py: {"dgl": 0 | 1, "backend": 0 | 1} (which half of the step rounds its operands)
        if ov is not None:
            return fill_dgl_struct(dgl.native_tensors(full=full), ov.get("dgl", bf)), fill_gwnet_struct(be.native_tensors(), ov.get("backend", bf))
        if self._flat_param is None:
            return fill_dgl_struct(dgl.native_tensors(full=full), bf), fill_gwnet_struct(be.native_tensors(), bf)
        fcw = dgl.fc.weight if (dgl._shard is None or full) else dgl.fc_weight_slice
        key = (bf, full, self._flat_param.data_ptr(), fcw.data_ptr(), be.nodevec1.data_ptr(), be.bn[0].running_mean.data_ptr(),
               dgl.bn3.running_var.data_ptr(), be.end_conv_2.bias.data_ptr())
        c = self._struct_cache.get((bf, full))
        if c is None or c[0] != key:
            c = (key, fill_dgl_struct(dgl.native_tensors(full=full), bf), fill_gwnet_struct(be.native_tensors(), bf))
            self._struct_cache[(bf, full)] = c
        return c[1], c[2]

    def _grad_buffers(self, layout, dev):
        """-> (flat gradient buffer, name -> view, StepGwnetParams of the views, StepDglParams of the views).  With flattened parameters and no .grad left on the parameters (zero_grad(set_to_none=True), the training loop's normal
        state) ONE buffer and its ~100 views / two pointer structs are reused step after step: creating them took 0.7 ms of host time per
        backward (profiles/r05_t_host_sections_C2.log).  Any other situation -- gradients being accumulated over several backwards, no flat
        parameter buffer -- gets fresh ones, with autograd's usual accumulate semantics."""
        tr = self._trainable_list()
        reuse = self._flat_param is not None and tr[0].grad is None and tr[-1].grad is None and tr[len(tr) // 2].grad is None
        c = self._bwd_cache
        if reuse and c is not None and c["layout"] is layout and c["flat"].device == dev:
            return c["flat"], c["views"], c["gw"], c["dg"]
        flat = torch.empty(layout["total"], device=dev, dtype=torch.float32)
        views = {k: flat[o:o + n].view(shape) for k, (o, n, shape) in layout["items"].items()}
        gw = fill_gwnet_struct({k[3:]: v for k, v in views.items() if k.startswith("be.")})
        dg = fill_dgl_struct({k[4:]: v for k, v in views.items() if k.startswith("dgl.")})
        if reuse:
            self._bwd_cache = {"layout": layout, "flat": flat, "views": views, "gw": gw, "dg": dg}
        return flat, views, gw, dg

    def _module_key(self):
        """identity of everything a cached Parameter list depends on that does not go through _apply / load_state_dict:
        shard_time_slices() swaps the graph learner's fc parameter, user code may replace a sub-module"""
        dgl = self.discrete_graph_learning
        return (id(self.tsformer), id(self.backend), id(dgl), id(dgl._parameters.get("fc_weight_slice")),
                len(self._parameters), len(dgl._parameters), len(self.backend._parameters))

    def _trainable_list(self):
        t, key = self._trainable_cache, self._module_key()
        if t is not None and t[0] != key:
            # a direct discrete_graph_learning.shard_time_slices() or a replaced sub-module / Parameter: everything derived from the old
            # Parameter objects is stale (the kernels would compute with the new tensors while gradients were attached to the old ones)
            self._layout, self._zg_params, self._bwd_cache, self._struct_cache, t = None, None, None, {}, None
            if self._flat_param is not None:
                self._flat_param = self._flat_grad = None
        if t is None:
            t = self._trainable_cache = (key, [v for _, v in self._trainable()])
        return t[1]

    def _grad_layout(self):
        if self._layout is None:
            off, items, order = 0, {}, []
            norm_slot = None
            for k, v in self._trainable():
                if k == "dgl.fc_w":
                    # four spare floats in front of the fc weight: with time slices the squared norm of this rank's fc-slice gradient
                    # rides here through the all-reduce of everything before fc_w (the clip norm needs the other ranks' slices)
                    norm_slot = off
                    off += 4
                n = v.numel()
                items[k] = (off, n, tuple(v.shape))
                order.append(k)
                off += (n + 3) & ~3
            self._layout = {"items": items, "order": order, "total": off, "norm_slot": norm_slot}
        return self._layout

    def flatten_parameters(self):
        """Re-home every parameter that receives a gradient into ONE flat device buffer (same order and
        offsets as the flat gradient buffer), so the fused clip+Adam kernel updates the model in one pass.
        Call after the module is on its device; parameters keep their identity, names, shapes and values."""
        lay = self._grad_layout()
        dev = self.backend.nodevec1.device
        flat = torch.zeros(lay["total"], device=dev, dtype=torch.float32)
        for k, v in self._trainable():
            o, n, shape = lay["items"][k]
            flat[o:o + n].copy_(v.detach().reshape(-1))
            v.data = flat[o:o + n].view(shape)
        self._flat_param = flat
        self._struct_cache, self._bwd_cache, self._trainable_cache = {}, None, None
        return flat

    def enable_native_data_parallel(self, process_group=None, sync_module_states=True, shard_graph_learner=False, single_rank_collectives=False,
                                    collectives="auto"):
        """Average the flat gradient buffer over ranks inside backward (RCCL all-reduce of the flat buffer, in two asynchronous
        chunks; replaces DDP's bucketed reducer -- do not also wrap the module in DistributedDataParallel).  Like DDP's
        constructor, it first broadcasts rank 0's parameters and buffers (``sync_module_states``).
        ``single_rank_collectives``: issue every collective of the data-parallel path even in a group of ONE rank (the parameter
        broadcast, the chunked asynchronous all-reduce, the time-sliced graph learner's small sums with a single slice), so that
        the collective library, its stream and the event ordering against the step's three streams run on a one-GPU box
        (`bench.py --gpus 1 --force-process-group`); arithmetically a no-op.
        ``collectives``: "rccl" -- the step's collectives are RCCL C-API calls issued in stream order by libstep_hip
        (`step_amd.comm.NativeComm`: `step_grad_allreduce_begin / _join`, `step_comm_allreduce`; torch.distributed only carries the
        communicator's unique id and the initial parameter broadcast); "torch" -- `torch.distributed.all_reduce` (any backend; what
        the CPU tests run over gloo); "auto" (default) -- "rccl" when the group's backend is nccl (= RCCL) and the parameters live on a
        GPU, else "torch"."""
        import torch.distributed as dist
        self._process_group = process_group if process_group is not None else dist.group.WORLD
        self._min_world = 0 if single_rank_collectives else 1
        if self._comm is not None:
            self._comm.close()
            self._comm = None
        on_gpu = self.backend.nodevec1.is_cuda
        auto = collectives == "auto"
        if auto:
            collectives = "rccl" if (on_gpu and dist.get_backend(self._process_group) == "nccl") else "torch"
        if collectives not in ("rccl", "torch"):
            raise ValueError(f"collectives = {collectives!r}: expected 'auto', 'rccl' or 'torch'")
        if collectives == "rccl" and dist.get_world_size(self._process_group) > self._min_world:
            from .. import comm as _comm
            if not on_gpu:
                raise RuntimeError("collectives='rccl' needs the module on a GPU (move it first)")
            if not _comm.available() and not auto:
                raise RuntimeError("collectives='rccl': librccl could not be loaded into this process (STEP_RCCL_LIB names another copy)")
            usable = _comm.available()
            if auto and dist.get_world_size(self._process_group) > 1:
                # agree BEFORE anyone enters ncclCommInitRank: a rank whose librccl did not load would otherwise go on to the all-reduce
                # below while the others block in the communicator's rendezvous
                have = torch.tensor([1 if usable else 0], device=self.backend.nodevec1.device, dtype=torch.int32)
                dist.all_reduce(have, op=dist.ReduceOp.MIN, group=self._process_group)
                usable = bool(int(have.item()))
            try:
                if not usable:
                    raise RuntimeError("librccl could not be loaded on every rank")
                self._comm = _comm.NativeComm(self._process_group)
            except Exception as ex:          # noqa: BLE001
                if not auto:
                    raise
                # "auto" promised a working data-parallel step, not a particular transport: say so and stay on torch.distributed
                import warnings
                warnings.warn(f"step_amd: RCCL C-API communicator could not be created ({ex}); the step's collectives stay on torch.distributed")
                self._comm = None
            if auto and dist.get_world_size(self._process_group) > 1:
                # all ranks take the same transport: one rank without a communicator puts every rank on torch.distributed
                ok = torch.tensor([1 if self._comm is not None else 0], device=self.backend.nodevec1.device, dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self._process_group)
                if int(ok.item()) == 0 and self._comm is not None:
                    self._comm.close()
                    self._comm = None
        if self._comm is not None:
            # the all-reduce's stream: one that demonstrably overlaps with the compute stream (created AFTER RCCL's own streams exist)
            self._comm.use_side_stream(self._side_stream(self.backend.nodevec1.device, "comm"))
        if sync_module_states and dist.get_world_size(self._process_group) > self._min_world:
            # what DistributedDataParallel does when it wraps a module: every rank starts from rank 0's parameters and buffers
            src = dist.get_global_rank(self._process_group, 0)
            with torch.no_grad():
                for t in list(self.parameters()) + list(self.buffers()):
                    d = t.detach()
                    if d.is_contiguous():
                        dist.broadcast(d, src, group=self._process_group)
                    else:
                        c = d.contiguous()
                        dist.broadcast(c, src, group=self._process_group)
                        d.copy_(c)
        if shard_graph_learner and dist.get_world_size(self._process_group) > self._min_world:
            # every rank keeps one time slice of the graph learner's global branch and of fc.weight
            if self.matmul_precision != "bf16":
                raise ValueError("shard_graph_learner needs matmul_precision = 'bf16' (the sliced backward uses the fused BatchNorm backward)")
            self.discrete_graph_learning.shard_time_slices(dist.get_rank(self._process_group), dist.get_world_size(self._process_group))
            self._layout = None
            self._flat_param = None
            self._zg_params = None
            self._struct_cache, self._bwd_cache, self._trainable_cache = {}, None, None

    def _reduce_begin(self, chunk):
        """Start the sum of one contiguous chunk of the flat gradient buffer over the data-parallel group (RCCL all-reduce on
        the collective's own stream, ordered after the kernels already queued); returns the pending work handles."""
        if self._process_group is None:
            return []
        import torch.distributed as dist
        if dist.get_world_size(self._process_group) <= self._min_world or chunk.numel() == 0:
            return []
        if self._comm is not None:
            # RCCL C API: the mean all-reduce runs on the communicator's own stream behind what is queued here; _reduce_finish joins it
            self._comm.grad_allreduce_begin(chunk)
            return [None]
        return [dist.all_reduce(chunk, group=self._process_group, async_op=True)]

    def _sum_over_ranks(self, t):
        """in-place sum of a small device tensor over the data-parallel group, ordered on the current stream"""
        import torch.distributed as dist
        if self._process_group is not None and dist.get_world_size(self._process_group) > self._min_world:
            timed = self._reduce_wait_ms is not None and t.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self._comm is not None:
                self._comm.allreduce_(t)           # in stream order, no host wait
            else:
                dist.all_reduce(t, group=self._process_group)
            if timed:
                e1.record()
                self._small_events.append((e0, e1, t.numel() * t.element_size()))

    def _reduce_finish(self, flat, pending, reduced=None):
        """Wait for the chunks and turn the sums into means.  With ``_reduce_wait_ms`` set to a list (bench.py), the time the
        compute stream spends waiting for the collectives is recorded with events (read after a synchronize)."""
        if not pending:
            return
        import torch.distributed as dist
        timed = self._reduce_wait_ms is not None and flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._comm is not None:
            self._comm.grad_allreduce_join()       # (ncclAvg: the chunks already hold means)
        else:
            for w in pending:
                w.wait()
        if timed:
            e1.record()
            self._reduce_events.append((e0, e1))
        if self._comm is None:
            (flat if reduced is None else reduced).mul_(1.0 / dist.get_world_size(self._process_group))

    def collect_reduce_waits(self):
        """ms the compute stream waited for the gradient all-reduce in each backward since the last call (needs a synchronize)."""
        out = [a.elapsed_time(b) for a, b in self._reduce_events]
        self._reduce_events = []
        if self._reduce_wait_ms is not None:
            self._reduce_wait_ms.extend(out)
        return out

    def collect_small_collectives(self):
        """{"per_step": n, "exposed_ms_per_step": t, "bytes": [...]} of the blocking sums issued since collect_reduce_waits() was
        armed (needs a synchronize): each is timed on the stream that waits for it."""
        ev, self._small_events = self._small_events, []
        steps = max(len(self._reduce_wait_ms or []), 1)
        if not ev:
            return None
        ms = [a.elapsed_time(b) for a, b, _ in ev]
        return {"per_step": len(ev) / steps, "exposed_ms_per_step": float(sum(ms)) / steps,
                "bytes": sorted({int(n) for _, _, n in ev})}

    def _reduce_flat_grads(self, flat):
        self._reduce_finish(flat, self._reduce_begin(flat), flat)

    def zero_grad(self, set_to_none=True):
        """also forgets the flat gradient buffer of the last native backward (so `model.zero_grad(); loss.backward(); opt.step()`
        loops work with FusedAdamClip like `opt.zero_grad()` ones)"""
        if set_to_none:
            # nn.Module.zero_grad walks the module tree for its parameters (0.3 ms of host time per step); the Parameter objects
            # of the tree do not change between steps, so the list is kept (dropped by _apply / load_state_dict / time slicing)
            # and re-checked against the identity of the sub-modules and of the graph learner's slice parameter, which is what
            # shard_time_slices() / a replaced sub-module change without going through _apply / load_state_dict)
            key = self._module_key()
            ps = self._zg_params
            if ps is None or self._zg_key != key:
                ps = self._zg_params = list(self.parameters())
                self._zg_key = key
            for q in ps:
                q.grad = None
        else:
            super().zero_grad(set_to_none=False)
        self._flat_grad = None
        self._backward_count = 0

    def train(self, mode=True):
        if bool(mode) != self.training and self._prefetched:
            self.cancel_prefetch()          # a branch queued for the other mode will not be consumed (see _take_prefetched)
        return super().train(mode)

    def _apply(self, fn, recurse=True):
        self._zg_params = None
        self._struct_cache, self._bwd_cache, self._trainable_cache = {}, None, None
        out = super()._apply(fn, recurse)
        flat = self._flat_param
        if flat is not None:
            # (see TSFormer._apply) parameters re-homed by .cuda() / .to() / .double(): the flat buffer is stale
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            if not all(lo <= v.data_ptr() < hi and v.dtype == flat.dtype for _, v in self._trainable()):
                self._flat_param = None
                self._flat_grad = None
        return out

    def load_state_dict(self, *a, **kw):
        self._zg_params = None
        self._struct_cache, self._bwd_cache, self._trainable_cache = {}, None, None
        return super().load_state_dict(*a, **kw)

    # ------------------------------------------------------------------ forward
    def forward(self, history_data, long_history_data, future_data, batch_seen, epoch, **kwargs):
        if not history_data.is_cuda:
            raise RuntimeError("step_amd.STEP runs only on an AMD GPU: libstep_hip has no CPU fallback")
        B, _, N, _ = history_data.shape
        if self._noise_override is not None:
            u = self._noise_override.to(history_data.device).float().contiguous()
        elif self.gumbel_noise == "torch_cpu":
            u = torch.rand(B, N * N, 2).to(history_data.device)
        else:
            u = None
        params = self._trainable_list()
        pred, theta, adj_knn = _StepFunction.apply(self, history_data, long_history_data, u, *params)
        if epoch is not None:
            gsl_coefficient = 1 / (int(epoch / 6) + 1)
        else:
            gsl_coefficient = 0
        return pred, theta, adj_knn, gsl_coefficient
