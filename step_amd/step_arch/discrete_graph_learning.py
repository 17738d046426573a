"""DiscreteGraphLearning with the reference's module surface (parameter holder + native calls).

Mirrors ``step/step_arch/discrete_graph_learning.py:48-168``: same constructor keywords
(``dataset_name, k, input_seq_len, output_seq_len``), same ``state_dict`` keys (``conv1, conv2,
fc, bn1..3, fc_mean`` (unused by the reference forward), ``fc_cat, fc_out``), same side effect of
reading ``datasets/<name>/data_in{in}_out{out}.pkl`` relative to the cwd (:57).

Differences that are deliberate (SURVEY.md headline 5):
  * the ``[N^2, N]`` one-hot gather matrices ``rel_rec``/``rel_send`` (:81-89) are never built;
    the edge MLP is evaluated in its factorised form on device;
  * sizes come from the data file instead of hard-coded per-dataset tables, so any dataset name
    works (optional ``num_nodes`` / ``train_length`` / ``data`` keywords override, which is how the
    4096-node stress config is built).  For the reference's dataset names the table values
    (:55-56) are used so ``fc`` has the reference's shape.
"""
import os
import pickle

import numpy as np
import torch
from torch import nn

from .. import _lib

# discrete_graph_learning.py:56 (train_length per dataset); num_nodes comes from the data file
_TRAIN_LENGTH = {"METR-LA": 23990, "PEMS04": 13599, "PEMS03": 15303, "PEMS07": 16513, "PEMS-BAY": 36482,
                 "PEMS08": 14284}


def _load_pkl(path):
    with open(path, "rb") as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding="latin1")


class DiscreteGraphLearning(nn.Module):
    def __init__(self, dataset_name, k, input_seq_len, output_seq_len, num_nodes=None, train_length=None, data=None,
                 tsformer_tokens=None):
        super().__init__()
        self.k = k
        if data is None:
            path = "datasets/" + dataset_name + "/data_in{0}_out{1}.pkl".format(input_seq_len, output_seq_len)
            data = _load_pkl(path)["processed_data"]
        data = np.asarray(data)
        if train_length is None:
            train_length = _TRAIN_LENGTH.get(dataset_name, int(data.shape[0] * 0.8))
        self.train_length = int(min(train_length, data.shape[0]))
        self.num_nodes = int(num_nodes if num_nodes is not None else data.shape[1])
        feats = torch.from_numpy(np.ascontiguousarray(data[:self.train_length, :self.num_nodes, 0])).float()
        # node-major copy [N, T] (the reference transposes on every forward, :130)
        self.register_buffer("_series_nt", feats.t().contiguous(), persistent=False)
        self.dim_fc = 16 * (self.train_length - 18)
        self.embedding_dim = 100
        self.conv1 = nn.Conv1d(1, 8, 10, stride=1)
        self.conv2 = nn.Conv1d(8, 16, 10, stride=1)
        self.fc = nn.Linear(self.dim_fc, self.embedding_dim)
        self.bn1 = nn.BatchNorm1d(8)
        self.bn2 = nn.BatchNorm1d(16)
        self.bn3 = nn.BatchNorm1d(self.embedding_dim)
        # fc_mean is dead in the reference forward (:142 commented out) but part of its state_dict
        tokens = tsformer_tokens if tsformer_tokens is not None else \
            {"METR-LA": 168, "PEMS-BAY": 168, "PEMS03": 336, "PEMS04": 336, "PEMS07": 168, "PEMS08": 336}.get(dataset_name, 168)
        self.dim_fc_mean = tokens * 96
        self.fc_mean = nn.Linear(self.dim_fc_mean, 100)
        self.fc_cat = nn.Linear(self.embedding_dim, 2)
        self.fc_out = nn.Linear(self.embedding_dim * 2, self.embedding_dim)

        self._shard = None

    @property
    def node_feats(self):          # [T, N], the reference attribute (:57)
        return self._series_nt.t()

    # ------------------------------------------------------------------ time slices over data-parallel ranks
    @staticmethod
    def slice_bounds(T2, world):
        """conv2-output columns [bounds[r], bounds[r+1]) of rank r: balanced, contiguous, covering [0, T2)."""
        return [(T2 * r) // world for r in range(world + 1)]

    def shard_time_slices(self, rank, world):
        """Keep only this rank's time slice of the global branch (conv1 -> bn1 -> conv2 -> bn2 -> fc, :131-134): series columns
        [a, b+18), conv2-output columns [a, b) and the matching columns of ``fc.weight`` as the new parameter ``fc_weight_slice``
        [100, 16*(b-a)] (``fc.weight`` itself stops being trained and is refreshed by ``gather_fc_weight()``).  The native
        forward/backward then run on the slice and exchange only the BatchNorm sums, the [N,100] partial fc product and 1328
        backward sums (step_dgl_global_*_shard in include/step_hip.h)."""
        assert self._shard is None, "already sharded"
        T2 = self.train_length - 18
        bounds = self.slice_bounds(T2, world)
        a, b = bounds[rank], bounds[rank + 1]
        if min(bounds[r + 1] - bounds[r] for r in range(world)) < 128:
            raise ValueError(f"time slices of {T2} conv2 columns over {world} ranks are shorter than 128 columns")
        self._shard = {"rank": rank, "world": world, "a": a, "b": b, "Ts": b - a + 18, "bounds": bounds,
                       "own1": (b - a) + (9 if rank == world - 1 else 0),
                       "count1": float(self.num_nodes) * (self.train_length - 9), "count2": float(self.num_nodes) * T2}
        self.register_buffer("_series_slice", self._series_nt[:, a:b + 18].contiguous(), persistent=False)
        w = self.fc.weight.detach().view(self.embedding_dim, 16, T2)[:, :, a:b].contiguous().view(self.embedding_dim, -1)
        self.fc_weight_slice = nn.Parameter(w.clone())
        self.fc.weight.requires_grad_(False)
        self._slice_dirty = False            # set by the native backward: fc.weight no longer holds what the slices hold
        return self._shard

    def shard_struct(self):
        sh = self._shard
        s = _lib.StepDglShard()
        s.own1, s.count1, s.count2 = sh["own1"], sh["count1"], sh["count2"]
        return s

    def gather_fc_weight(self, process_group=None):
        """Collective: write every rank's trained slice back into ``fc.weight`` (call on all ranks before ``state_dict()`` /
        checkpointing; the reference's state_dict layout is then complete again)."""
        import torch.distributed as dist
        sh = self._shard
        if sh is None:
            return
        T2 = self.train_length - 18
        full = self.fc.weight.data.view(self.embedding_dim, 16, T2)
        for r in range(sh["world"]):
            a, b = sh["bounds"][r], sh["bounds"][r + 1]
            buf = self.fc_weight_slice.data.view(self.embedding_dim, 16, b - a).clone() if r == sh["rank"] else \
                torch.empty(self.embedding_dim, 16, b - a, device=full.device, dtype=full.dtype)
            dist.broadcast(buf, dist.get_global_rank(process_group, r) if process_group is not None else r, group=process_group)
            full[:, :, a:b].copy_(buf)
        self._slice_dirty = False

    def refresh_fc_weight_slice(self):
        """after loading a (full) state_dict into a sharded module: re-cut this rank's slice from ``fc.weight``"""
        sh = self._shard
        if sh is not None:
            T2 = self.train_length - 18
            with torch.no_grad():
                self.fc_weight_slice.copy_(self.fc.weight.view(self.embedding_dim, 16, T2)[:, :, sh["a"]:sh["b"]].reshape(self.embedding_dim, -1))

    def state_dict(self, *args, **kwargs):
        if self._shard is not None and self._slice_dirty:
            # a checkpoint written now would silently carry the UNTRAINED fc.weight (the trained values live in the ranks' slices)
            raise RuntimeError("DiscreteGraphLearning is sharded in time slices and has been trained since the last gather: call "
                               "gather_fc_weight() on ALL ranks (e.g. in the runner's on_epoch_end) before state_dict() / save_model()")
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
        sd.pop(prefix + "fc_weight_slice", None)          # the reference's keys only (fc.weight carries the gathered matrix)
        return sd

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if prefix + "fc_weight_slice" in missing_keys:
            missing_keys.remove(prefix + "fc_weight_slice")
        if self._shard is not None:          # a (full) checkpoint was loaded into a sharded module: re-cut this rank's slice
            self.refresh_fc_weight_slice()
            self._slice_dirty = False

    def native_tensors(self, full=False):
        """full=True: the whole fc.weight even when sharded (the unsharded evaluation path)"""
        return {"conv1_w": self.conv1.weight, "conv1_b": self.conv1.bias, "conv2_w": self.conv2.weight,
                "conv2_b": self.conv2.bias, "fc_w": self.fc.weight if (self._shard is None or full) else self.fc_weight_slice, "fc_b": self.fc.bias,
                "bn1_w": self.bn1.weight, "bn1_b": self.bn1.bias, "bn1_rm": self.bn1.running_mean, "bn1_rv": self.bn1.running_var,
                "bn2_w": self.bn2.weight, "bn2_b": self.bn2.bias, "bn2_rm": self.bn2.running_mean, "bn2_rv": self.bn2.running_var,
                "bn3_w": self.bn3.weight, "bn3_b": self.bn3.bias, "bn3_rm": self.bn3.running_mean, "bn3_rv": self.bn3.running_var,
                "fc_out_w": self.fc_out.weight, "fc_out_b": self.fc_out.bias,
                "fc_cat_w": self.fc_cat.weight, "fc_cat_b": self.fc_cat.bias}

    def trainable_native(self):
        return {k: v for k, v in self.native_tensors().items() if not (k.endswith("_rm") or k.endswith("_rv"))}

    def forward(self, long_term_history, tsformer):
        """Standalone call with the reference's signature (discrete_graph_learning.py:113-168): long_term_history [B, L, N, C],
        tsformer = a step_amd.TSFormer -> (bernoulli_unnorm [B, N*N, 2], hidden_states [B, N, P, 96], adj_knn [B, N, N],
        sampled_adj [B, N, N]).  The native edge kernel produces theta = softmax(logits)[..., 0] rather than both logits; the first
        output is (log theta, log(1 - theta)), which equals the reference's logits up to a per-edge constant -- every use of them
        (softmax at step.py:72, the Gumbel sample) is invariant to it.  Forward only (``torch.no_grad()``): training goes through
        ``step_amd.STEP``.  Not available for a time-sliced (data-parallel sharded) module."""
        if not long_term_history.is_cuda:
            raise RuntimeError("step_amd.DiscreteGraphLearning runs only on an AMD GPU: libstep_hip has no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("step_amd.DiscreteGraphLearning.forward is forward-only (wrap the call in torch.no_grad()); training goes "
                               "through step_amd.STEP")
        if self._shard is not None:
            raise RuntimeError("standalone DiscreteGraphLearning.forward is not available for a time-sliced module")
        import ctypes
        L = _lib
        B, Lh, N, _ = long_term_history.shape
        dev = long_term_history.device
        P = Lh // 12
        hidden = tsformer(long_term_history[..., [0]])                      # [B, N, P, 96] (:139)
        T = self.train_length
        struct = fill_dgl_struct(self.native_tensors(), int(getattr(self, "matmul_precision", "f32") == "bf16"))
        gsaved = torch.empty(int(L.lib().step_dgl_global_saved_floats(N, T)), device=dev)
        gwork = torch.empty(int(L.lib().step_dgl_global_work_floats(N, T, 0)), device=dev)
        g = torch.empty(N, 100, device=dev)
        L.call("step_dgl_global_forward", L.ptr(self._series_nt), N, T, ctypes.byref(struct), int(self.training), 0.1,
               L.ptr(gsaved), L.ptr(gwork), L.ptr(g), L.stream())
        esaved = torch.empty(int(L.lib().step_dgl_edges_saved_floats(B, N)), device=dev)
        theta = torch.empty(B, N, N, device=dev)
        adj = torch.empty(B, N, N, device=dev)
        u = torch.rand(B, N * N, 2).to(dev)                                 # the reference draws on the host (:12)
        L.call("step_dgl_edges_forward", L.ptr(g), N, B, ctypes.byref(struct), L.ptr(u), 0, 0.5, L.ptr(esaved), L.ptr(theta), L.ptr(adj),
               L.stream())
        hb = hidden.reshape(B * N, P * 96).to(torch.bfloat16).contiguous()
        sim = torch.empty(B, N, N, device=dev)
        knn = torch.empty(B, N, N, device=dev)
        kwork = torch.empty(int(L.lib().step_knn_workspace_bytes(B, N, P * 96)), dtype=torch.uint8, device=dev)
        L.call("step_knn_graph", L.ptr(hb), None, B, N, P * 96, self.k * N, L.ptr(sim), L.ptr(knn), L.ptr(kwork), kwork.numel(), L.stream())
        if self.training:
            for m in (self.bn1, self.bn2, self.bn3):
                m.num_batches_tracked += 1
        th = theta.reshape(B, N * N).clamp(1e-30, 1.0)
        logits = torch.stack([th.log(), torch.log1p(-th.clamp(max=1 - 1e-7))], dim=-1)
        return logits, hidden, knn, adj


def fill_dgl_struct(tensors, gemm_bf16=False):
    s = _lib.StepDglParams()
    s.gemm_bf16 = int(gemm_bf16)
    for k, v in tensors.items():
        if v is None:
            continue
        assert v.is_cuda and v.is_contiguous() and v.dtype == torch.float32, k
        setattr(s, k, v.data_ptr())
    return s
