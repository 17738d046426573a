"""CPU-side checks: the C-ABI library loads and exports every symbol include/step_hip.h declares, the
host-side module surface matches the reference's state_dict, the product path refuses to run without a
GPU (no silent fallback), and the data-parallel gradient reduction is correct on a 2-process gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from step_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "step_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(step_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _lib.lib()                          # loads the .so, resolves the ctypes signatures
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/step_hip.h but not exported"
    assert set(_lib.exported_symbols()) == declared, set(_lib.exported_symbols()) ^ declared
    assert lib.step_abi_version() == _lib.ABI_VERSION == 10


def test_error_reporting_without_compute():
    from step_amd import _lib
    lib = _lib.lib()
    rc = lib.step_gemm(None, None)
    assert rc != 0
    assert b"null descriptor" in lib.step_last_error()
    with pytest.raises(RuntimeError, match="null descriptor"):
        _lib.check(rc, "step_gemm")


def _tiny_model():
    from tests.test_gpu_step import build_native  # builds on CPU, moves to cuda only if asked
    g = load_golden("step_tiny")
    from step_amd import STEP
    N, L, T, B, k, ep, tr = [int(x) for x in g["meta"]]
    targs = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=L / 12,
                 mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="forecasting")
    bargs = dict(num_nodes=N, support_len=2, dropout=0.3, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=2, out_dim=12,
                 residual_channels=32, dilation_channels=32, skip_channels=256, end_channels=512, kernel_size=2, blocks=4, layers=2)
    data = np.zeros((T, N, 3), dtype=np.float32)
    return STEP("SYNTH", None, targs, bargs, dict(dataset_name="SYNTH", k=k, input_seq_len=12, output_seq_len=12, data=data,
                                                  train_length=T, tsformer_tokens=L // 12)), g


def test_state_dict_matches_reference_and_checkpoint_roundtrip(tmp_path):
    model, g = _tiny_model()
    sd = {k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}
    assert set(model.state_dict().keys()) == set(sd.keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    # pre-trained TSFormer checkpoint format of the reference: {"model_state_dict": ...} (step.py:31-32)
    ck = tmp_path / "TSFormer_SYNTH.pt"
    torch.save({"model_state_dict": {k[len("tsformer."):]: v for k, v in sd.items() if k.startswith("tsformer.")}}, ck)
    model.pre_trained_tsformer_path = str(ck)
    model.load_pre_trained_model()
    assert all(not p.requires_grad for p in model.tsformer.parameters())
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    nograd = set(str(s) for s in g["meta.nograd"])
    assert nograd <= trainable                      # present and trainable, but never touched by backward


def test_no_cpu_fallback():
    model, g = _tiny_model()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(history_data=g["in.hist"], long_history_data=torch.zeros(2, 96, 20, 3), future_data=None, batch_seen=0, epoch=1)


def test_grad_layout_covers_used_parameters():
    model, g = _tiny_model()
    lay = model._grad_layout()
    names = set(lay["order"])
    assert "be.gconv_w.7" not in names and "be.bn_w.7" not in names          # dead layer (model.py:202-213)
    assert "be.gconv_w.6" in names and "dgl.fc_w" in names
    offs = sorted((o, n) for o, n, _ in lay["items"].values())
    for (o1, n1), (o2, _) in zip(offs, offs[1:]):
        assert o1 + n1 <= o2
    assert all(o % 4 == 0 for o, _ in offs)


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tests.test_abi_and_host import _tiny_model
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
model, g = _tiny_model()
torch.manual_seed(100 + rank)                      # ranks start from different parameters and BatchNorm statistics ...
with torch.no_grad():
    for t in list(model.parameters()) + [b for b in model.buffers() if b.is_floating_point()]:
        t.add_(torch.randn_like(t))
model.enable_native_data_parallel()                # ... and leave with rank 0's, like a DistributedDataParallel wrap
state = torch.cat([t.detach().double().reshape(-1) for t in list(model.parameters()) + list(model.buffers())])
both = [torch.empty_like(state) for _ in range(world)]
dist.all_gather(both, state)
assert torch.equal(both[0], both[1]) and torch.equal(both[rank], state)
lay = model._grad_layout()
flat = torch.arange(lay["total"], dtype=torch.float32) * (rank + 1)
model._reduce_flat_grads(flat)
want = torch.arange(lay["total"], dtype=torch.float32) * (sum(range(1, world + 1)) / world)
assert torch.allclose(flat, want), (flat[:5], want[:5])
# the backward's chunked form: the fc weight gradient (last in the layout) first, the rest afterwards
assert lay["order"][-1] == "dgl.fc_w"
fo, fn, _ = lay["items"]["dgl.fc_w"]
flat = torch.arange(lay["total"], dtype=torch.float32) * (rank + 1)
pending = model._reduce_begin(flat[fo:fo + fn])
pending += model._reduce_begin(flat[:fo])
model._reduce_finish(flat, pending)
assert torch.allclose(flat[:fo + fn], want[:fo + fn])
# weak-scaling shard check: per-rank window streams differ
import numpy as np
ts = np.random.default_rng(1234 + rank).integers(100, 1000, size=8)
gathered = [None] * world
dist.all_gather_object(gathered, ts.tolist())
assert gathered[0] != gathered[1]
# bench.py's exchange step of the pre-training config (C3)
import bench
ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
ps[0].grad = torch.full((3, 5), float(rank + 1)); ps[1].grad = torch.arange(7.0) * (rank + 1)      # ps[2] has no gradient
class _M:            # (a) gradients cloned by autograd: the flatten / scatter-back form
    _flat_grad = torch.zeros(4)
bench.average_grads(_M, ps, world)
assert torch.equal(ps[0].grad, torch.full((3, 5), 1.5)) and torch.equal(ps[1].grad, torch.arange(7.0) * 1.5) and ps[2].grad is None
_M._flat_grad = torch.zeros(24)          # (b) gradients that ARE views of the flat buffer (what the native backward leaves): one all-reduce
ps[0].grad = _M._flat_grad[:15].view(3, 5); ps[1].grad = _M._flat_grad[16:23]
ps[0].grad.fill_(float(rank + 1)); ps[1].grad.copy_(torch.arange(7.0) * (rank + 1))
bench.average_grads(_M, ps, world)
assert torch.equal(ps[0].grad, torch.full((3, 5), 1.5)) and torch.equal(ps[1].grad, torch.arange(7.0) * 1.5) and ps[2].grad is None
# time slices of the graph learner (SURVEY.md 8(f) row 2), host side: bounds, the slice parameter, state_dict keys, the gather
from step_amd.step_arch.discrete_graph_learning import DiscreteGraphLearning as DGL
try:
    model.matmul_precision = "bf16"
    model.enable_native_data_parallel(shard_graph_learner=True)          # step_tiny: 102 conv2 columns -> slices too short
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "shorter than 128" in str(e)
torch.manual_seed(7)
dgl = DGL("X", 3, 12, 12, data=np.random.default_rng(3).standard_normal((700, 6, 3)).astype(np.float32), train_length=600)
keys = set(dgl.state_dict())
full0 = dgl.fc.weight.detach().clone()
T2 = 600 - 18
assert DGL.slice_bounds(T2, world) == [0, 291, 582] and DGL.slice_bounds(13581, 8)[-1] == 13581
sh = dgl.shard_time_slices(rank, world)
assert (sh["a"], sh["b"], sh["Ts"], sh["own1"]) == ((0, 291, 309, 291) if rank == 0 else (291, 582, 309, 300))
assert sh["count1"] == 6 * 591 and sh["count2"] == 6 * 582 and tuple(dgl._series_slice.shape) == (6, 309)
assert torch.equal(dgl._series_slice, dgl._series_nt[:, sh["a"]:sh["b"] + 18])
assert set(dgl.state_dict()) == keys and not dgl.fc.weight.requires_grad and dgl.fc_weight_slice.requires_grad
assert dgl.native_tensors()["fc_w"] is dgl.fc_weight_slice and tuple(dgl.fc_weight_slice.shape) == (100, 16 * 291)
assert torch.equal(dgl.fc_weight_slice.view(100, 16, 291), full0.view(100, 16, T2)[:, :, sh["a"]:sh["b"]])
with torch.no_grad():
    dgl.fc_weight_slice.add_(rank + 1.0)                                  # "training": every rank moves its slice
    dgl.fc.weight.zero_()
dgl._slice_dirty = True                                                   # what the native backward sets
try:
    dgl.state_dict()
    raise SystemExit("expected RuntimeError: stale fc.weight")
except RuntimeError as e:
    assert "gather_fc_weight" in str(e)
dgl.gather_fc_weight()
assert not dgl._slice_dirty
want = full0.view(100, 16, T2).clone()
want[:, :, :291] += 1.0; want[:, :, 291:] += 2.0
assert torch.equal(dgl.fc.weight.view(100, 16, T2), want)
sd_full = dgl.state_dict()
with torch.no_grad():
    dgl.fc_weight_slice.zero_()
dgl.load_state_dict(sd_full)                                              # reference-layout checkpoints load into a sharded module: the slice is re-cut
assert torch.equal(dgl.fc_weight_slice.view(100, 16, 291), want[:, :, sh["a"]:sh["b"]])
assert dgl.native_tensors(full=True)["fc_w"] is dgl.fc.weight                # the unsharded evaluation path reads the gathered matrix
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_data_parallel_flat_allreduce_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):              # the rendezvous itself (a free port, gloo's full-mesh connect) gets one retry
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]
        if all(p.returncode == 0 for p in procs):
            break
        if attempt == 0 and not any("AssertionError" in o for o in outs):
            continue
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o


def _null_call_names():
    import ctypes
    from step_amd import _lib
    return sorted(n for n, (res, args) in _lib._SIGS.items()
                  if res is ctypes.c_int and args and n not in ("step_abi_version",))


@pytest.mark.parametrize("name", _null_call_names())
def test_every_entry_point_rejects_null_arguments_before_touching_the_device(name):
    """Error behaviour of the boundary (include/step_hip.h): a call with NULL buffers / zero sizes returns STEP_ERR_ARG (1) and
    leaves a message naming the operation in step_last_error(); nothing is launched, so this runs without a GPU."""
    import ctypes
    from step_amd import _lib
    lib = _lib.lib()
    res, argtypes = _lib._SIGS[name]
    zero = []
    for t in argtypes:
        if t in (ctypes.c_void_p,) or hasattr(t, "contents") or getattr(t, "_type_", None) not in ("i", "l", "f", "d", "I", "L", "Q", "q"):
            zero.append(None)
        elif t in (ctypes.c_float, ctypes.c_double):
            zero.append(0.0)
        else:
            zero.append(0)
    rc = getattr(lib, name)(*zero)
    assert rc == 1, (name, rc)
    msg = lib.step_last_error().decode()
    assert len(msg) > 8, (name, msg)
    with pytest.raises(RuntimeError, match="rc=1"):
        _lib.check(rc, name)


def test_workspace_size_queries_are_pure_host_functions():
    from step_amd import _lib
    lib = _lib.lib()
    N, T, B = 307, 13599, 8
    slices = (N * N + 4095) // 4096
    sel = B * (3 * 2048 + slices + 2) * 4                                      # selection state of the top-k
    assert lib.step_knn_workspace_bytes(B, N, 0) == sel                        # step_topk_mask alone
    # step_knn_graph: + the partial tiles of the split-K Gram product (9 tiles of 128 x 128 per sample, 72 in all: 11 splits to reach
    # 768 workgroups), 256-byte aligned
    assert lib.step_knn_workspace_bytes(B, N, 96 * 336) == ((sel + 255) & ~255) + 11 * B * N * N * 4
    assert lib.step_knn_workspace_bytes(1, 4096, 96 * 168) == ((1 * (3 * 2048 + 4096 + 2) * 4 + 255) & ~255)     # 1024 tiles: no split, no scratch
    assert lib.step_dgl_edges_saved_floats(B, N) == 2 * N * 100 + N * N + 2 * B * N * N
    assert lib.step_dgl_edges_work_floats(N) == N * N + 2 * N * 100
    assert lib.step_dgl_global_saved_floats(N, T) > N * 8 * (T - 9)           # at least the conv1 activations
    assert lib.step_dgl_global_work_floats(N, T, 1) >= lib.step_dgl_global_work_floats(N, T, 0)
    assert lib.step_gwnet_work_floats(B, N, 1) >= lib.step_gwnet_work_floats(B, N, 0) > 0
    assert lib.step_gwnet_saved_floats(B, N, 1) > lib.step_gwnet_saved_floats(B, N, 0) > 0
    assert lib.step_gwnet_saved_floats(2 * B, N, 0) > lib.step_gwnet_saved_floats(B, N, 0)
    assert lib.step_adam_work_floats() > 0


def test_mask_generator_seeded_matches_reference():
    """T3 (tsformer/mask.py:15-28): python random.shuffle of range(P), first 75 % masked, both lists sorted.  The golden
    lists were drawn by the reference's MaskGenerator after random.seed(4) (tools/make_golden.py run_pretrain_case)."""
    import random
    from step_amd.step_arch.tsformer import MaskGenerator
    g = load_golden("tsformer_pretrain_tiny")
    P = (g["in.x"].shape[1]) // 12
    random.seed(4)
    um, mk = MaskGenerator(P, 0.75)()
    assert um == g["in.unmasked"].tolist() and mk == g["in.masked"].tolist()
    assert sorted(um + mk) == list(range(P)) and len(mk) == int(P * 0.75)


def test_host_side_caches_follow_the_module():
    """The per-step host work that is cached (round 3: C1 was bound by it) must notice when its inputs change: the parameter list of
    zero_grad and of the TSFormer's packed-weight key, the WaveNet's tensor dictionary (BatchNorm buffers are REPLACED by module
    conversions), the process-wide streams."""
    from step_amd.step_arch import step as step_mod
    model, g = _tiny_model()
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)
    model.zero_grad()
    assert all(p.grad is None for p in model.parameters()) and model._zg_params is not None
    model._flat_grad, model._backward_count = object(), 3
    model.zero_grad()
    assert model._flat_grad is None and model._backward_count == 0              # the flat-gradient bookkeeping is reset too
    nt = model.backend.native_tensors()
    assert model.backend.native_tensors() is nt                                  # built once ...
    rm_before = nt["bn_rm.0"]
    model.double().float()                                                       # ... dropped by a conversion (buffers become new tensors)
    nt2 = model.backend.native_tensors()
    assert nt2 is not nt and nt2["bn_rm.0"] is model.backend.bn[0].running_mean and nt2["bn_rm.0"] is not rm_before
    assert model._zg_params is None and model.tsformer._plist is None            # and so are the parameter lists
    k1 = model.tsformer._pack_key(14)
    with torch.no_grad():
        next(model.tsformer.parameters()).add_(1.0)                              # an in-place update changes the key (version counter)
    assert model.tsformer._pack_key(14) != k1
    assert model.backend.trainable_native() is model.backend.trainable_native()
    assert set(model.backend.trainable_native()) < set(nt2)
    # streams are per process and device, not per model (hardware queues are few)
    assert step_mod._STREAMS is not None and isinstance(step_mod._STREAMS, dict)
    # parameters that appear WITHOUT _apply / load_state_dict: a direct shard_time_slices() creates the slice
    # parameter, a replaced sub-module brings new ones -- zero_grad must clear their gradients too
    model.zero_grad()
    model.discrete_graph_learning.fc_weight_slice = torch.nn.Parameter(torch.zeros(100, 16))      # what shard_time_slices() registers
    sl = model.discrete_graph_learning.fc_weight_slice
    sl.grad = torch.ones_like(sl)
    model.zero_grad()
    assert sl.grad is None
    from step_amd import TSFormer
    model.tsformer = TSFormer(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=14,
                              mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="forecasting")
    q = next(model.tsformer.parameters())
    q.grad = torch.ones_like(q)
    model.zero_grad()
    assert q.grad is None
    # a prefetch record can be dropped safely (no device in this test: no record, nothing to wait for)
    model.cancel_prefetch()
    assert model._prefetched is None


def test_loss_target_stride_detection():
    """step_loss_native reads the target in place when it is one feature of the batch tensor (`future[..., :1]`): the stride helper must
    accept exactly the layouts that are a constant element stride apart and nothing else."""
    from step_amd.step_loss import _flat_stride
    x = torch.zeros(3, 12, 23, 3)
    assert _flat_stride(x) == 1 and _flat_stride(x[..., :1]) == 3 and _flat_stride(x[..., 1:2]) == 3
    assert _flat_stride(x[..., :1].contiguous()) == 1
    assert _flat_stride(x[:, :, ::2, :1]) is None and _flat_stride(x[..., :2]) is None and _flat_stride(x.transpose(1, 2)) is None
    assert _flat_stride(x[:, :6, :, :1]) is None                                 # rows skipped between samples
    assert _flat_stride(x[:1, :, :, :1]) == 3                                    # a single sample: the batch stride does not matter
    assert _flat_stride(torch.zeros(5, 1)) == 1


def test_bench_sets_two_hardware_queues_for_single_process_runs_only():
    """bench.py puts GPU_MAX_HW_QUEUES into the environment before torch (the HIP runtime) is imported -- for single-process runs of the
    program itself, unless the caller set it: FOUR when the next batch's frozen branch is prefetched (the default at PEMS04 / PEMS07: main,
    side / aux and the prefetch stream with the persistent encoder need a queue each), TWO for the round-4 schedule (--no-prefetch, the
    configs without an ENC_SPLIT entry, the validation forward); multi-process runs (WORLD_SIZE > 1) keep the runtime's default."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = ("try:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit:\n    pass\nprint(os.environ.get('GPU_MAX_HW_QUEUES'))")
    code = "import os, sys, runpy\nsys.argv = ['bench.py', '--help']\n" + body
    as_module = "import os, sys; sys.argv = ['bench.py']; import bench; print(os.environ.get('GPU_MAX_HW_QUEUES'))"

    def argv(*a):
        return "import os, sys, runpy\nsys.argv = ['bench.py', " + ", ".join(repr(x) for x in a) + ", '--help']\n" + body

    def run(extra, code=code):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE")}
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout.strip().splitlines()[-1]
    assert run({}) == "4"                                                   # PEMS04: prefetch + persistent encoder
    assert run({}, argv("--no-prefetch")) == "2"
    assert run({}, argv("--config", "SYNTH_4096")) == "2"
    assert run({}, argv("--config", "STEP_PEMS07")) == "4"
    assert run({}, argv("--forward-only")) == "2"
    assert run({}, argv("--force-process-group", "--no-prefetch")) == "2"   # one-rank process group: the streams are probed for concurrency
    assert run({"GPU_MAX_HW_QUEUES": "3"}) == "3"                           # an explicit setting wins
    assert run({"WORLD_SIZE": "2"}) == "None"                               # real multi-rank runs: the runtime's default
    assert run({}, as_module) == "None"                                    # imported as a module (the tests): the process's settings stay


def test_bench_json_line_is_the_only_line_on_stdout():
    """The driver reads ONE JSON line from bench.py's stdout.  Text that C code left in the stdio buffer before the line (RCCL's version
    banner under a process group: on a pipe it is flushed at process exit, i.e. after the line) must leave through stderr, a rank other
    than 0 must write nothing to stdout at all, and whatever C code prints after the line goes to stderr too (bench.print_json_line,
    bench.stdout_is_for_the_json_line; seen on the GPU box: profiles/r05_zr_stdout_one_json_line_under_a_process_group.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys, bench\n"
            "libc = ctypes.CDLL(None)\n"
            "rank = int(sys.argv[1])\n"
            "bench.stdout_is_for_the_json_line(rank)\n"
            "libc.printf(b'banner through C stdio before the line\\n')\n"
            "if rank == 0:\n"
            "    bench.print_json_line({'metric': 'm', 'value': 1.5})\n"
            "libc.printf(b'banner through C stdio after the line\\n')\n")
    for rank, want in ((0, ['{"metric": "m", "value": 1.5}']), (1, [])):
        out = subprocess.run([sys.executable, "-c", code, str(rank)], cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.splitlines() == want, (rank, out.stdout)
        assert out.stderr.count("banner through C stdio") == 2, out.stderr[-500:]


def test_tsformer_flatten_parameters_keeps_the_state_dict_and_shares_one_buffer():
    """TSFormer.flatten_parameters() (host logic, no device needed): every parameter becomes a view into ONE f32 buffer laid out like the
    native backward's flat gradient buffer (parameters in `_pt_names` order, each at a multiple of 4 floats), values, `state_dict` keys and
    shapes unchanged, and `zero_grad()` rewinds the backward counter the fused optimizer checks."""
    import torch
    from step_amd import TSFormer
    torch.manual_seed(0)
    m = TSFormer(12, 1, 96, 4, 4, 0.1, 24, 0.75, 4, 1, mode="pre-train")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = m.flatten_parameters()
    after = m.state_dict()
    assert list(after) == list(before)
    assert all(torch.equal(after[k], before[k]) and after[k].shape == before[k].shape for k in before)
    off = 0
    for p in m._pt_params():
        assert p.data.data_ptr() == flat.data_ptr() + 4 * off and off % 4 == 0
        off += (p.numel() + 3) & ~3
    assert off == flat.numel()
    with torch.no_grad():
        flat.mul_(2.0)                                    # the buffer IS the parameters
    assert all(torch.equal(m.state_dict()[k], 2.0 * before[k]) for k in before)
    m._backward_count = 3
    m.zero_grad()
    assert m._backward_count == 0
    # re-homing the parameters (.double(), .to(another device)) drops the stale flat buffer; a no-op .to() keeps it
    m.to("cpu")
    assert m._flat_param is flat
    m.double()
    assert m._flat_param is None and m._flat_grad is None


def test_product_never_imports_the_oracle_or_the_reference():
    """The oracle (oracle/), the staged reference sources (oracle/_ref/, tools/stage_reference.sh) and /root/reference are test
    infrastructure: nothing under step_amd/ or include/ may import, load or execute them (only tests/, bench.py's cpu_baseline leg,
    __graft_entry__.smoke() and the golden-generating tools do)."""
    bad = []
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b|reference_loader|oracle/_ref|oracle\._ref|/root/reference|import_reference|step_oracle)")
    for base in ("step_amd", "include"):
        for dirpath, dirnames, files in os.walk(os.path.join(ROOT, base)):
            dirnames[:] = [d for d in dirnames if d not in ("build", "__pycache__")]
            for f in files:
                if not f.endswith((".py", ".h", ".hip", ".cpp")):
                    continue
                for n, line in enumerate(open(os.path.join(dirpath, f), errors="replace"), 1):
                    if pat.search(line) and "import" in line or ("CDLL" in line and "oracle" in line) or ("dlopen" in line and "oracle" in line):
                        bad.append(f"{os.path.relpath(os.path.join(dirpath, f), ROOT)}:{n}: {line.strip()}")
    assert not bad, bad


def test_f16_operand_fit_check_is_host_side():
    """pack-time half of the encoder's float16 range guard (step_arch/tsformer.py): which state_dict values float16 cannot hold"""
    from step_amd.step_arch.tsformer import TSFormer
    from tests import train_problem as TPb
    targs, _ = TPb.model_args(8, 480)
    torch.manual_seed(0)
    m = TSFormer(**targs)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert TSFormer.f16_operands_fit(sd)[0]
    sd["decoder.transformer_encoder.layers.0.linear1.weight"][0, 0] = 1e6          # the decoder is not on the forecasting path
    assert TSFormer.f16_operands_fit(sd)[0]
    sd["encoder.transformer_encoder.layers.2.linear2.weight"][3, 5] = -7.0e4
    fits, name, worst = TSFormer.f16_operands_fit(sd)
    assert not fits and name == "encoder.transformer_encoder.layers.2.linear2.weight" and worst == 7.0e4
    sd["encoder.transformer_encoder.layers.2.linear2.weight"][3, 5] = float("nan")
    assert not TSFormer.f16_operands_fit(sd)[0]


def test_stale_parameter_list_is_rebuilt_after_direct_resharding():
    """the cached Parameter list handed to autograd (and the gradient layout) follow a Parameter swapped by a DIRECT
    shard_time_slices() call, not only _apply / load_state_dict / enable_native_data_parallel()."""
    from tests import train_problem as TPb
    model = TPb.build_native(16, 96, 400, TPb.make_series(16, 600))
    before = list(model._trainable_list())
    whole = model.discrete_graph_learning.fc.weight
    assert any(a is whole for a in before)
    n_whole = model._grad_layout()["items"]["dgl.fc_w"][1]
    model.discrete_graph_learning.shard_time_slices(0, 2)
    after = model._trainable_list()
    sl = model.discrete_graph_learning.fc_weight_slice
    assert any(a is sl for a in after) and not any(a is whole for a in after)
    assert model._grad_layout()["items"]["dgl.fc_w"][1] == sl.numel() < n_whole


# ---------------------------------------------------------------------------------------------- step_amd.runner (host logic)
class _DummyBase:
    """the hooks of the reference's runner classes that native_runner() overrides, reduced to their contracts"""

    def __init__(self, cfg):
        self.model = torch.nn.Linear(2, 2)
        self.loss = lambda a, b, null_val=0.0: (a - b).abs().mean()
        self.metrics = {"MAE": lambda a, b, null_val=0.0: (a - b).abs().mean(), "MSE": lambda a, b, null_val=0.0: ((a - b) ** 2).mean()}
        self.meters = {}
        self.forward_features = [0, 1, 2]
        self.printed = []

    def build_train_data_loader(self, cfg):
        return torch.utils.data.DataLoader(cfg["dataset"], batch_size=2, shuffle=False)

    def build_val_data_loader(self, cfg):
        return None

    def build_test_data_loader(self, cfg):
        return None

    def select_input_features(self, data):
        return data[:, :, :, self.forward_features]

    def init_training(self, cfg):
        self.optim = torch.optim.SGD(self.model.parameters(), lr=0.1)

    def metric_forward(self, f, args):
        return f(*args, null_val=0.0)

    def update_epoch_meter(self, name, value, n=1):
        s, c = self.meters.get(name, (0.0, 0))
        self.meters[name] = (s + float(value) * n, c + n)

    def print_epoch_meters(self, kind):
        self.printed.append({k: s / c for k, (s, c) in self.meters.items()})

    def plt_epoch_meters(self, kind, step):
        pass

    def train_iters(self, epoch, it, data):          # base_tsf_runner.py:225-255 in miniature
        a, b = data
        loss = self.metric_forward(self.loss, [a, b])
        for name, f in self.metrics.items():
            self.update_epoch_meter("train_" + name, self.metric_forward(f, [a, b]).item())
        return loss


def test_native_runner_hooks_on_the_host():
    from step_amd.runner import LookaheadLoader, native_runner
    ds = torch.utils.data.TensorDataset(torch.arange(12.0).view(6, 2), torch.arange(12.0).view(6, 2) * 0.5)
    R = native_runner(_DummyBase)
    assert R.__name__ == "Native_DummyBase" and issubclass(R, _DummyBase)
    r = R({})
    loader = r.build_train_data_loader({"dataset": ds})
    assert isinstance(loader, LookaheadLoader) and len(loader) == 3 and loader.batch_size == 2
    assert r.build_val_data_loader({}) is None
    plain = list(torch.utils.data.DataLoader(ds, batch_size=2))
    got = list(loader)                      # CPU model: batches pass through, in order, nothing is announced
    assert len(got) == 3 and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, plain))
    assert loader.staged_batches == 3 and loader.prefetched_batches == 0
    # CPU tensors: metrics are not deferred (nothing to wait for), the meters behave like the base class's
    for it, d in enumerate(plain):
        r.train_iters(1, it, d)
    base = _DummyBase({})
    for it, d in enumerate(plain):
        base.train_iters(1, it, d)
    r.print_epoch_meters("train")
    base.print_epoch_meters("train")
    assert r.printed == base.printed and not r._pending
    r.init_training({})
    assert isinstance(r.optim, torch.optim.SGD)          # not a STEP module on a GPU / not Adam: left alone


def test_device_forecasting_dataset_reads_the_reference_files(tmp_path):
    import pickle
    from step_amd.runner import DeviceForecastingDataset
    series = np.random.default_rng(0).normal(size=(300, 5, 3)).astype(np.float32)
    with open(tmp_path / "data.pkl", "wb") as f:
        pickle.dump({"processed_data": series}, f)
    idx = {"train": [(t - 12, t, t + 12) for t in (12, 50, 200)], "valid": [(88, 100, 112)], "test": [(88, 100, 112)]}
    with open(tmp_path / "index.pkl", "wb") as f:
        pickle.dump(idx, f)
    ds = DeviceForecastingDataset(str(tmp_path / "data.pkl"), str(tmp_path / "index.pkl"), "train", 96)
    assert len(ds) == 3 and [int(ds[i]) for i in range(3)] == [12, 50, 200] and ds[0].dtype == torch.int64
    assert (ds.history_len, ds.future_len, ds.seq_len) == (12, 12, 96) and ds.index_only
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3)))
    assert batch.tolist() == [12, 50, 200] and batch.dtype == torch.int64
    # through worker processes, as the reference's configs ask for (NUM_WORKERS = 2, PIN_MEMORY = True: STEP_PEMS04.py:121-122); the device
    # copies of the series never travel to a worker
    ds._loaders[("cuda", 0)] = object()
    import pickle as _p
    assert _p.loads(_p.dumps(ds))._loaders == {}
    del ds._loaders[("cuda", 0)]
    assert [b.tolist() for b in torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2)] == [[12, 50], [200]]
    with pytest.raises(FileNotFoundError):
        DeviceForecastingDataset(str(tmp_path / "nope.pkl"), str(tmp_path / "index.pkl"), "train", 96)
    with pytest.raises(AssertionError):
        DeviceForecastingDataset(str(tmp_path / "data.pkl"), str(tmp_path / "index.pkl"), "all", 96)
