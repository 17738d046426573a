"""Data-parallel collectives of the native training step on RCCL's C API (libstep_hip: csrc/comm.cpp, include/step_hip.h).

`NativeComm` is one RCCL communicator per process created next to a `torch.distributed` process group: rank 0 makes the 128-byte
unique id, the group's own object broadcast carries it to the other ranks (the only use of torch.distributed on this path), every
rank calls `step_comm_init_rank`.  After that the step's gradient all-reduce and the small sums of the time-sliced graph learner are
plain `ncclAllReduce` calls in stream order, issued through ctypes: no work objects, no `Work.wait()`, no stream juggling on the
host -- through `torch.distributed` the two all-reduce calls of a step cost 0.45 ms of host time in front of the optimizer even
on one rank (profiles/r04_b_C2_rccl1_noshard_timeline.md).

The reference gets its data parallelism from easytorch wrapping the model in `DistributedDataParallel` (GPU_NUM > 1:
step/STEP_PEMS04.py:30, step/STEP_PEMS07.py:29,83,117); this is the same exchange -- one mean all-reduce of the gradients per step.
"""
import ctypes

import torch

from . import _lib


def available():
    """librccl can be loaded into this process (the copy torch has already loaded, when there is one)"""
    return bool(_lib.lib().step_comm_available())


class NativeComm:
    """RCCL communicator of the current device for the ranks of `process_group` (None: the default group)."""

    def __init__(self, process_group=None):
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise RuntimeError("NativeComm needs a GPU (RCCL); CPU groups keep using torch.distributed")
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        # every rank must leave this constructor the same way: a failure on rank 0 travels to the others as an empty id instead of
        # leaving them in the broadcast, and the finished communicator is checked with one collective of each kind before it is used
        ident, err = None, None
        if self.rank == 0:
            try:
                buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
                _lib.call("step_comm_unique_id", buf)
                ident = buf.raw
            except Exception as ex:          # noqa: BLE001
                err = ex
        box = [ident]
        if self.world > 1:
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            dist.broadcast_object_list(box, src=src, group=process_group)
        if box[0] is None:
            raise RuntimeError(f"rank 0 could not make an RCCL unique id{'' if err is None else f' ({err})'}")
        ident = ctypes.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
        h = ctypes.c_void_p()
        _lib.call("step_comm_init_rank", ident, self.world, self.rank, ctypes.byref(h))
        self._h = h
        self.version = int(_lib.lib().step_comm_version())
        self._side = None
        self.self_check()

    def self_check(self):
        """one mean all-reduce (f32), one sum (f64) and one broadcast over the new communicator against their closed forms; raises on a
        wrong answer, so that a communicator that does not work is found here and not inside a backward pass"""
        w, r = self.world, self.rank
        a = torch.full((1024,), float(r + 1), device="cuda")
        b = torch.full((48,), float(r + 1), device="cuda", dtype=torch.float64)
        c = torch.full((256,), r + 7, device="cuda", dtype=torch.int64)
        self.allreduce_(a, average=True)
        self.allreduce_(b)
        self.broadcast_(c, root=0)
        torch.cuda.current_stream().synchronize()
        want_a, want_b = (w + 1) / 2.0, w * (w + 1) / 2.0
        if not (bool((a - want_a).abs().max() < 1e-5) and bool((b == want_b).all()) and bool((c == 7).all())):
            raise RuntimeError(f"RCCL communicator self-check failed on rank {r} of {w}: mean {float(a[0])} (want {want_a}), "
                               f"sum {float(b[0])} (want {want_b}), broadcast {int(c[0])} (want 7)")

    def use_side_stream(self, stream):
        """run the overlapped gradient all-reduce on `stream` (a torch.cuda.Stream the caller has checked to run concurrently with its
        compute stream, see step_arch.step._concurrent_stream); kept alive here"""
        _lib.call("step_comm_set_side_stream", self._h, ctypes.c_void_p(stream.cuda_stream))
        self._side = stream

    # ---- in stream order on the current stream
    def allreduce_(self, t, average=False):
        """t <- sum (or mean) of t over the ranks, in place, queued on the current stream"""
        code = {torch.float32: _lib.COMM_F32, torch.float64: _lib.COMM_F64, torch.uint8: _lib.COMM_U8}[t.dtype]
        _lib.call("step_comm_allreduce", self._h, _lib.ptr(t), t.numel(), code, int(average), _lib.stream())
        return t

    def broadcast_(self, t, root=0):
        v = t.view(torch.uint8) if t.dtype not in (torch.float32, torch.float64, torch.uint8) else t
        code = {torch.float32: _lib.COMM_F32, torch.float64: _lib.COMM_F64, torch.uint8: _lib.COMM_U8}[v.dtype]
        _lib.call("step_comm_broadcast", self._h, _lib.ptr(v), v.numel(), code, int(root), _lib.stream())
        return t

    # ---- the step's gradient all-reduce, overlapped with the rest of the backward
    def grad_allreduce_begin(self, chunk):
        """mean of a contiguous f32 chunk of the flat gradient buffer over the ranks, ordered behind the kernels queued so far,
        running on the communicator's own stream"""
        _lib.call("step_grad_allreduce_begin", self._h, _lib.ptr(chunk), chunk.numel(), _lib.stream())

    def grad_allreduce_join(self):
        """the current stream waits for every reduction begun since the last join"""
        _lib.call("step_grad_allreduce_join", self._h, _lib.stream())

    def close(self):
        if self._h is not None:
            _lib.lib().step_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # noqa: BLE001 -- interpreter shutdown
            pass
