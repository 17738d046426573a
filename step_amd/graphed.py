"""One STEP training iteration captured once into a hipGraph and replayed ("one call per step").

The reference's loop (``base_tsf_runner.py:225-255`` through easytorch's ``Runner.train_iters`` / ``backward``) issues a step as
``zero_grad -> forward -> loss -> backward -> clip_grad_norm_ -> optimizer.step`` from one Python thread.  For the native module
that is ~110 C-ABI launches plus ~25 stream / event calls per step: 2.5-3 ms of host time, which bounds the small configurations
(STEP_METR-LA: the device needs 1.9 ms) and leaves 6-7 us bubbles on the dependent chain wherever an event is recorded.
``GraphedTrainStep`` runs the same Python once under stream capture -- the three-stream fork / join of forward and backward
included -- and afterwards replays the recorded graph with ONE launch per step.

What a replay cannot take as launch arguments lives in a 24-byte device struct (``StepDynState``, include/step_hip.h) that the
``*_dyn`` entry points read at run time: the seed offset of the three random streams (encoder keep-mask pool, Gumbel noise, gcn
dropout), Adam's step count and learning rate, the loss's graph-term coefficient.  ``step_dyn_advance`` -- the first node of the
graph -- moves the seed offset and the step count on the device; learning rate and coefficient are written by the host when the
scheduler / the epoch change them (``set_lr``, ``set_epoch``).  Inputs are copied into static buffers before each replay.

Usage (a runner's ``train_iters`` replacement)::

    step = GraphedTrainStep(model, FusedAdamClip(model, ...), first_batch, scaler=(mean, std), epoch=1)
    for hist, long_hist, fut in loader:          # tensors of the example's shapes (or use step.hist / .long_hist / .fut directly)
        loss = step(hist, long_hist, fut)        # device scalar of THIS step, valid until the next call

Restrictions: fixed batch shape; single process (the gradient collectives of ``enable_native_data_parallel`` are not captured);
``FusedAdamClip`` as the optimizer; the warm-up iterations before the capture are real optimizer steps on the example batch.
"""
import ctypes
import os

import torch

from . import _lib
from .optim import FusedAdamClip
from .step_loss import step_loss_native


_KEEP = object()          # GraphedTrainStep.__call__(epoch=...): leave the epoch as it is


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch, scaler=(0.0, 1.0), epoch=1, null_val=0.0, warmup=3, seed=None):
        if not isinstance(optimizer, FusedAdamClip):
            raise TypeError("GraphedTrainStep drives step_amd.optim.FusedAdamClip (its step count and learning rate live on the device)")
        if model._process_group is not None:
            raise RuntimeError("GraphedTrainStep is single-process: the data-parallel collectives are not captured")
        hist, long_hist, fut = example_batch
        if not hist.is_cuda:
            raise RuntimeError("step_amd runs only on an AMD GPU: libstep_hip has no CPU fallback")
        queues = os.environ.get("GPU_MAX_HW_QUEUES")
        if queues is not None and queues.strip().isdigit() and int(queues) < 4:
            # ROCm 7.2: hipGraphLaunch of this graph (parallel branches: main / side / aux stream) crashes when the runtime was initialised
            # with fewer hardware queues than its default of four (1, 2, 3: segmentation fault; 4, 8, unset: fine --
            # profiles/r04_n_graph_replay_hw_queues.log) -- refuse instead
            raise RuntimeError(f"GraphedTrainStep needs the HIP runtime's default hardware queues (GPU_MAX_HW_QUEUES={queues} is set): "
                               "a replay of the captured graph crashes inside hipGraphLaunch with fewer than four")
        self.model, self.opt = model, optimizer
        self.mean, self.std = float(scaler[0]), float(scaler[1])
        self.null_val = float(null_val)
        self.hist, self.long_hist, self.fut = hist.clone(), long_hist.clone(), fut.clone()          # static inputs of the graph
        dev = hist.device
        self.dyn = torch.zeros(3, dtype=torch.int64, device=dev)          # StepDynState (24 bytes)
        self._host = _lib.StepDynState()
        self._host.seed_xor = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & ((1 << 64) - 1) if seed is None else int(seed)
        self._host.adam_step = int(optimizer.step_count)
        self._host.lr = float(optimizer.param_groups[0]["lr"])
        self._host.gsl_coef = 1.0 / (int(epoch / 6) + 1) if epoch is not None else 0.0               # step.py:68-69
        self._epoch = epoch
        self._push()
        model._dyn = model.tsformer._dyn = optimizer.dyn = self.dyn
        model.cancel_prefetch()
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):          # fills every cache the capture must not touch (packed weights, event pools, ...)
                    self._iteration()
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._iteration()
        except BaseException:
            self.close()
            raise
        self.replays = 0

    # ------------------------------------------------------------------ the captured iteration
    def _iteration(self):
        _lib.call("step_dyn_advance", _lib.ptr(self.dyn), _lib.stream())
        self.opt.zero_grad(set_to_none=True)
        pred, theta, knn, _ = self.model(history_data=self.hist, long_history_data=self.long_hist, future_data=None, batch_seen=0,
                                         epoch=self._epoch)
        # target-feature selection + inverse scaling as the runner does (step_runner.py:86-92, base_tsf_runner.py:240-250); the graph
        # term's coefficient is read from the device state
        loss = step_loss_native(pred[..., :1], self.fut[..., :1], theta, knn, self.dyn, null_val=self.null_val, rescale=(self.mean, self.std))
        loss.backward()
        self.opt.step()
        return loss

    def _push(self):
        """host mirror -> device, the whole struct (initial write only: afterwards seed and step count belong to the device)"""
        buf = torch.frombuffer(bytearray(bytes(self._host)), dtype=torch.int64).clone()
        self.dyn.copy_(buf.to(self.dyn.device))

    def _set_f32(self, index, value):
        """one float field of the device struct (3: lr, 4: gsl_coef), written in stream order -- no synchronisation, replays in flight keep theirs"""
        self.dyn.view(torch.float32)[index:index + 1].fill_(float(value))

    # ------------------------------------------------------------------ per-step API
    def __call__(self, hist=None, long_hist=None, fut=None, epoch=_KEEP):
        # a scheduler (the reference's MultiStepLR, STEP_PEMS04.py:98-102) edits param_groups[0]["lr"] on the host: follow it here rather than
        # rely on the caller to remember set_lr() -- one float compare per replay; the epoch (graph-term coefficient) can ride along
        self.set_lr(self.opt.param_groups[0]["lr"])
        if epoch is not _KEEP:
            self.set_epoch(epoch)
        if hist is not None:
            self.hist.copy_(hist, non_blocking=True)
        if long_hist is not None:
            self.long_hist.copy_(long_hist, non_blocking=True)
        if fut is not None:
            self.fut.copy_(fut, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.loss

    def set_lr(self, lr):
        """the scheduler changed the learning rate (``torch.optim.lr_scheduler.MultiStepLR.step()`` on the optimizer, STEP_PEMS04.py:98-102)"""
        lr = float(lr)
        self.opt.param_groups[0]["lr"] = lr          # one source of truth: __call__ follows the optimizer's value
        if lr != self._host.lr:
            self._host.lr = lr
            self._set_f32(3, lr)

    def set_epoch(self, epoch):
        """coefficient of the graph term of step_loss, step.py:68-69"""
        coef = 1.0 / (int(epoch / 6) + 1) if epoch is not None else 0.0
        self._epoch = epoch
        if coef != self._host.gsl_coef:
            self._host.gsl_coef = coef
            self._set_f32(4, coef)

    def state(self):
        """the device state as a dict (synchronises)"""
        cur = self.dyn.cpu()
        h = _lib.StepDynState()
        ctypes.memmove(ctypes.addressof(h), cur.numpy().ctypes.data, 24)
        return {"seed_xor": int(h.seed_xor), "adam_step": int(h.adam_step), "lr": float(h.lr), "gsl_coef": float(h.gsl_coef)}

    def close(self):
        """hand the model and the optimizer back to eager stepping (the optimizer's host-side step count is brought up to date)"""
        if self.opt.dyn is self.dyn:
            try:
                self.opt.step_count = self.state()["adam_step"]
            except Exception:          # noqa: BLE001 -- closing must not fail on a lost device
                pass
            self.opt.dyn = None
        if self.model._dyn is self.dyn:
            self.model._dyn = None
            self.model.tsformer._dyn = None
