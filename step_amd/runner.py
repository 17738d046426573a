"""BasicTS runner hooks for the native STEP module: the pieces that make the schedule ``bench.py`` times reachable from the
reference's OWN training loop by config lines only ("missing" 3).

The reference's loop (easytorch ``Runner.train`` -> ``BaseTimeSeriesForecastingRunner.train_iters`` -> ``STEPRunner.forward``,
``basicts/runners/base_tsf_runner.py:225-255``, ``step/step_runner/step_runner.py:43-78``) never calls ``STEP.prefetch``, moves
14.9 MB of long history per window across PCIe (``step/step_data/forecasting_dataset.py:52-71``) and reads three metrics back to the
host every iteration (``metric_item.item()``, ``base_tsf_runner.py:253-254``: three device synchronisations per step).  The config
already has the slots to change all of that without touching the loop -- ``CFG.RUNNER`` and ``CFG.DATASET_CLS``
(``step/STEP_PEMS04.py:21-22``):

    from step_amd import STEP
    from step_amd.runner import native_runner, DeviceForecastingDataset
    CFG.RUNNER = native_runner(STEPRunner)            # look-ahead loader + STEP.prefetch, deferred meters, fused clip + Adam
    CFG.DATASET_CLS = DeviceForecastingDataset        # optional: index-only windows over a device-resident series (no PCIe)
    CFG.MODEL.ARCH = STEP

``native_runner(base)`` returns a subclass of the reference's runner class; every hook it overrides calls the reference's own
implementation and only changes WHERE the data lives and WHEN the host reads numbers back:

* ``build_train_data_loader`` / ``build_val_data_loader`` / ``build_test_data_loader`` wrap the loader the reference builds in a
  ``LookaheadLoader``: it holds one batch ahead, puts that batch on the device (pinned memory + an asynchronous copy on its own stream
  for host batches; one gather launch for index batches) and calls ``model.prefetch(next long history)`` BEFORE it hands out the
  current batch -- the frozen TSFormer + kNN prior of batch i + 1 then run next to the whole of step i ("Step schedule").
* the epoch meters are fed once per epoch (when ``print_epoch_meters`` needs them) from values that stayed on the device instead of
  ``.item()`` per iteration, and the three metrics of ``basicts/metrics`` come from one launch (``step_masked_metrics``) instead of
  ~30 element-wise ones; the printed averages are the same numbers.  The feature selections of ``step_runner.py:25-26,39-40`` become
  slices where the feature list is a contiguous range (advanced indexing copies an index tensor to the device: a blocking copy).
* ``init_training`` swaps the ``torch.optim.Adam`` + ``clip_grad_norm_`` pair the config asks for (``STEP_PEMS04.py:89-106``) for
  ``step_amd.optim.FusedAdamClip`` with the same hyper-parameters (one pass over the flat buffers); any other optimizer is left alone.

Nothing here imports the reference or easytorch: the base class is handed in by the config file, which lives in the reference tree.
"""
import os
import pickle

import torch
from torch.utils.data import Dataset

from .step_arch.step import STEP, DeviceWindowLoader, LongHistoryRef


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module) else model


class DeviceForecastingDataset(Dataset):
    """Drop-in for ``step/step_data/forecasting_dataset.py:8-80`` (same constructor arguments, same files, same length) whose samples
    are FORECAST ORIGINS: ``__getitem__`` returns the int64 scalar ``idx[1]`` of the reference's index triple, the DataLoader's default
    collate makes an int64 ``[B]`` tensor of them, and ``LookaheadLoader`` turns that into ``(future_data, history_data,
    LongHistoryRef)`` with one gather launch over the series, which is moved to the device once.  Windows whose long history would
    start before the series does are all-zero, like the reference's (``forecasting_dataset.py:66-67``)."""

    index_only = True

    def __init__(self, data_file_path, index_file_path, mode, seq_len):
        super().__init__()
        assert mode in ["train", "valid", "test"], "error mode"
        for p, what in ((data_file_path, "data"), (index_file_path, "index")):
            if not os.path.isfile(p):
                raise FileNotFoundError("BasicTS can not find {0} file {1}".format(what, p))
        with open(data_file_path, "rb") as f:
            self.data = torch.from_numpy(pickle.load(f)["processed_data"]).float()
        with open(index_file_path, "rb") as f:
            self.index = pickle.load(f)[mode]
        self.seq_len = int(seq_len)
        first = self.index[0]
        self.history_len, self.future_len = int(first[1] - first[0]), int(first[2] - first[1])
        if self.history_len != self.future_len:
            raise ValueError("DeviceForecastingDataset gathers history and future windows of one length (the STEP configs: 12 -> 12)")
        self._loaders = {}

    def __getstate__(self):          # DataLoader workers get the index and the host series, never the device copies
        d = dict(self.__dict__)
        d["_loaders"] = {}
        return d

    def __getitem__(self, index):
        idx = self.index[index]
        if idx[1] - idx[0] != self.history_len or idx[2] - idx[1] != self.future_len:
            raise ValueError(f"index entry {index} = {tuple(idx)} does not have the dataset's window lengths")
        return torch.tensor(int(idx[1]), dtype=torch.int64)

    def __len__(self):
        return len(self.index)

    def windows(self, origins, device):
        """int64 [B] forecast origins -> (future_data [B,H,N,C], history_data [B,H,N,C], LongHistoryRef [B,L,N,C]) on ``device``"""
        key = (device.type, device.index)
        if key not in self._loaders:
            self._loaders[key] = DeviceWindowLoader(self.data.to(device), self.seq_len, self.history_len)
        hist, long_ref, fut = self._loaders[key].batch(origins)
        return fut, hist, long_ref


class LookaheadLoader:
    """Iterates ``loader`` one batch ahead of its consumer (see the module docstring).  ``runner`` supplies the device
    (``to_running_device``) and the model; ``prefetch=False`` only stages the batches on the device."""

    def __init__(self, loader, runner, prefetch=True):
        self.loader, self.runner, self.prefetch = loader, runner, prefetch
        self.dataset = getattr(loader, "dataset", None)
        self._copy_stream = None
        self.staged_batches = 0
        self.prefetched_batches = 0

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):          # batch_size, sampler, ... of the wrapped loader (easytorch reads a few)
        return getattr(self.__dict__["loader"], name)

    def _device(self):
        p = next(_unwrap(self.runner.model).parameters())
        return p.device

    def _stage(self, batch):
        """host batch -> device batch, without making the host wait: (future, history, long history)"""
        dev = self._device()
        self.staged_batches += 1
        if torch.is_tensor(batch) and batch.dim() == 1 and getattr(self.dataset, "index_only", False):
            origins = batch if batch.is_pinned() or batch.is_cuda else batch.pin_memory()
            return self.dataset.windows(origins.to(dev, non_blocking=True), dev)
        if dev.type != "cuda":
            return batch
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        out = []
        with torch.cuda.stream(self._copy_stream):
            for t in batch:
                if torch.is_tensor(t) and not t.is_cuda:
                    # pinned batches (the reference's loader pins: PIN_MEMORY = True, STEP_PEMS04.py:122) are copied asynchronously; a pageable
                    # one goes as it is -- pinning it here would be one more pass over 119 MB of long history on the host
                    d = t.to(dev, non_blocking=t.is_pinned())
                    d.record_stream(main)
                    out.append(d)
                else:
                    out.append(t)
        done = torch.cuda.Event()
        done.record(self._copy_stream)
        main.wait_event(done)          # queued, not waited for on the host: kernels issued after this point see the batch
        return tuple(out) if isinstance(batch, (tuple, list)) else out[0]

    def _announce(self, staged):
        model = _unwrap(self.runner.model)
        if not (self.prefetch and isinstance(model, STEP) and model.prefetch_enabled and isinstance(staged, (tuple, list)) and len(staged) == 3):
            return
        long_hist = staged[2]
        if not (isinstance(long_hist, LongHistoryRef) or (torch.is_tensor(long_hist) and long_hist.is_cuda)):
            return
        ff = getattr(self.runner, "forward_features", None)
        model.prefetch(long_hist, channel=0 if ff is None else int(ff[0]))
        self.prefetched_batches += 1

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            if nxt is not None:
                self._announce(nxt)          # batch i + 1's frozen branch is queued before batch i is consumed
            yield cur


class _Deferred:
    """what a metric returns while the meters are deferred: ``.item()`` hands back the device value instead of waiting for it"""

    def __init__(self, value):
        self.value = value

    def item(self):
        return self


def native_runner(base, prefetch=True, defer_meters=True, fused_optimizer=True, encoder_workgroups=None, native_metrics=True):
    """-> subclass of ``base`` (the reference's ``STEPRunner``, or any ``BaseTimeSeriesForecastingRunner``) for ``CFG.RUNNER``.
    ``encoder_workgroups``: compute units of the persistent encoder launch next to the step (None: 160 at 307 nodes / 336 tokens, the
    measured optimum of config C2; 0: one workgroup per sequence) -- only used while batches are prefetched."""

    class NativeRunner(base):
        native_hooks = True

        def __init__(self, cfg):
            super().__init__(cfg)
            self._deferring = False
            self._pending = {}          # meter name -> [(device value, n), ...] since the last flush
            self._metric_cache = None

        # ---------------------------------------------------------------- loaders
        def _lookahead(self, loader, on):
            return LookaheadLoader(loader, self, prefetch=on) if loader is not None else None

        def build_train_data_loader(self, cfg):
            loader = self._lookahead(super().build_train_data_loader(cfg), prefetch)
            model = _unwrap(self.model)
            if prefetch and isinstance(model, STEP) and next(model.parameters()).is_cuda:
                n = encoder_workgroups
                if n is None:
                    n = 160 if (model.backend.num_nodes, int(model.tsformer.num_token)) == (307, 336) else 0
                model.tsformer.encoder_workgroups = int(n)
                model.prefetch_knn_stream = int(n) > 0          # (the look-ahead loader announces at the start of the step: bench.py's policy)
            return loader

        def build_val_data_loader(self, cfg):
            return self._lookahead(super().build_val_data_loader(cfg), False)

        def build_test_data_loader(self, cfg):
            return self._lookahead(super().build_test_data_loader(cfg), False)

        # ---------------------------------------------------------------- feature selection without a host-device synchronisation
        # ``data[:, :, :, [0, 1, 2]]`` (step_runner.py:25-26,39-40) builds an index tensor on the host and copies it to the device: a blocking
        # copy behind everything queued on the stream, twice per iteration.  For the feature lists the STEP configs use -- contiguous
        # ranges -- a slice selects the same values as a VIEW: no index tensor, no copy, and forward() sees the very tensor the loader
        # announced to STEP.prefetch.  Any other list goes through the reference's indexing (and ``alias_batch``).
        @staticmethod
        def _range_of(features):
            if features is None:
                return None
            f = [int(i) for i in features]
            return (f[0], f[0] + len(f)) if f and f == list(range(f[0], f[0] + len(f))) else None

        def select_input_features(self, data):
            ff = getattr(self, "forward_features", None)
            if isinstance(data, LongHistoryRef):
                return data if ff is None else data[:, :, :, list(ff)]
            r = self._range_of(ff)
            if r is not None and torch.is_tensor(data) and data.dim() == 4 and r[1] <= data.shape[3]:
                out = data[:, :, :, r[0]:r[1]]
                if r[0] != 0 and data.is_cuda and isinstance(_unwrap(self.model), STEP):
                    _unwrap(self.model).alias_batch(data, out, channel=r[0])          # (a view that starts at another channel: another address)
                return out
            out = super().select_input_features(data)
            if out is not data and torch.is_tensor(out) and torch.is_tensor(data) and data.is_cuda:
                model = _unwrap(self.model)
                if isinstance(model, STEP):
                    # advanced indexing made a copy: tell the module that it is the batch the loader announced (same values in the
                    # channel the TSFormer reads), so that forward() finds the prefetched branch
                    model.alias_batch(data, out, channel=0 if ff is None else int(ff[0]))
            return out

        def select_target_features(self, data):
            r = self._range_of(getattr(self, "target_features", None))
            if r is not None and torch.is_tensor(data) and data.dim() == 4 and r[1] <= data.shape[3]:
                return data[:, :, :, r[0]:r[1]]
            return super().select_target_features(data)

        # ---------------------------------------------------------------- optimizer
        def init_training(self, cfg):
            super().init_training(cfg)
            model = _unwrap(self.model)
            opt = getattr(self, "optim", None)
            if not (fused_optimizer and isinstance(model, STEP) and model is self.model and type(opt) is torch.optim.Adam
                    and next(model.parameters()).is_cuda):
                return          # (under a DistributedDataParallel wrap the reducer owns the .grad tensors: torch's optimizer stays)
            from .optim import FusedAdamClip
            g = opt.param_groups[0]
            if len(opt.param_groups) != 1 or g.get("amsgrad") or g.get("maximize"):
                return
            clip = getattr(self, "clip_grad_param", None) or {}
            if clip and float(clip.get("norm_type", 2.0)) != 2.0:
                return
            fused = FusedAdamClip(model, lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"],
                                  max_norm=clip.get("max_norm"), param_grads=False)
            sched = getattr(self, "scheduler", None)
            if sched is not None:
                s = cfg["TRAIN"].get("LR_SCHEDULER")
                self.scheduler = type(sched)(fused, **s["PARAM"])
            self.optim, self.clip_grad_param = fused, None          # the clip is part of the fused pass

        # ---------------------------------------------------------------- meters without a device synchronisation per iteration
        def train_iters(self, epoch, iter_index, data):
            self._deferring, self._metric_cache = defer_meters, None
            try:
                return super().train_iters(epoch, iter_index, data)
            finally:
                self._deferring, self._metric_cache = False, None

        _NATIVE_METRICS = {"masked_mae": 0, "masked_rmse": 1, "masked_mape": 2}

        def metric_forward(self, metric_func, args):
            if self._deferring and metric_func is not self.loss and native_metrics:
                # the three metrics of basicts/metrics the runner evaluates per iteration (base_tsf_runner.py:252-254), from one launch
                k = self._NATIVE_METRICS.get(getattr(metric_func, "__name__", None))
                nv = getattr(self, "null_val", 0.0)
                if (k is not None and str(getattr(metric_func, "__module__", "")).startswith("basicts.metrics") and len(args) == 2
                        and torch.is_tensor(args[0]) and args[0].is_cuda and args[0].dtype == torch.float32 and torch.is_tensor(args[1])
                        and args[1].dtype == torch.float32 and args[0].shape == args[1].shape and nv == nv):
                    c = self._metric_cache
                    if c is None or c[0] is not args[0] or c[1] is not args[1]:
                        from .step_loss import masked_metrics_native
                        c = self._metric_cache = (args[0], args[1], masked_metrics_native(args[0], args[1], float(nv)))
                    return _Deferred(c[2][k])
            out = super().metric_forward(metric_func, args)
            if self._deferring and metric_func is not self.loss and torch.is_tensor(out) and out.is_cuda:
                return _Deferred(out.detach())
            return out

        def update_epoch_meter(self, name, value, n=1):
            if isinstance(value, _Deferred):
                self._pending.setdefault(name, []).append((value.value, n))          # kept on the device: nothing is launched, nothing waited for
                return
            super().update_epoch_meter(name, value, n)

        def flush_meters(self):
            """feed the deferred values to the epoch meters (one device synchronisation for all of them)"""
            pending, self._pending = self._pending, {}
            if pending:
                means, counts = [], []
                for name, items in pending.items():
                    vals = torch.stack([v.float().reshape(()) for v, _ in items])
                    ns = torch.tensor([float(n) for _, n in items], device=vals.device)
                    means.append((vals * ns).sum() / ns.sum())
                    counts.append(sum(n for _, n in items))
                for name, m, c in zip(pending, torch.stack(means).cpu().tolist(), counts):
                    super().update_epoch_meter(name, m, c)

        def print_epoch_meters(self, meter_type):
            self.flush_meters()
            return super().print_epoch_meters(meter_type)

        def plt_epoch_meters(self, meter_type, step):
            self.flush_meters()
            return super().plt_epoch_meters(meter_type, step)

    NativeRunner.__name__ = "Native" + base.__name__
    NativeRunner.__qualname__ = NativeRunner.__name__
    return NativeRunner
