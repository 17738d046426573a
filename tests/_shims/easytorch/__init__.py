"""Test-only stand-in for easy-torch's Runner: the pieces of the training loop the reference's BasicTS runners rely on
(model construction + device placement, optimizer / scheduler from the config, epoch meters, the train loop with
zero_grad -> backward -> clip_grad_norm_ -> step).  One process, one device, no checkpoints, no logging to disk."""
import logging

import torch

from .device import to_device


class _Meter:
    def __init__(self):
        self.sum, self.n, self.last = 0.0, 0, None

    def update(self, v, n=1):
        self.sum += v * n
        self.n += n
        self.last = v

    @property
    def avg(self):
        return self.sum / max(self.n, 1)


class Runner:
    def __init__(self, cfg):
        self.logger = logging.getLogger("easytorch-shim")
        self.device = torch.device(cfg.get("_DEVICE", "cuda" if torch.cuda.is_available() else "cpu"))
        self.model = self.define_model(cfg).to(self.device)
        self.meters = {}
        self.optim = self.scheduler = None
        self.clip_grad_param = None
        self.train_data_loader = self.val_data_loader = None
        self.num_epochs = None

    # ---- what BasicTS overrides / calls
    @staticmethod
    def define_model(cfg):
        raise NotImplementedError

    def to_running_device(self, x):
        return to_device(x, self.device)

    def register_epoch_meter(self, name, meter_type, fmt="{:f}", plt=True):
        self.meters[name] = _Meter()

    def update_epoch_meter(self, name, value, n=1):
        self.meters[name].update(value, n)

    def reset_epoch_meters(self):
        for k in self.meters:
            self.meters[k] = _Meter()

    def print_epoch_meters(self, meter_type):
        pass

    def plt_epoch_meters(self, meter_type, step):
        pass

    def save_best_model(self, epoch, metric_name, greater_best=True):
        pass

    def save_model(self, epoch):
        pass

    def build_train_data_loader(self, cfg):
        from .core.data_loader import build_data_loader
        return build_data_loader(self.build_train_dataset(cfg), cfg["TRAIN"]["DATA"])

    def build_val_data_loader(self, cfg):
        from .core.data_loader import build_data_loader
        return build_data_loader(self.build_val_dataset(cfg), cfg["VAL"]["DATA"])

    def init_training(self, cfg):
        self.train_data_loader = self.build_train_data_loader(cfg)
        self.register_epoch_meter("train_time", "train", "{:.2f} (s)", plt=False)
        self.register_epoch_meter("lr", "train", "{:.2e}")
        o = cfg["TRAIN"]["OPTIM"]
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optim = getattr(torch.optim, o["TYPE"])(params, **o["PARAM"])
        s = cfg["TRAIN"].get("LR_SCHEDULER")
        if s is not None:
            self.scheduler = getattr(torch.optim.lr_scheduler, s["TYPE"])(self.optim, **s["PARAM"])
        self.clip_grad_param = cfg["TRAIN"].get("CLIP_GRAD_PARAM")
        self.num_epochs = cfg["TRAIN"]["NUM_EPOCHS"]

    def init_validation(self, cfg):
        self.val_data_loader = self.build_val_data_loader(cfg)
        self.register_epoch_meter("val_time", "val", "{:.2f} (s)", plt=False)

    def backward(self, loss):
        self.optim.zero_grad()
        loss.backward()
        if self.clip_grad_param is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), **self.clip_grad_param)
        self.optim.step()

    def train(self, cfg, max_iters=None):
        """epoch loop of easytorch.Runner.train, cut after max_iters iterations in total (tests)."""
        self.init_training(cfg)
        done, losses = 0, []
        for epoch in range(1, self.num_epochs + 1):
            self.model.train()
            for it, data in enumerate(self.train_data_loader):
                loss = self.train_iters(epoch, it, data)
                self.backward(loss)
                losses.append(loss.detach())          # (easytorch does not read the loss back per iteration either)
                done += 1
                if max_iters is not None and done >= max_iters:
                    return [float(l) for l in losses]
            if self.scheduler is not None:
                self.scheduler.step()
        return [float(l) for l in losses]


def launch_training(cfg, devices=None, node_rank=0):
    raise NotImplementedError("the test shim is not a launcher")


def launch_runner(cfg, fn, args=(), devices=None):
    raise NotImplementedError("the test shim is not a launcher")
