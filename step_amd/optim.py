"""Fused clip_grad_norm_ + Adam for the native STEP module -- and for `TSFormer(mode="pre-train")` after `flatten_parameters()` -- (one pass over flat
buffers, libstep_hip).

Drop-in for the pair of torch calls the reference's training loop makes through easytorch
(``CFG.TRAIN.CLIP_GRAD_PARAM`` + ``CFG.TRAIN.OPTIM``, reference ``step/STEP_PEMS04.py:90-106``): same update
rule as ``torch.optim.Adam`` (L2 weight decay folded into the gradient, bias-corrected moments, eps added
after the square root) preceded by ``torch.nn.utils.clip_grad_norm_``.  Parameters that receive no gradient
(the reference leaves them at ``grad=None``, so torch's Adam skips them) are not part of the flat buffer.

It consumes the flat gradient buffer of the LAST native backward (``model._flat_grad``): one backward per ``step()``, as in the
reference's loop.  With gradient accumulation over several backwards use ``torch.optim.Adam`` on ``model.parameters()``
(``bench.py --torch-optim``), which reads the accumulated ``.grad`` tensors.
"""
import torch

from . import _lib


class FusedAdamClip(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` (so ``torch.optim.lr_scheduler.MultiStepLR`` -- the reference's ``CFG.TRAIN.LR_SCHEDULER``,
    step/STEP_PEMS04.py:98-102 -- drives ``param_groups[0]["lr"]`` as usual) whose single parameter is the model's flat buffer."""

    def __init__(self, model, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None, param_grads=True):
        """``param_grads=False`` (STEP only): the native backward leaves the gradients in the flat buffer this optimizer reads and does
        not hand ~100 per-parameter ``.grad`` views to autograd (0.3-0.5 ms of host time per step); ``p.grad`` stays None."""
        self.model = model
        if not param_grads and hasattr(model, "flat_gradients_only"):
            model.flat_gradients_only = True
        self.flat = model._flat_param if model._flat_param is not None else model.flatten_parameters()
        super().__init__([self.flat], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.max_norm = max_norm
        self.step_count = 0
        self.work = torch.empty(int(_lib.lib().step_adam_work_floats()), device=self.flat.device)
        self.grad_norm = torch.zeros(1, device=self.flat.device)
        self.dyn = None                     # device StepDynState while a GraphedTrainStep drives this optimizer
        model._backward_count = 0

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)          # STEP.zero_grad also resets the flat-gradient bookkeeping

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("FusedAdamClip.step() takes no closure")
        g = self.model._flat_grad
        if g is None:
            raise RuntimeError("FusedAdamClip.step(): no native backward has run since zero_grad()")
        if self.model._backward_count != 1:
            raise RuntimeError(f"FusedAdamClip.step(): {self.model._backward_count} native backwards since zero_grad() -- the flat gradient buffer "
                               "holds the last one only; accumulate with torch.optim.Adam on model.parameters() instead")
        if self.flat is not self.model._flat_param or g.numel() != self.flat.numel():
            # enable_native_data_parallel(shard_graph_learner=True) / flatten_parameters() after this optimizer was built re-home the
            # parameters: the captured buffer and moments no longer match the gradient layout
            raise RuntimeError("FusedAdamClip: the model's flat parameter buffer changed after the optimizer was created "
                               f"({self.flat.numel()} parameters here, {g.numel()} gradient values): build the optimizer after "
                               "enable_native_data_parallel() / flatten_parameters()")
        # Whatever averaged the gradients (DDP bucket views, a hand-written all-reduce over p.grad) must have acted on THIS buffer --
        # autograd clones a view that has another owner, DDP may swap in its own.  First, middle and last parameter are enough to tell
        # (the buffer is adopted or replaced as a whole).  Both the pre-training TSFormer and STEP hand out views of one flat buffer.
        ps = self.model._pt_params() if hasattr(self.model, "_pt_params") else self.model._trainable_list()
        lo, hi = g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()
        for q in (ps[0], ps[len(ps) // 2], ps[-1]):
            if q.grad is not None and not (lo <= q.grad.data_ptr() < hi):
                raise RuntimeError("FusedAdamClip: a parameter's .grad is not a view of the native flat gradient buffer (cloned by autograd or "
                                   "re-homed by a DDP wrapper): reduce model._flat_grad itself, or step with torch.optim.Adam")
        pg = self.param_groups[0]
        self.step_count += 1
        extra = None
        max_norm = float(self.max_norm or 0.0)
        sh = getattr(getattr(self.model, "discrete_graph_learning", None), "_shard", None)      # (a TSFormer in pre-training mode has no graph learner)
        if sh is not None:
            # the fc weight slices of the other ranks belong to the model's gradient norm: the sum of their squared norms came back
            # in the layout's spare slot with the gradient all-reduce (step.py backward) -- no collective of its own
            extra = self.model._other_slices_sumsq
            if max_norm > 0.0 and self.dyn is None and getattr(self.model, "_total_sumsq", None) is not None:
                # ... as the whole squared norm, formed identically on every rank (step.py backward): max_norm < 0 tells the kernel to take
                # it as is.  Only with a real threshold: -0.0 is not "< 0" for the kernel, which would then add its own sum to the total
                extra, max_norm = self.model._total_sumsq, -max_norm
        if self.dyn is not None:
            # replayed (graph-captured) step: the step count and the learning rate are read from the device state (step_amd/graphed.py
            # advances the count at the head of every replay and copies the scheduler's rate when it changes)
            _lib.call("step_adam_clip_dyn", _lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                      self.flat.numel(), float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]), float(pg["weight_decay"]),
                      float(self.max_norm or 0.0), _lib.ptr(extra), _lib.ptr(self.work), _lib.ptr(self.grad_norm), _lib.ptr(self.dyn),
                      _lib.stream())
            return
        _lib.call("step_adam_clip_sharded", _lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                  self.flat.numel(), float(pg["lr"]), float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]),
                  float(pg["weight_decay"]), self.step_count, max_norm, _lib.ptr(extra), _lib.ptr(self.work),
                  _lib.ptr(self.grad_norm), _lib.stream())

    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "param_groups": groups}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in src.items() if k != "params"})
