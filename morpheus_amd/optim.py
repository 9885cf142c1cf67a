"""Optimiser step for the hot path's parameters: ONE flat bucket, one HIP launch.

`FlatAdam` stands where the reference builds `Adam(self.model.get_params_all(lr), betas=(0.9, 0.99), eps=1e-15)`
(morpheus.py:154-155) and is used the same way: `param_groups` keep their `name`/`lr` keys, so
`update_learning_rate` / `freeze_lr_deform` / `reset_lr_deform` (morpheus.py:472-528), which mutate
`param_group['lr']` by group name, work unchanged; `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format
(`step`, `exp_avg`, `exp_avg_sq` per parameter), so the reference's checkpoints (morpheus.py:329-358) interchange.

What is different underneath: parameters, gradients and both moments live in four flat fp32 buffers (1.86 M elements
= 7.45 MB each for snoopy.yaml; `p.data` / `p.grad` / the state tensors are views), groups are contiguous segments,
and `step()` is a single `mh_adam_step` launch instead of torch's ~10 multi-tensor launches over ~60 tensors per
group list.  The gradient buffer is a `dist.GradBucket`, so the data-parallel exchange is the same single all-reduce.

`FlatEMA` mirrors `torch_ema.ExponentialMovingAverage` as the reference uses it (morpheus.py:160-162, 1299-1301,
1368-1369, 1432-1433: `update()` once per epoch, `store()/copy_to()/restore()` around evaluation).  torch_ema is a
third-party package that is not in the reference tree (requirements.txt lists it unpinned); its published update rule is
restated here: decay_t = min(decay, (1 + n) / (10 + n)), shadow -= (1 - decay_t) (shadow - param).

There is no CPU path: `step()` raises if the parameters are not on a GPU or the HIP library is missing.
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List

import torch

from . import _lib
from ._lib import MorpheusHipError, check, ptr, stream
from .dist import GradBucket


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        plist: List[torch.nn.Parameter] = [p for g in self.param_groups for p in g["params"]]
        if not plist:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = plist[0].device
        for p in plist:
            if p.dtype != torch.float32 or p.device != dev or not p.requires_grad:
                raise NotImplementedError("FlatAdam: fp32 trainable parameters on one device only")
        for g in self.param_groups:
            if tuple(g["betas"]) != tuple(self.param_groups[0]["betas"]) or g["eps"] != self.param_groups[0]["eps"]:
                raise NotImplementedError("FlatAdam: betas/eps are shared by all groups (as in the reference)")
        # every group starts on a 4-element boundary so that the kernel's float4 lanes never straddle two groups' tensors
        # needlessly; the pad elements have zero gradient and never move
        # segments of the kernel: one per parameter tensor (its own step count and skip flag, as torch.optim.Adam keeps
        # them) plus one per alignment pad (permanently skipped)
        self._views = []
        off = 0
        self._seg_end = []          # group boundaries (kept for callers / tests)
        self._kseg_end, self._kseg_group, self._kseg_param = [], [], []     # kernel segments: end, group index, param index|-1
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                self._kseg_param.append(len(self._views))
                self._views.append((p, off, p.numel()))
                off += p.numel()
                self._kseg_end.append(off)
                self._kseg_group.append(gi)
            pad = (off + 3) // 4 * 4
            if pad != off:
                off = pad
                self._kseg_end.append(off)
                self._kseg_group.append(gi)
                self._kseg_param.append(-1)
            self._seg_end.append(off)
        if len(self._kseg_end) > 160:
            raise NotImplementedError("mh_adam_step takes at most 160 segments (parameter tensors + pads)")
        self.n = off
        self.flat_p = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        with torch.no_grad():
            for p, o, k in self._views:
                self.flat_p[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
        self._plist = [p for p, _, _ in self._views]
        self.bucket = GradBucket.from_layout(self._views, self.n, dev)
        self._steps = [0] * len(self._views)      # per-parameter step counts (torch.optim.Adam's state['step'])
        self._kseg_end_c = (ctypes.c_int64 * len(self._kseg_end))(*self._kseg_end)
        # data-parallel runs keep the counts on the device (mh_adam_step_dev: the skip decision is the all-reduced has-gradient
        # flag, which only the device knows without a sync); created at the first multi-rank step
        self._steps_dev = None
        self._bind_state()

    # ---- torch.optim.Adam-format state -----------------------------------------------------------------------------
    def _bind_state(self):
        for i, (p, o, k) in enumerate(self._views):
            self.state[p] = {"step": torch.tensor(float(self._steps[i])),
                             "exp_avg": self.exp_avg[o:o + k].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape)}

    def _pull_steps(self):
        """device-side step counts (data-parallel runs) -> the host list; one sync, at checkpoint time only"""
        if self._steps_dev is not None:
            per_seg = self._steps_dev.tolist()
            for s, pi in enumerate(self._kseg_param):
                if pi >= 0:
                    self._steps[pi] = int(per_seg[s])

    def state_dict(self):
        self._pull_steps()
        for i, (p, _, _) in enumerate(self._views):
            self.state[p]["step"] = torch.tensor(float(self._steps[i]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for i, (p, o, k) in enumerate(self._views):
                st = self.state.get(p, {})
                if "exp_avg" in st:
                    self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                    self._steps[i] = int(float(st["step"]))     # per parameter: a freeze_lr run steps the pose group less often
                else:   # torch's Adam creates state lazily: a parameter that never had a gradient has none
                    self.exp_avg[o:o + k].zero_()
                    self.exp_avg_sq[o:o + k].zero_()
                    self._steps[i] = 0
        self._steps_dev = None        # rebuilt from the host counts at the next multi-rank step
        self._bind_state()

    # ---- stepping ---------------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero()          # clears the bucket and detaches p.grad; step() gathers what backward produced

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step takes no closure")
        if not self.flat_p.is_cuda:
            raise MorpheusHipError("FlatAdam steps on an MI355X only; there is no CPU path")
        lib = _lib.load()
        self.bucket.collect()       # no-op when allreduce_mean() already gathered the gradients
        # the kernels below write the parameters through raw pointers: move their version counters as an in-place torch update
        # would (the model's step cache -- model.scene_representation._step_cache -- reads them to see that a step has ended)
        torch.autograd.graph.increment_version(self._plist)
        g0 = self.param_groups[0]
        ns = len(self._kseg_end)
        lrs = (ctypes.c_float * ns)(*[float(self.param_groups[gi]["lr"]) for gi in self._kseg_group])
        import torch.distributed as tdist
        if tdist.is_initialized() and tdist.get_world_size() > 1:
            # data-parallel: "has a gradient" is a property of all ranks -- the all-reduced flags of the bucket, on the device
            if not self.bucket.exchanged:
                raise RuntimeError("FlatAdam.step() on several ranks needs bucket.allreduce_mean() after backward (its "
                                   "has-gradient flags decide, identically on every rank, which parameters are stepped)")
            dev = self.flat_p.device
            if self._steps_dev is None:
                self._steps_dev = torch.tensor([0 if pi < 0 else self._steps[pi] for pi in self._kseg_param], dtype=torch.int64,
                                               device=dev)
                n_par = len(self._views)
                self._seg_flag_index = torch.tensor([n_par if pi < 0 else pi for pi in self._kseg_param], dtype=torch.long,
                                                    device=dev)           # pads read the spare slot that is always 0
                self._seg_scratch = torch.empty(2 * ns, dtype=torch.float32, device=dev)
            seg_flags = self.bucket.grad_counts[self._seg_flag_index]
            self.bucket.missing = set()
            check(lib.mh_adam_step_dev(ptr(self.flat_p), ptr(self.bucket.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n, ns,
                                       self._kseg_end_c, lrs, ptr(seg_flags), ptr(self._steps_dev), ptr(self._seg_scratch),
                                       float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), stream()),
                  "mh_adam_step_dev")
            return
        # torch.optim.Adam skips a parameter whose gradient is None: no moment decay, no move, no step increment
        no_grad = self.bucket.missing
        steps = []
        for pi in self._kseg_param:
            if pi < 0 or pi in no_grad:
                steps.append(0)
            else:
                self._steps[pi] += 1
                steps.append(self._steps[pi])
        self.bucket.missing = set()
        check(lib.mh_adam_step(ptr(self.flat_p), ptr(self.bucket.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n, ns,
                               self._kseg_end_c, lrs, (ctypes.c_int64 * ns)(*steps), float(g0["betas"][0]),
                               float(g0["betas"][1]), float(g0["eps"]), stream()), "mh_adam_step")


class FlatEMA:
    """Exponential moving average of a FlatAdam bucket -- ONE flat tensor instead of torch_ema's per-parameter list, so an update
    is three torch launches (subtract, scale, subtract) and one temporary whatever the number of parameters; torch_ema's
    interface.  (The reference updates its EMA once per epoch, morpheus.py:1432-1433: no kernel of its own.)"""

    def __init__(self, optimizer: FlatAdam, decay: float, use_num_updates: bool = True, parameters=None):
        """parameters: the iterable torch_ema would be given -- the reference passes `self.model.parameters()`
        (morpheus.py:160-162), whose REGISTRATION order (pose_array, deform_code, deform_net, ...) is the order of
        `shadow_params` in its checkpoints; without it the optimiser's group order is used (not interchangeable)."""
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.opt = optimizer
        by_id = {id(p): (p, o, k) for p, o, k in optimizer._views}
        if parameters is None:
            self._order = list(optimizer._views)
        else:
            plist = [p for p in parameters if p.requires_grad]
            missing = [tuple(p.shape) for p in plist if id(p) not in by_id]
            if missing or len(plist) != len(by_id):
                raise ValueError(f"FlatEMA: parameters and optimizer disagree ({len(plist)} vs {len(by_id)}; unknown {missing[:3]})")
            self._order = [by_id[id(p)] for p in plist]
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow = optimizer.flat_p.detach().clone()
        self.collected = None

    @torch.no_grad()
    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        self.shadow.sub_((self.shadow - self.opt.flat_p).mul_(1.0 - decay))

    @torch.no_grad()
    def store(self):
        self.collected = self.opt.flat_p.detach().clone()

    @torch.no_grad()
    def copy_to(self):
        self.opt.flat_p.copy_(self.shadow)
        torch.autograd.graph.increment_version(self.opt._plist)     # (the parameters are views of flat_p with counters of their own)

    @torch.no_grad()
    def restore(self):
        if self.collected is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        self.opt.flat_p.copy_(self.collected)
        torch.autograd.graph.increment_version(self.opt._plist)
        self.collected = None

    def state_dict(self):
        per = lambda flat: None if flat is None else [flat[o:o + k].view(p.shape).clone() for p, o, k in self._order]
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": per(self.shadow),
                "collected_params": per(self.collected)}

    @torch.no_grad()
    def load_state_dict(self, sd):
        for name in ("shadow_params", "collected_params"):
            lst = sd.get(name)
            if lst is None:
                continue
            if len(lst) != len(self._order):
                raise ValueError(f"FlatEMA.load_state_dict: {name} has {len(lst)} tensors, expected {len(self._order)}")
            for (p, o, k), t in zip(self._order, lst):
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"FlatEMA.load_state_dict: {name} shape {tuple(t.shape)} vs parameter {tuple(p.shape)} "
                                     "(was the EMA built with the same parameter order?)")
        self.decay, self.num_updates = sd["decay"], sd["num_updates"]
        for (p, o, k), t in zip(self._order, sd["shadow_params"]):
            self.shadow[o:o + k].copy_(t.reshape(-1))
        if sd.get("collected_params") is not None:
            self.collected = torch.zeros_like(self.shadow)
            for (p, o, k), t in zip(self._order, sd["collected_params"]):
                self.collected[o:o + k].copy_(t.reshape(-1))
