"""Data-parallel ray sharding: one process per GPU, replicated parameters, ONE all-reduce per step.

The reference has no distributed code at all; this is the build's addition.  Rays (or
frames) are independent given the parameters, so the only exchange on the path is the gradient sum.
All gradients live in ONE flat fp32 bucket (~1.86 M elements = 7.45 MB for the snoopy model: two
3.2 MB hash tables + MLPs + codes + poses); `p.grad` are views into it, so backward accumulates in
place and a single RCCL all-reduce over xGMI moves everything -- at this size the collective is
latency-bound (~13 MB per GPU on a ring), which is why it is one call and not one per parameter.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT) -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # MORPHEUS_DIST_BACKEND=gloo lets the N>1 code path be exercised on a box with fewer GPUs than ranks
            backend = os.environ.get("MORPHEUS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
    return rank, local, world


class GradBucket:
    """Flat gradient bucket; `zero()` ... backward ... `allreduce_mean()` (collect + the exchange step) is a step.
    After `collect()` / `allreduce_mean()` every `p.grad` is a view of `flat`.

    Replicas stay identical by construction, whatever the data does on one rank:
      * the SEQUENCE of collectives of a step depends only on how the bucket was configured (early range or not), never on
        which gradients happened to arrive on a rank -- an early range that was not sent from the hooks (a rank whose batch
        produced no gradient for one table) is sent at `allreduce_mean()`, before the remainder, so every rank issues
        [early range][remainder] in that order;
      * one has-gradient flag per parameter rides behind the gradients in the same buffer and is summed by the remainder's
        all-reduce: a parameter counts as "no gradient" (torch.optim.Adam then skips it: no moment decay, no step
        increment) only when it had none on EVERY rank.  The summed flags stay on the DEVICE (`grad_counts`):
        optim.FlatAdam hands them to mh_adam_step_dev, which skips and counts per parameter without a host
        synchronisation; `resolve_missing()` reads them back for callers that need the answer on the host;
      * the early exchange fires after `backwards_per_step` backward passes (the reference accumulates a virtual-view and
        a real-view backward before one optimiser step when `freeze_lr` is off, morpheus.py:1396-1424).  A backward pass
        BEYOND the declared count finds the early range already summed across ranks; its gradients are kept apart, summed
        by one extra all-reduce at `allreduce_mean()` and added -- the result is still the mean of the accumulated
        gradients -- and the bucket then stops overlapping (one all-reduce after backward from the next step on; the call
        pattern is code, identical on every rank, so every rank switches at the same step)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        plist = [p for p in params if p.requires_grad]
        layout, o = [], 0
        for p in plist:
            layout.append((p, o, p.numel()))
            o += p.numel()
        self._init(layout, o, plist[0].device)

    @classmethod
    def from_layout(cls, layout, n: int, device):
        """Bucket with a caller-chosen layout [(param, offset, numel)] of total length n (optim.FlatAdam shares its
        parameter layout so that one offset addresses parameter, gradient and both moments)."""
        self = cls.__new__(cls)
        self._init(list(layout), n, device)
        return self

    def _init(self, layout, n, device):
        self._layout = layout
        self.params: List[torch.nn.Parameter] = [p for p, _, _ in layout]
        self._index = {id(p): i for i, (p, _, _) in enumerate(layout)}
        # [gradients: n | has-gradient flags: one per parameter | one spare that stays 0]; `flat` is the gradient part
        self._buf = torch.zeros(n + len(layout) + 1, dtype=torch.float32, device=device)
        self.flat = self._buf[:n]
        self._flags = self._buf[n:n + len(layout)]
        self.exchanged = False    # allreduce_mean() ran on several ranks since zero(): `grad_counts` is this step's
        self._early = {}          # id(param) -> (param, offset, numel): gradients exchanged as soon as they land
        self._early_hooks = []
        self._early_hits = 0      # post-accumulate hooks fired since zero()
        self._early_fired = False  # the early range of this step is summed (or in flight)
        self._early_disabled = False
        self._backwards = 1
        self._landed = set()      # layout indices whose gradient a hook already moved into the bucket this step
        self._late = None         # gradients of backward passes beyond `backwards_per_step` (early range only)
        self._late_used = False
        self._early_work = []     # outstanding async collectives of this step
        self._side = None         # side stream the early exchange is queued on (GPU only)
        self.missing = set()      # layout indices without gradient at the last collect() (on every rank, once exchanged)
        # the parameters' gradient views of `flat`, made once (two tensor operations per parameter: collect() runs twice per step
        # over ~60 parameters, and an eager training step is host-bound)
        self._gviews = [self.flat[o:o + k].view(p.shape) for p, o, k in layout]
        self._gptrs = [v.data_ptr() for v in self._gviews]
        self.rebind(force=True)

    # ---- overlap of the exchange with the tail of backward -----------------------------------------------
    def overlap_early(self, early_params: Iterable[torch.nn.Parameter], backwards_per_step: int = 1):
        """Exchange the gradients of `early_params` while backward is still running.

        On the render path the two hash tables (6.4 of the 7.45 MB) get their gradients from the brick kernels in the
        MIDDLE of backward; everything after that (the warp nets' backward-data and weight gradients, about half of the
        backward pass) does not touch them.  A post-accumulate hook on each early parameter moves its fresh gradient
        into the bucket and, once all of them have landed `backwards_per_step` times, queues ONE all-reduce of their
        (contiguous) range on a side stream, so it runs under the rest of backward; `allreduce_mean()` then only has the
        small remainder left on the critical path.  `overlap_early([])` switches it off."""
        for h in self._early_hooks:
            h.remove()
        self._early_hooks, self._early = [], {}
        self._early_disabled, self._backwards = False, max(int(backwards_per_step), 1)
        ids = {id(p) for p in early_params}
        for p, o, k in self._layout:
            if id(p) in ids:
                self._early[id(p)] = (p, o, k)
        if not self._early:
            return
        spans = sorted((o, o + k) for _, o, k in self._early.values())
        self._early_span = (spans[0][0], spans[-1][1])
        covered = sum(b - a for a, b in spans)
        # pad elements between groups (optim.FlatAdam aligns groups to 4) may sit inside the span: they are zero everywhere
        if self._early_span[1] - self._early_span[0] - covered > 4 * len(spans):
            raise ValueError("early parameters must be adjacent in the bucket layout (put them first)")
        for p, _, _ in self._early.values():
            self._early_hooks.append(p.register_post_accumulate_grad_hook(self._on_early_grad))

    @staticmethod
    def _multi_rank() -> bool:
        return dist.is_initialized() and dist.get_world_size() > 1

    def _send_early(self, async_op: bool):
        a, b = self._early_span
        seg = self.flat[a:b]
        if async_op and seg.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=seg.device)
            cur = torch.cuda.current_stream(seg.device)      # the autograd thread's stream: the hooks' copies are on it
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._early_work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        elif async_op:
            self._early_work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        else:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        self._early_fired = True

    def _on_early_grad(self, p):
        if not self._multi_rank() or self._early_disabled or p.grad is None:
            return
        _, o, k = self._early[id(p)]
        i = self._index[id(p)]
        g = p.grad
        p.grad = None               # the next backward pass hands over a fresh tensor instead of adding into the bucket
        if self._early_fired:
            # a backward pass beyond `backwards_per_step`: the range is already summed over the ranks (maybe still in
            # flight) -- keep this gradient apart; allreduce_mean() sums and adds it
            a, b = self._early_span
            if self._late is None:
                self._late = torch.zeros(b - a, dtype=torch.float32, device=self.flat.device)
            self._late[o - a:o - a + k].view(p.shape).add_(g)
            self._late_used = True
            return
        view = self.flat[o:o + k].view(p.shape)
        if i in self._landed:
            view.add_(g)
        else:
            if g.data_ptr() != view.data_ptr():
                view.copy_(g)
            self._landed.add(i)
        self._early_hits += 1
        if self._early_hits == len(self._early) * self._backwards:
            self._send_early(async_op=True)     # a failing collective is fatal (the peers are already inside theirs): let it raise

    def zero(self):
        """Start a step: clear the bucket and detach `p.grad`, so that autograd hands each parameter its fresh gradient
        tensor (a pointer move) instead of launching one accumulate-add per parameter into a bound view; `collect()`
        gathers them with one multi-tensor copy."""
        if self._early_work:
            raise RuntimeError("GradBucket.zero(): the early exchange of the previous step was never completed "
                               "(call allreduce_mean() once per step)")
        self.flat.zero_()
        self.missing = set()
        self.exchanged = False
        self._landed = set()
        self._early_hits, self._early_fired = 0, False
        if self._late_used:
            self._late.zero_()
            self._late_used = False
        for p in self.params:
            p.grad = None

    def collect(self):
        """Gather the gradients autograd produced into the flat bucket and re-bind `p.grad` to its views."""
        dst, src = [], []
        views, ptrs, landed = self._gviews, self._gptrs, self._landed
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                if i not in landed:
                    self.missing.add(i)                         # no gradient this step: the zeros of zero()
                p.grad = views[i]
            elif g.data_ptr() != ptrs[i]:
                if i in landed:                                 # landed early, then another (un-hooked) accumulation
                    views[i].add_(g)
                else:
                    dst.append(views[i])
                    src.append(g)
                p.grad = views[i]
        if dst:
            torch._foreach_copy_(dst, src)

    def rebind(self, force: bool = False):
        """Re-attach views (call if something replaced p.grad, e.g. zero_grad(set_to_none=True))."""
        for i, p in enumerate(self.params):
            if force or p.grad is None or p.grad.data_ptr() != self._gptrs[i]:
                p.grad = self._gviews[i]

    def allreduce_mean(self):
        self.collect()
        if not self._multi_rank():
            return
        world = dist.get_world_size()
        # has-gradient flags: 1 everywhere, 0 for this rank's missing parameters -- two launches whatever their number (the index
        # tensor of a missing SET is built once: cfg3 has the same nine gradient-less tensors, pose + background net, every step)
        self._flags.fill_(1.0)
        if self.missing:
            self._flags.index_fill_(0, self._missing_index(), 0.0)
        n, total = self.flat.numel(), self._buf.numel()
        if self._early and not self._early_disabled:
            if not self._early_fired:
                # not every early gradient arrived on this rank (or fewer backward passes than declared): the peers'
                # sequence starts with the early range, so does ours
                self._send_early(async_op=False)
            a, b = self._early_span
            if a > 0:
                dist.all_reduce(self.flat[0:a], op=dist.ReduceOp.SUM)
            dist.all_reduce(self._buf[b:total], op=dist.ReduceOp.SUM)       # remainder + flags
            for w in self._early_work:
                w.wait()                       # nccl: the current stream waits for the collective; gloo: host wait
            if self._side is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            self._early_work = []
            if self._late_used:
                import warnings
                dist.all_reduce(self._late, op=dist.ReduceOp.SUM)
                self.flat[a:b].add_(self._late)
                warnings.warn(f"GradBucket: more than backwards_per_step={self._backwards} backward passes before the "
                              "exchange; the extra gradients were summed by a second all-reduce and the early overlap is "
                              "switched off from the next step on (declare overlap_early(..., backwards_per_step=k))")
                self._early_disabled = True
        else:
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
        self.flat.div_(world)
        self.exchanged = True

    def _missing_index(self) -> torch.Tensor:
        key = tuple(sorted(self.missing))
        cache = self.__dict__.setdefault("_missing_idx_cache", {})
        idx = cache.get(key)
        if idx is None:
            if len(cache) >= 16:
                cache.clear()
            idx = cache[key] = torch.tensor(key, dtype=torch.long, device=self._flags.device)
        return idx

    @property
    def grad_counts(self) -> torch.Tensor:
        """[n_params + 1] device floats, valid after a multi-rank allreduce_mean(): how many ranks had a gradient for each
        parameter of the layout this step; the extra last element is always 0 (a slot for segments that are no parameter)."""
        return self._buf[self.flat.numel():]

    def resolve_missing(self) -> set:
        """`missing` as a property of ALL ranks (one device->host read): the parameters that had no gradient anywhere."""
        if self.exchanged and self.missing:
            idx = sorted(self.missing)
            had = self._flags[self._missing_index()].tolist()
            self.missing = {i for i, c in zip(idx, had) if c == 0.0}
        return self.missing

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def shard_rays(n_rays: int, rank: int, world: int):
    """Contiguous block of a frame's rays for this rank (single-frame strong split)."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)
