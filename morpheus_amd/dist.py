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
    """Flat gradient bucket; `zero()` ... backward ... `allreduce_mean()` (collect + the one exchange step) is a step.
    After `collect()` / `allreduce_mean()` every `p.grad` is a view of `flat`."""

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
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self._early = {}          # id(param) -> (param, offset, numel): gradients exchanged as soon as they land
        self._early_hooks = []
        self._early_seen = 0
        self._early_work = []     # outstanding async collectives of this step
        self._side = None         # side stream the early exchange is queued on (GPU only)
        self.missing = set()      # layout indices whose p.grad was None at the last collect() (optim.FlatAdam skips them)
        self.rebind(force=True)

    # ---- overlap of the exchange with the tail of backward -----------------------------------------------
    def overlap_early(self, early_params: Iterable[torch.nn.Parameter]):
        """Exchange the gradients of `early_params` while backward is still running.

        On the render path the two hash tables (6.4 of the 7.45 MB) get their gradients from the brick kernels in the
        MIDDLE of backward; everything after that (the warp nets' backward-data and weight gradients, about half of the
        backward pass) does not touch them.  A post-accumulate hook on each early parameter copies its fresh gradient
        into the bucket and, once all of them have landed, queues ONE all-reduce of their (contiguous) range on a side
        stream, so it runs under the rest of backward; `allreduce_mean()` then only has the small remainder left on the
        critical path.  Contract: one backward per `zero()` (the bench / the reference's real-view steps); call
        `overlap_early([])` to switch it off for gradient-accumulating callers."""
        for h in self._early_hooks:
            h.remove()
        self._early_hooks, self._early = [], {}
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

    def _on_early_grad(self, p):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        _, o, k = self._early[id(p)]
        view = self.flat[o:o + k].view(p.shape)
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        self._early_seen += 1
        if self._early_seen != len(self._early):
            return
        a, b = self._early_span
        seg = self.flat[a:b]
        try:
            if seg.is_cuda:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=seg.device)
                cur = torch.cuda.current_stream(seg.device)      # the autograd thread's stream: the copies above are on it
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._early_work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
            else:
                self._early_work.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        except Exception as e:      # noqa: BLE001 -- never lose a step to the overlap: fall back to the single exchange
            import warnings
            warnings.warn(f"early gradient exchange disabled ({type(e).__name__}: {e}); using one all-reduce after backward")
            self._early_work = []
            for h in self._early_hooks:
                h.remove()
            self._early_hooks, self._early = [], {}

    def zero(self):
        """Start a step: clear the bucket and detach `p.grad`, so that autograd hands each parameter its fresh gradient
        tensor (a pointer move) instead of launching one accumulate-add per parameter into a bound view; `collect()`
        gathers them with one multi-tensor copy."""
        self.flat.zero_()
        self.missing = set()
        self._early_seen = 0
        for p in self.params:
            p.grad = None

    def collect(self):
        """Gather the gradients autograd produced into the flat bucket and re-bind `p.grad` to its views."""
        dst, src = [], []
        for i, (p, o, k) in enumerate(self._layout):
            g = p.grad
            if g is None:
                self.missing.add(i)
                p.grad = self.flat[o:o + k].view(p.shape)       # no gradient this step: the zeros of zero()
            elif g.data_ptr() != self.flat[o:o + k].data_ptr():
                view = self.flat[o:o + k].view(p.shape)
                dst.append(view)
                src.append(g)
                p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def rebind(self, force: bool = False):
        """Re-attach views (call if something replaced p.grad, e.g. zero_grad(set_to_none=True))."""
        for p, o, k in self._layout:
            if force or p.grad is None or p.grad.data_ptr() != self.flat[o:o + k].data_ptr():
                p.grad = self.flat[o:o + k].view(p.shape)

    def allreduce_mean(self):
        self.collect()
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        if self._early_work:
            # the early range is already summed (or in flight on the side stream): exchange only what is left
            a, b = self._early_span
            n = self.flat.numel()
            for lo, hi in ((0, a), (b, n)):
                if hi > lo:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM)
            for w in self._early_work:
                w.wait()                       # nccl: the current stream waits for the collective; gloo: host wait
            if self._side is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            self._early_work = []
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4


def shard_rays(n_rays: int, rank: int, world: int):
    """Contiguous block of a frame's rays for this rank (single-frame strong split)."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)
