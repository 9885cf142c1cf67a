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
        self.rebind(force=True)

    def zero(self):
        """Start a step: clear the bucket and detach `p.grad`, so that autograd hands each parameter its fresh gradient
        tensor (a pointer move) instead of launching one accumulate-add per parameter into a bound view; `collect()`
        gathers them with one multi-tensor copy."""
        self.flat.zero_()
        for p in self.params:
            p.grad = None

    def collect(self):
        """Gather the gradients autograd produced into the flat bucket and re-bind `p.grad` to its views."""
        dst, src = [], []
        for p, o, k in self._layout:
            g = p.grad
            if g is None:
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
        if dist.is_initialized() and dist.get_world_size() > 1:
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
