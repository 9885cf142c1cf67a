"""`OccupancyGrid` -- stands where nerfacc's `OccGridEstimator` stands in the reference
(morpheus.py:196-202 construction, :628-638 sampling, :905-913 update, :341,355 checkpoint key 'estimator').

nerfacc is third-party, un-vendored and version un-pinned in the reference (docs/INSTALL.md:21), so the semantics
here follow its published 0.5.x behaviour as summarised in SURVEY.md C.9 and are otherwise defined by this build:
  * binary grid resolution^3 over the AABB; `occs` float EMA-max field;
  * update_every_n_steps(step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16): every n-th step
    evaluate occ_eval_fn at one jittered point per selected cell (all cells during warm-up, afterwards a quarter
    uniformly at random plus a quarter of the occupied ones), occs = max(occs*decay, new),
    binaries = occs > min(mean(occs), occ_thre);
  * sampling(...): fixed-step marching of the occupied cells with one stratified near-plane jitter per ray
    (csrc/sampler.hip:march_kernel); sigma_fn / alpha_thre / early_stop_eps pruning is not applied (the reference
    passes sigma_fn=None, alpha_thre=0, early_stop_eps=0).
The heavy part of an update is occ_eval_fn = model.density on up to resolution^3 points, which runs on the HIP kernels.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops


class OccupancyGrid(torch.nn.Module):
    def __init__(self, roi_aabb, resolution: int = 128):
        super().__init__()
        aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(-1)
        assert aabb.numel() == 6
        b = float(aabb[3])
        assert torch.allclose(aabb[:3], -aabb[3:]) and torch.allclose(aabb[3:], aabb[3:4].expand(3)), \
            "the marcher assumes the reference's cubic AABB [-bound, bound]^3 (morpheus.py:113-127)"
        self.bound = b
        R = int(resolution)
        # buffer names / shapes / dtypes of nerfacc 0.5.x's OccGridEstimator with levels = 1 (as recalled from its public
        # source; the package is not in the reference tree): resolution int32 [3], aabbs [1,6], occs [R^3],
        # binaries bool [1,R,R,R] -- so that the reference's checkpoint entry 'estimator' (morpheus.py:341,355) loads
        self.register_buffer("resolution", torch.tensor([R, R, R], dtype=torch.int32))
        self.register_buffer("aabbs", aabb[None].clone())
        self.register_buffer("occs", torch.zeros(R ** 3))
        self.register_buffer("binaries", torch.zeros(1, R, R, R, dtype=torch.bool))
        self._R = R
        self.packed = None       # (ray_start, ray_cnt) of the last sampling() call, consumed by the compositor
        self.fixed_jitter: Optional[torch.Tensor] = None   # parity runs pin the per-ray jitter
        # Fixed-capacity sampling (None = off: ragged packed samples sized by one device->host sync, as nerfacc does).
        # With a capacity the packed arrays always have that length, the first `n_valid` (a device int) entries are the
        # samples and the rest is padding no ray owns: constant shapes and no sync, so a whole training step can be captured
        # in a HIP graph (trainstep.GraphedRealViewStep).  `overflow` (device int, sticky) is set when a batch had more
        # samples than the capacity -- its tail rays were truncated -- and is the caller's cue to raise the capacity.
        self.sample_capacity: Optional[int] = None
        self.n_valid: Optional[torch.Tensor] = None
        self.overflow: Optional[torch.Tensor] = None

    # -- sampling --------------------------------------------------------------------------------
    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, render_step_size=1e-3, alpha_thre=0.0, stratified=False,
                 cone_angle=0.0, early_stop_eps=0.0):
        if sigma_fn is not None or alpha_thre > 0 or early_stop_eps > 0 or cone_angle != 0.0:
            raise NotImplementedError("density-based pruning / cone marching are not used by the reference's call "
                                      "(morpheus.py:629-638) and are not implemented")
        n = rays_o.shape[0]
        if isinstance(self.fixed_jitter, float):
            u = torch.full((n,), self.fixed_jitter, device=rays_o.device)     # one value for every ray (chunk-invariant)
        elif self.fixed_jitter is not None:
            u = self.fixed_jitter
        elif stratified:
            u = torch.rand(n, device=rays_o.device)
        else:
            u = None
        # a bool tensor is one byte per cell holding 0/1: the marcher reads it as uint8 without a copy
        binary = self.binaries[0].view(torch.uint8)
        if self.sample_capacity is not None:
            ri, ts, te, rs, rc, self.n_valid, ovf = ops.march_rays_capped(rays_o, rays_d, u, float(render_step_size), self.bound,
                                                                          binary, int(self.sample_capacity))
            if self.overflow is None:
                self.overflow = torch.zeros((), dtype=torch.int32, device=rays_o.device)
            self.overflow.copy_(torch.maximum(self.overflow, ovf))
        else:
            ri, ts, te, rs, rc = ops.march_rays(rays_o, rays_d, u, float(render_step_size), self.bound, binary)
            self.n_valid = None
        self.packed = (rs, rc)
        return ri, ts, te

    # -- occupancy update -------------------------------------------------------------------------
    def _cell_points(self, idx: torch.Tensor) -> torch.Tensor:
        R = self._R
        ijk = torch.stack([idx // (R * R), (idx // R) % R, idx % R], -1).float()
        x = (ijk + torch.rand_like(ijk)) / R                       # [0,1]^3, one jittered point per cell
        return x * (2 * self.bound) - self.bound

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16):
        if step % n != 0:
            return
        R3, dev = self._R ** 3, self.occs.device
        if step < warmup_steps:
            idx = torch.arange(R3, device=dev)
        else:
            # a quarter of the cells uniformly at random plus up to a quarter drawn among the occupied ones -- without the
            # device->host sync of a nonzero(): every cell gets a random key, unoccupied cells key 0, and the k largest keys
            # are k occupied cells drawn without replacement (fewer occupied cells than k: the zero-key remainder re-evaluates
            # unoccupied cells, which only refreshes them).  Fixed shapes: k + k points every refresh.
            k = R3 // 4
            uni = torch.randint(R3, (k,), device=dev)
            keys = torch.rand(R3, device=dev) * self.binaries.reshape(-1)
            occ_idx = torch.topk(keys, k, sorted=False).indices
            idx = torch.cat([uni, occ_idx])
        occ = occ_eval_fn(self._cell_points(idx)).reshape(-1).float()
        # EMA-max update.  `idx` may name a cell more than once (two uniform draws, or a uniform and an occupied draw): the
        # cell then takes the LARGEST of its evaluations -- a scatter-max, deterministic -- where an indexed assignment
        # would keep whichever duplicate the hardware wrote last.
        new = torch.maximum(self.occs[idx] * ema_decay, occ)
        self.occs.scatter_reduce_(0, idx, new, reduce="amax", include_self=False)
        thre = torch.clamp(self.occs.mean(), max=occ_thre)
        self.binaries.copy_((self.occs > thre).view_as(self.binaries))

    def set_binary(self, binary: torch.Tensor):
        """Install a binary grid directly (tests / loading an external estimator)."""
        self.binaries.copy_(binary.to(torch.bool).view_as(self.binaries))
