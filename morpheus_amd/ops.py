"""torch.autograd wrappers over the C ABI (include/morpheus_hip.h).

Each Function allocates its outputs/workspaces as torch tensors (the reference's ownership model,
external/encoders/gridencoder/grid.py:50,56,84,87), hands raw device pointers plus torch's CURRENT
stream to libmorpheus_hip.so and returns.  There is no CPU or PyTorch fallback: a CPU tensor or a
missing library raises MorpheusHipError.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_gpu, stream
from .packing import field_joint_packer, field_packer, warp_joint_packer, warp_packer



class KernelTimer:
    """Optional per-C-ABI-call timing with events recorded on the launch stream (torch's current
    stream, which is the one handed to the library).  Used by bench.py for the roofline numbers.
    Events come from a pool that is recycled at reset(): creating two fresh events per call costs the host 20-50 us,
    which a short step (cfg2: 5 ms of GPU work, ~12 timed calls) would feel."""

    def __init__(self):
        self.enabled = False
        self.records = {}
        self._pool, self._used = [], 0

    def reset(self, enabled: bool):
        self.enabled, self.records, self._used = enabled, {}, 0

    def _event(self):
        if self._used == len(self._pool):
            self._pool.append(torch.cuda.Event(enable_timing=True))
        e = self._pool[self._used]
        self._used += 1
        return e

    @staticmethod
    def _stream():
        """the current stream as a torch Stream object, one object per raw handle (Event.record() without a stream argument
        resolves torch.cuda.current_stream() through three Python layers: 5.4 us against 1.5 -- twice per timed call)"""
        raw = stream()
        s = _STREAM_OBJECTS.get(raw)
        if s is None:
            s = _STREAM_OBJECTS[raw] = torch.cuda.current_stream()
        return s

    def start(self):
        if not self.enabled:
            return None
        e = self._event()
        e.record(self._stream())
        return e

    def stop(self, name: str, e0):
        if e0 is None:
            return
        e1 = self._event()
        e1.record(self._stream())
        self.records.setdefault(name, []).append((e0, e1))

    def summary(self):
        """name -> (calls, total_ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.records.items()}


_STREAM_OBJECTS: dict = {}
TIMER = KernelTimer()


_I32_CACHE = {}


def _i32arr(a):
    """host int32 array + its ctypes pointer for the C ABI's *_host arguments.  The arrays the hot path passes are static
    geometry (level tables, tile offsets, layer lists): lists / tuples and read-only numpy arrays are converted once and cached
    by value -- a training step makes ~150 of these calls."""
    # one key form for every source kind: (length, values...) of Python ints -- a list [3, 1, 2, 3] and the array [1, 2, 3] must
    # not meet in one entry
    if type(a) is tuple or type(a) is list:
        # fast path: the tuple of the values themselves (hash-equal to the canonical key's tail for Python and numpy ints alike);
        # the element types are checked once, when the entry is made
        k2 = ("seq", tuple(a))
        hit = _I32_CACHE.get(k2)
        if hit is not None:
            return hit
        if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in a):
            key = (len(a),) + tuple(int(v) for v in a)
            hit = _I32_CACHE.get(key)
            if hit is None:
                arr = np.ascontiguousarray(a, dtype=np.int32)
                arr.setflags(write=False)
                hit = _I32_CACHE[key] = (arr, arr.ctypes.data_as(ctypes.c_void_p))
            _I32_CACHE[k2] = hit
            return hit
        key = None
    elif isinstance(a, np.ndarray) and a.dtype == np.int32 and a.ndim == 1 and a.size <= 64:
        key = (a.size,) + tuple(a.tolist())
    else:
        key = None
    if key is not None:
        hit = _I32_CACHE.get(key)
        if hit is None:
            arr = np.ascontiguousarray(a, dtype=np.int32)
            arr.setflags(write=False)
            hit = _I32_CACHE[key] = (arr, arr.ctypes.data_as(ctypes.c_void_p))
        return hit
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------ hash grid
def _scratch(n: int, device, dtype=torch.float32) -> torch.Tensor:
    """An uninitialised 1-D scratch buffer of n elements for the kernels' parked tiles / per-point rows, allocated in SIZE CLASSES
    (eight per octave: n rounded up by at most 12.5 %).  The packed sample count of a training step changes from step to step, and
    so did the byte counts of its ~40 large allocations: the caching allocator can split a cached block but cannot grow one, so
    every step whose batch was a little larger than any before it opened new segments beside the cached ones -- the 180 x 180
    virtual-view step allocates 104 GB at its peak and had 245 GB RESERVED (profiles/r06_park_alloc.txt).  With size classes a
    step's requests find the previous step's blocks."""
    n = int(n)
    if n >= (1 << 18):
        q = 1 << (n.bit_length() - 4)
        return torch.empty((n + q - 1) // q * q, dtype=dtype, device=device)[:n]
    return torch.empty(n, dtype=dtype, device=device)


def level_resolutions(L: int, per_level_scale: float, base: int) -> np.ndarray:
    """res_l = (uint32)ceil(exp2f(l*S)*H) in float32, S = (float)log2(per_level_scale)
    (gridencoder.cu:133, grid.py:39).  Host-computed so no device libm is involved."""
    S = np.float32(np.log2(per_level_scale))
    l = np.arange(L, dtype=np.float32)
    return np.ceil(np.exp2(l * S).astype(np.float32) * np.float32(base)).astype(np.int32)


def effective_levels(max_level, L: int) -> int:
    """grid.py:42."""
    return L if max_level is None else max(min(int(math.ceil(max_level * L)), L), 1)


def _bin_points(lib, x, bound):
    """Counting-sort the points into 16^3 bricks (shared by every table evaluated at x)."""
    M, dev = x.shape[0], x.device
    ws = torch.empty(lib.mh_grid_bin_workspace_ints(), dtype=torch.int32, device=dev)
    perm = _scratch(M, dev, torch.int32)
    bstart = torch.empty(lib.mh_grid_bin_index_ints(), dtype=torch.int32, device=dev)
    _e = TIMER.start()
    check(lib.mh_grid_bin_points(ptr(x), M, bound, ptr(ws), ptr(perm), ptr(bstart), stream()), "mh_grid_bin_points")
    TIMER.stop("mh_grid_bin_points", _e)
    return perm, bstart


# ---- arithmetic mode of the MLP kernels (MORPHEUS_MLP, or set_mlp_mode() at run time) -------------------------------------
# Values, parked tiles, accumulation and results are fp32 in every mode; what differs is how a product reaches the matrix cores.
#   "b3"  (default) bf16 x 3: every fp32 operand is cut EXACTLY into three bf16 slices (all 24 significand bits), the six
#         significant cross products go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation (csrc/mlp_b3.hip): warp nets
#         forward / backward-data / weight gradients and the field forward.  fp32-faithful: what is dropped is below 2^-24 of a
#         product, less than the rounding of an fp32 multiply-add.
#   "f32" the native fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere (csrc/mlp.hip): the reference's arithmetic instruction for
#         instruction, and the slowest (the fp32 MFMA runs at the vector-ALU rate on gfx950).
# (A third form, two fp16 slices at power-of-two scales -- 22-bit operands, NOT fp32-faithful -- existed in rounds 3-5 and was
# deleted in round 6: git history, csrc/mlp_h2.hip.)
# The field nets' fused backward follows the b3 mode too (mh_field_bwd_fused_b3); the f32 mode runs it on the native fp32
# MFMA (mh_field_bwd_fused); see FIELD_BWD below.
MLP_MODES = ("b3", "f32")
MODE_DTYPE = {"f32": "f32", "b3": "f32 (exact 3 x bf16 operand split, 6 slice products per MAC on the bf16 MFMA pipe, fp32 accumulate)"}
_mode = os.environ.get("MORPHEUS_MLP", "b3")
if _mode not in MLP_MODES:
    raise ValueError(f"MORPHEUS_MLP={_mode!r}: expected one of {MLP_MODES}")


def mlp_mode() -> str:
    return _mode


def set_mlp_mode(mode: str) -> str:
    """Select the PROCESS DEFAULT arithmetic of the MLP kernels, for operand packs prepared from now on by models that do not
    carry a mode of their own (`scene_representation.mlp_mode`).  A pack keeps the arithmetic it was prepared with (see
    _warp_mode): changing the default between a forward and its backward, or inside a model.operand_scope(), does not mix
    arithmetic forms.  -> the previous default."""
    global _mode
    if mode not in MLP_MODES:
        raise ValueError(f"mlp mode {mode!r}: expected one of {MLP_MODES}")
    prev, _mode = _mode, mode
    return prev


def _warp_mode(mode: Optional[str] = None) -> str:
    """operand-pack tag of the MLP nets: "b3" / "" (native fp32 MFMA).  `mode`: an explicit choice (a model's own
    `mlp_mode`); None = the process default (MORPHEUS_MLP / set_mlp_mode).  The tag is fixed when a pack is prepared and travels
    with it (MLPOperands.mode -> every forward's ctx): a forward and its backward always use the same arithmetic, and two models
    of one process may differ."""
    mode = _mode if mode is None else mode
    if mode not in MLP_MODES:
        raise ValueError(f"mlp mode {mode!r}: expected one of {MLP_MODES}")
    return "" if mode == "f32" else mode


def _grid_fwd(lib, x, embs, o_p, r_p, L, n_levels, bound, group):
    """One forward launch per table at the same points -> (list of [M, L*2], binning or None).
    Calls of at least mh_grid_stage_min_points() points (2^20 unless tuned) are binned into bricks FIRST and run the
    brick-staged forward (rows read from LDS instead of eight gathers per point and level; same bits); the binning is handed
    back for the caller's backward, which would otherwise make it itself.  That includes the large finite-difference-tap calls
    of a whole-view step (group = 6; same-box A/B against the gather-sharing grouped kernel: 72 x 72 view 0.66 -> 0.49 ms of
    forward per step, 180 x 180 3.7 -> 2.2 ms); smaller calls keep the grouped kernel."""
    M = x.shape[0]
    outs = []
    binned = None
    if L == 16 and M >= lib.mh_grid_stage_min_points(-1):     # (whatever `group`: a staged brick shares a cell's rows among all its points)
        binned = _bin_points(lib, x, bound)
    for emb in embs:
        out = _scratch(M * L * 2, x.device).view(M, L * 2)
        _e = TIMER.start()
        if binned is not None:
            check(lib.mh_grid_encode_fwd_binned(ptr(x), ptr(emb), o_p, r_p, ptr(binned[0]), ptr(binned[1]), ptr(out), M, L,
                                                n_levels, float(bound), stream()), "mh_grid_encode_fwd_binned")
        else:
            check(lib.mh_grid_encode_fwd(ptr(x), ptr(emb), o_p, r_p, ptr(out), M, L, n_levels, float(bound), int(group), stream()),
                  "mh_grid_encode_fwd")
        TIMER.stop("mh_grid_encode_fwd_binned" if binned is not None else "mh_grid_encode_fwd", _e)
        outs.append(out)
    return outs, binned


def _grid_bwd(lib, x, embs, grads, o_p, r_p, L, n_levels, bound, need_dx, gmax_ptrs=None, sums=None, g_x_into=None, binned=None):
    """Embedding (and position) gradients of several tables evaluated at the same points: ONE brick binning shared by all
    tables.  gmax_ptrs[k]: device address of max|grads[k]| as float bits when its producer reduced it on the fly
    (mh_field_bwd_data), else None -> the kernel reduces it itself.
    sums[k]: device ADDRESS of a running table-gradient sum this call adds into (the brick kernel's flush is an atomic add
    anyway), or None -> a fresh zero-filled table.  g_x_into: a [M,3] gradient of the same points that the tables' d/dx is added
    to in the kernel (brick path only), or None.
    -> (g_x or None, [g_emb; None where the gradient went into sums[k]])."""
    M = x.shape[0]
    g_x_total, g_embs = g_x_into, []          # binned: the forward's binning of the same points, when it made one
    brick = M > 0 and L == 16          # the brick kernel's level table is sized for the shipped 16-level geometry
    for k, (emb, grad) in enumerate(zip(embs, grads)):
        if grad is None:
            g_embs.append(None)
            continue
        grad = grad.contiguous()
        into = None if sums is None else sums[k]
        g_emb = torch.zeros_like(emb) if into is None else None
        g_emb_p = ptr(g_emb) if into is None else ctypes.c_void_p(into)
        if brick:
            if binned is None:
                binned = _bin_points(lib, x, bound)
            acc_dx = need_dx and g_x_total is not None
            g_x = g_x_total if acc_dx else (torch.empty_like(x) if need_dx else None)
            _e = TIMER.start()
            check(lib.mh_grid_encode_bwd_binned(ptr(grad), ptr(x), ptr(emb), o_p, r_p, ptr(binned[0]), ptr(binned[1]),
                                                g_emb_p, ptr(g_x), int(acc_dx), M, L, n_levels, bound,
                                                None if gmax_ptrs is None else gmax_ptrs[k], stream()),
                  "mh_grid_encode_bwd_binned")
            TIMER.stop("mh_grid_encode_bwd_binned", _e)
            g_x_total = g_x
        else:
            g_x = torch.empty_like(x) if need_dx else None
            _e = TIMER.start()
            check(lib.mh_grid_encode_bwd(ptr(grad), ptr(x), ptr(emb), o_p, r_p, g_emb_p, ptr(g_x), M, L, n_levels,
                                         bound, stream()), "mh_grid_encode_bwd")
            TIMER.stop("mh_grid_encode_bwd", _e)
            if need_dx:
                g_x_total = g_x if g_x_total is None else g_x_total + g_x
        g_embs.append(g_emb)
    return (g_x_total if need_dx else None), g_embs


class _GridEncode(torch.autograd.Function):
    """grid.py:25-96 (_grid_encode) on the HIP kernels, for one or several tables evaluated at the
    same points (sdf + colour encoders share x: one index computation pattern, one binning).
    No dy_dx tensor is materialised; backward recomputes corner weights."""

    @staticmethod
    def forward(ctx, x, offsets_np, res_np, n_levels, bound, group, *embs):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(x, *embs)
        lib = _lib.load()
        x = x.detach().contiguous().float()
        L = len(res_np)
        o_np, o_p = _i32arr(offsets_np)
        r_np, r_p = _i32arr(res_np)
        saved = [emb.detach().contiguous() for emb in embs]
        outs, ctx.binned = _grid_fwd(lib, x, saved, o_p, r_p, L, n_levels, bound, group)
        ctx.save_for_backward(x, *saved)
        ctx.meta = (o_np, r_np, n_levels, float(bound), L)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        x, *embs = ctx.saved_tensors
        o_np, r_np, n_levels, bound, L = ctx.meta
        o_p, r_p = o_np.ctypes.data_as(ctypes.c_void_p), r_np.ctypes.data_as(ctypes.c_void_p)
        g_x, g_embs = _grid_bwd(lib, x, embs, grads, o_p, r_p, L, n_levels, bound, ctx.needs_input_grad[0], binned=ctx.binned)
        return (g_x, None, None, None, None, None, *g_embs)


def grid_encode(x, emb, offsets_np, res_np, bound, max_level=None, group: int = 1):
    """x [..,3] in world units -> [.., L*2] (grid.py:152-169).  group: every `group` consecutive points are neighbours
    (6 = finite-difference taps, point-major): a gather-sharing hint, results do not depend on it."""
    L = len(res_np)
    lead = list(x.shape[:-1])
    (out,) = _GridEncode.apply(x.reshape(-1, 3), offsets_np, res_np, effective_levels(max_level, L), bound, group, emb)
    return out.view(lead + [L * 2])


def grid_encode_multi(x, embs, offsets_np, res_np, bound, max_level=None):
    """Several tables with identical level geometry at the same points -> tuple of [M, L*2]."""
    L = len(res_np)
    return _GridEncode.apply(x.reshape(-1, 3), offsets_np, res_np, effective_levels(max_level, L), bound, 1, *embs)


# ------------------------------------------------------------------------------------ compositor
class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma, t_starts, t_ends, rgb, ray_start, ray_cnt, padded=False):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt)
        lib = _lib.load()
        sigma, t_starts, t_ends = sigma.detach().contiguous(), t_starts.contiguous(), t_ends.contiguous()
        rgbc = rgb.detach().contiguous()
        N, M = ray_start.shape[0], sigma.shape[0]
        dev = sigma.device
        # padded: the packed arrays are longer than the rays' samples (fixed-capacity sampling); entries no ray owns are
        # never visited by the kernels (one wavefront per ray) and must read as weight 0 / gradient 0
        ctx.padded = bool(padded)
        weights = torch.zeros(M, device=dev) if padded else torch.empty(M, device=dev)
        opacity, depth = torch.empty(N, device=dev), torch.empty(N, device=dev)
        color = torch.empty(N, 3, device=dev)
        _e = TIMER.start()
        check(lib.mh_composite_fwd(ptr(sigma), ptr(t_starts), ptr(t_ends), ptr(rgbc), ptr(ray_start), ptr(ray_cnt),
                                   ptr(weights), ptr(opacity), ptr(depth), ptr(color), N, stream()), "mh_composite_fwd")
        TIMER.stop("mh_composite_fwd", _e)
        ctx.save_for_backward(sigma, t_starts, t_ends, rgbc, ray_start, ray_cnt, weights)
        return weights, opacity, depth, color

    @staticmethod
    def backward(ctx, g_w, g_o, g_d, g_c):
        lib = _lib.load()
        sigma, ts, te, rgb, ray_start, ray_cnt, weights = ctx.saved_tensors
        N = ray_start.shape[0]
        c = lambda t: None if t is None else t.contiguous()
        d_sigma = torch.zeros_like(sigma) if ctx.padded else torch.empty_like(sigma)
        d_rgb = torch.zeros_like(rgb) if ctx.padded else torch.empty_like(rgb)
        _e = TIMER.start()
        check(lib.mh_composite_bwd(ptr(sigma), ptr(ts), ptr(te), ptr(rgb), ptr(ray_start), ptr(ray_cnt), ptr(weights),
                                   ptr(c(g_w)), ptr(c(g_o)), ptr(c(g_d)), ptr(c(g_c)), ptr(d_sigma), ptr(d_rgb), N,
                                   stream()), "mh_composite_bwd")
        TIMER.stop("mh_composite_bwd", _e)
        return d_sigma, None, None, d_rgb, None, None, None


def composite(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt, padded: bool = False):
    """-> weights [M], opacity [N], depth [N], color [N,3]  (morpheus.py:675-685).  padded: see _Composite.forward."""
    return _Composite.apply(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt, padded)


def packed_info(ray_indices: torch.Tensor, n_rays: int):
    """(ray_start, ray_cnt) int32 from sorted packed ray indices (no host sync)."""
    cnt = torch.bincount(ray_indices, minlength=n_rays)
    start = torch.cumsum(cnt, 0) - cnt
    return start.to(torch.int32), cnt.to(torch.int32)


# ------------------------------------------------------------------------------------ sampler / rays
def generate_rays(fx, fy, cx, cy, c2w, H, W, device):
    lib = _lib.load()
    c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(4, 4))
    o = torch.empty(H * W, 3, device=device)
    d = torch.empty(H * W, 3, device=device)
    require_gpu(o)
    _e = TIMER.start()
    check(lib.mh_generate_rays(float(fx), float(fy), float(cx), float(cy), c2w.ctypes.data_as(ctypes.c_void_p), H, W,
                               ptr(o), ptr(d), stream()), "mh_generate_rays")
    TIMER.stop("mh_generate_rays", _e)
    return o, d


def sample_uniform(rays_o, rays_d, jitter, S: int, bound: float, with_xyz: bool = False):
    """-> ray_idx int32 [N*S], t_starts, t_ends, xyz|None, ray_start, ray_cnt (no_grad, like morpheus.py:628)."""
    require_gpu(rays_o, rays_d, jitter)
    lib = _lib.load()
    o, d, j = rays_o.detach().contiguous(), rays_d.detach().contiguous(), _ray_jitter(jitter, rays_o.shape[0])
    N, dev = o.shape[0], o.device
    ri = torch.empty(N * S, dtype=torch.int32, device=dev)
    ts, te = torch.empty(N * S, device=dev), torch.empty(N * S, device=dev)
    xyz = torch.empty(N * S, 3, device=dev) if with_xyz else None
    rs, rc = torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
    _e = TIMER.start()
    check(lib.mh_sample_uniform(ptr(o), ptr(d), ptr(j), N, S, float(bound), ptr(ri), ptr(ts), ptr(te), ptr(xyz), ptr(rs),
                                ptr(rc), stream()), "mh_sample_uniform")
    TIMER.stop("mh_sample_uniform", _e)
    return ri, ts, te, xyz, rs, rc


def rays_sample_uniform(fx, fy, cx, cy, c2w, H: int, W: int, pix, jitter, S: int, bound: float, with_xyz: bool = False):
    """Ray generation + uniform sampler in one launch (`mh_rays_sample_uniform`).  `pix`: int32 [N] pixel indices
    (j*W+i) on the device, the draw of datasets/dataset.py:412-423; None = the whole image.
    -> rays_o, rays_d [N,3], then the `sample_uniform` tuple."""
    require_gpu(jitter)
    lib = _lib.load()
    c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(4, 4))
    j, dev = jitter.contiguous(), jitter.device
    N = j.shape[0]
    if pix is not None:
        require_gpu(pix)
        if pix.dtype != torch.int32 or pix.shape != (N,):
            raise ValueError("pix must be int32 [N], one pixel index per jitter value")
        pix = pix.contiguous()
    elif N > H * W:
        raise ValueError("whole-image mode renders at most H*W rays")
    o, d = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
    ri = torch.empty(N * S, dtype=torch.int32, device=dev)
    ts, te = torch.empty(N * S, device=dev), torch.empty(N * S, device=dev)
    xyz = torch.empty(N * S, 3, device=dev) if with_xyz else None
    rs, rc = torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev)
    _e = TIMER.start()
    check(lib.mh_rays_sample_uniform(float(fx), float(fy), float(cx), float(cy), c2w.ctypes.data_as(ctypes.c_void_p), H, W,
                                     ptr(pix), ptr(j), N, S, float(bound), ptr(o), ptr(d), ptr(ri), ptr(ts), ptr(te),
                                     ptr(xyz), ptr(rs), ptr(rc), stream()), "mh_rays_sample_uniform")
    TIMER.stop("mh_rays_sample_uniform", _e)
    return o, d, ri, ts, te, xyz, rs, rc


def _ray_jitter(jitter, n_rays: int):
    """the per-ray near-plane jitter as the marcher reads it: float32 [n_rays], one value per ray -- the kernels index it by ray
    without a length of their own, so a buffer sized for another batch (a graphed step's static jitter met by an eager render)
    must not get through"""
    if jitter is None:
        return None
    if jitter.numel() != n_rays or jitter.dtype != torch.float32:
        raise ValueError(f"per-ray jitter: expected float32 [{n_rays}], got {jitter.dtype} {tuple(jitter.shape)}")
    return jitter.reshape(-1).contiguous()


def march_rays(rays_o, rays_d, jitter, step: float, bound: float, binary: torch.Tensor):
    """Occupancy-grid marcher -> (ray_idx int32 [M], t_starts [M], t_ends [M], ray_start [N], ray_cnt [N]).
    One host sync on the path: M = total sample count sizes the packed arrays (nerfacc synchronises at the same point).
    Single pass, one wavefront per ray (mh_march_slots + mh_march_pack); a batch with rays that overflow the per-ray slot
    row (directions much shorter than unit length) is marched again with a slot row twice as long."""
    require_gpu(rays_o, rays_d, jitter, binary)
    lib = _lib.load()
    o, d = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    j = _ray_jitter(jitter, rays_o.shape[0])
    assert binary.dtype == torch.uint8 and binary.is_contiguous() and binary.dim() == 3
    N, R, dev = o.shape[0], binary.shape[0], o.device
    if N == 0:
        e = torch.empty(0, device=dev)
        z = torch.empty(0, dtype=torch.int32, device=dev)
        return z, e, e.clone(), z.clone(), z.clone()
    cap = int(lib.mh_march_cap(float(step), float(bound)))
    while True:
        cnt_ovf = torch.zeros(N + 1, dtype=torch.int32, device=dev)     # [ray_cnt | overflow flag]
        cnt = cnt_ovf[:N]
        slots = torch.empty(2, N, cap, device=dev)
        _e = TIMER.start()
        check(lib.mh_march_slots(ptr(o), ptr(d), ptr(j), N, float(step), float(bound), R, ptr(binary), cap, ptr(cnt),
                                 ptr(slots[0]), ptr(slots[1]), cnt_ovf.data_ptr() + 4 * N, stream()), "mh_march_slots")
        TIMER.stop("mh_march_slots", _e)
        csum = torch.cumsum(cnt_ovf, 0, dtype=torch.int32)              # last element = M + overflow flag
        start = (csum[:N] - cnt).contiguous()
        M, tot = csum[N - 1:].tolist()                                  # the one device->host sync
        if tot == M:                                                    # no overflow
            break
        if cap > (1 << 20):
            raise _lib.MorpheusHipError("march_rays: a ray takes more than 2^20 steps inside the box (step size / direction scale?)")
        cap *= 2
    ri = torch.empty(M, dtype=torch.int32, device=dev)
    ts, te = torch.empty(M, device=dev), torch.empty(M, device=dev)
    if M > 0:
        _e = TIMER.start()
        check(lib.mh_march_pack(ptr(start), ptr(cnt), ptr(slots[0]), ptr(slots[1]), N, cap, ptr(ri), ptr(ts), ptr(te),
                                stream()), "mh_march_pack")
        TIMER.stop("mh_march_pack", _e)
    return ri, ts, te, start, cnt.contiguous()


def march_count(rays_o, rays_d, jitter, step: float, bound: float, binary: torch.Tensor) -> torch.Tensor:
    """Total number of samples the marcher would emit for these rays, as a 0-dim int32 DEVICE tensor (no host sync): lets a
    caller size / pick a fixed-capacity buffer ahead of the step (trainstep.GraphedRealViewStep)."""
    require_gpu(rays_o, rays_d, jitter, binary)
    lib = _lib.load()
    o, d = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    j = _ray_jitter(jitter, rays_o.shape[0])
    N, R, dev = o.shape[0], binary.shape[0], o.device
    cap = int(lib.mh_march_cap(float(step), float(bound)))
    cnt_ovf = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    slots = torch.empty(2, N, cap, device=dev)
    check(lib.mh_march_slots(ptr(o), ptr(d), ptr(j), N, float(step), float(bound), R, ptr(binary), cap, ptr(cnt_ovf),
                             ptr(slots[0]), ptr(slots[1]), cnt_ovf.data_ptr() + 4 * N, stream()), "mh_march_slots")
    return cnt_ovf[:N].sum(dtype=torch.int32)


def march_rays_capped(rays_o, rays_d, jitter, step: float, bound: float, binary: torch.Tensor, capacity: int):
    """The same marcher with a FIXED packed length and no device->host sync, so that a training step has constant shapes and
    can be captured in a HIP graph: -> (ray_idx int32 [capacity], t_starts, t_ends [capacity], ray_start [N], ray_cnt [N],
    n_valid int32 0-dim, overflow int32 0-dim).  The first n_valid entries are the packed samples (identical to march_rays');
    the rest is padding (ray 0, t = 0) that no ray owns: ray_cnt is clamped so that start + cnt never passes `capacity`.
    overflow != 0: the batch had more samples than `capacity` (the tail rays were truncated) or a ray overflowed its slot
    row -- the caller re-captures with a larger capacity."""
    require_gpu(rays_o, rays_d, jitter, binary)
    lib = _lib.load()
    o, d = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    j = _ray_jitter(jitter, rays_o.shape[0])
    assert binary.dtype == torch.uint8 and binary.is_contiguous() and binary.dim() == 3
    N, R, dev = o.shape[0], binary.shape[0], o.device
    assert N > 0 and capacity > 0
    cap = int(lib.mh_march_cap(float(step), float(bound)))
    cnt_ovf = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    cnt = cnt_ovf[:N]
    slots = torch.empty(2, N, cap, device=dev)
    _e = TIMER.start()
    check(lib.mh_march_slots(ptr(o), ptr(d), ptr(j), N, float(step), float(bound), R, ptr(binary), cap, ptr(cnt),
                             ptr(slots[0]), ptr(slots[1]), cnt_ovf.data_ptr() + 4 * N, stream()), "mh_march_slots")
    TIMER.stop("mh_march_slots", _e)
    csum = torch.cumsum(cnt, 0, dtype=torch.int32)
    start = (csum - cnt).contiguous()
    total = csum[N - 1]
    cnt_c = torch.minimum(cnt, (capacity - start).clamp(min=0)).contiguous()
    n_valid = total.clamp(max=capacity)
    overflow = ((total > capacity) | (cnt_ovf[N] != 0)).to(torch.int32)
    ri = torch.zeros(capacity, dtype=torch.int32, device=dev)
    ts, te = torch.zeros(capacity, device=dev), torch.zeros(capacity, device=dev)
    _e = TIMER.start()
    check(lib.mh_march_pack(ptr(start), ptr(cnt_c), ptr(slots[0]), ptr(slots[1]), N, cap, ptr(ri), ptr(ts), ptr(te), stream()),
          "mh_march_pack")
    TIMER.stop("mh_march_pack", _e)
    return ri, ts, te, start, cnt_c, n_valid, overflow


# ------------------------------------------------------------------------------------ field-query glue (csrc/normal.hip)
class _FdTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, topo, eps, bound):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(x, topo)
        lib = _lib.load()
        xc = x.detach().contiguous().float()
        M, dev = xc.shape[0], xc.device
        tc = None if topo is None else topo.detach().contiguous().float()
        C = 0 if tc is None else tc.shape[1]
        taps = _scratch(6 * M * 3, dev).view(6 * M, 3)
        topo6 = torch.empty(6 * M, C, device=dev) if tc is not None else torch.empty(0, device=dev)
        check(lib.mh_fd_taps(ptr(xc), ptr(tc), C, float(eps), float(bound), M, ptr(taps), ptr(topo6 if tc is not None else None),
                             stream()), "mh_fd_taps")
        ctx.save_for_backward(xc)
        ctx.meta = (float(eps), float(bound), C, topo is not None)
        if not ctx.needs_input_grad[0]:
            ctx.mark_non_differentiable(taps)       # taps of positions without gradient: the field skips its d/dx stage
        if topo is None or not ctx.needs_input_grad[1]:
            ctx.mark_non_differentiable(topo6)
        return taps, topo6

    @staticmethod
    def backward(ctx, g_taps, g_topo6):
        lib = _lib.load()
        (xc,) = ctx.saved_tensors
        eps, bound, C, has_topo = ctx.meta
        M, dev = xc.shape[0], xc.device
        want_x = ctx.needs_input_grad[0] and g_taps is not None
        want_t = has_topo and ctx.needs_input_grad[1] and g_topo6 is not None
        g_x = torch.empty(M, 3, device=dev) if want_x else None
        g_t = torch.empty(M, C, device=dev) if want_t else None
        if want_x or want_t:
            check(lib.mh_fd_taps_bwd(ptr(xc), ptr(g_taps.contiguous() if want_x else None),
                                     ptr(g_topo6.contiguous() if want_t else None), C, eps, bound, M, ptr(g_x), ptr(g_t), stream()),
                  "mh_fd_taps_bwd")
        return g_x, g_t, None, None


def fd_taps(x, topo, eps: float, bound: float):
    """-> taps [6M,3] (point-major: +x,-x,+y,-y,+z,-z, clamped to the box), topo6 [6M,C] or None  (model.py:367-376)."""
    taps, topo6 = _FdTaps.apply(x, topo, eps, bound)
    return taps, (None if topo is None else topo6)


class _FdNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf6, eps):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(sdf6)
        lib = _lib.load()
        s = sdf6.detach().contiguous().float()
        M, dev = s.shape[0], s.device
        normal, raw = torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev)
        check(lib.mh_fd_normal_fwd(ptr(s), float(eps), M, ptr(normal), ptr(raw), stream()), "mh_fd_normal_fwd")
        ctx.save_for_backward(s)
        ctx.eps = float(eps)
        return normal, raw

    @staticmethod
    def backward(ctx, g_n, g_r):
        lib = _lib.load()
        (s,) = ctx.saved_tensors
        g = torch.empty_like(s)
        c = lambda t: None if t is None else t.contiguous()
        check(lib.mh_fd_normal_bwd(ptr(s), ptr(c(g_n)), ptr(c(g_r)), ctx.eps, s.shape[0], ptr(g), stream()), "mh_fd_normal_bwd")
        return g, None


def fd_normal(sdf6, eps: float):
    """sdf6 [M,6] -> (normal [M,3] = nan_to_num(safe_normalize(raw)), raw [M,3])   (model.py:377-398)."""
    return _FdNormal.apply(sdf6, eps)


class _MultiCode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, v0, v1, v2):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(t, v0, v1, v2)
        lib = _lib.load()
        tc = t.detach().reshape(-1).contiguous().float()
        vs = [v.detach().contiguous() for v in (v0, v1, v2)]           # [1, C, size, 1]: contiguous == [C, size]
        C, sizes, F = vs[0].shape[1], [v.shape[2] for v in vs], tc.shape[0]
        out = torch.empty(F, 3 * C, device=tc.device)
        check(lib.mh_multicode_fwd(ptr(tc), ptr(vs[0]), ptr(vs[1]), ptr(vs[2]), sizes[0], sizes[1], sizes[2], C, F, ptr(out),
                                   stream()), "mh_multicode_fwd")
        ctx.save_for_backward(tc)
        ctx.meta = (C, sizes, [tuple(v.shape) for v in vs])
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load()
        (tc,) = ctx.saved_tensors
        C, sizes, shapes = ctx.meta
        flat = torch.zeros(C * sum(sizes), device=tc.device)
        gs, o = [], 0
        for sz in sizes:
            gs.append(flat[o:o + C * sz])
            o += C * sz
        check(lib.mh_multicode_bwd(ptr(tc), ptr(g_out.contiguous()), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), sizes[0], sizes[1],
                                   sizes[2], C, tc.shape[0], stream()), "mh_multicode_bwd")
        return (None, *[g.view(sh) for g, sh in zip(gs, shapes)])


def multicode_sample(t, volumes):
    """t [F] or [F,1] in [0,1] (clamped) -> [F, 3*C]: deform_code.py:20-38 for the three-level code grid, one launch."""
    assert len(volumes) == 3
    return _MultiCode.apply(t, *volumes)


class _SdfLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_sdf, ts, te, ray_idx, rays_depth, rays_mask, trunc, n_valid=None):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(pred_sdf, ts, te, ray_idx, rays_depth, rays_mask, n_valid)
        lib = _lib.load()
        p = pred_sdf.detach().contiguous().float()
        dep = rays_depth.detach().reshape(-1).contiguous().float()
        msk = None if rays_mask is None else rays_mask.detach().reshape(-1).contiguous().float()
        sums = torch.empty(3, device=p.device)
        nv = None if n_valid is None else n_valid.detach().reshape(1).to(torch.int32).contiguous()
        check(lib.mh_sdf_losses_fwd(ptr(p), ptr(ts), ptr(te), ptr(ray_idx), ptr(dep), ptr(msk), float(trunc), p.shape[0], ptr(nv),
                                    ptr(sums), stream()), "mh_sdf_losses_fwd")
        ctx.save_for_backward(p, ts, te, ray_idx, dep, msk, sums, nv)
        ctx.trunc = float(trunc)
        return sums[0] / sums[2], sums[1] / sums[2]

    @staticmethod
    def backward(ctx, g_fs, g_sl):
        lib = _lib.load()
        p, ts, te, ray_idx, dep, msk, sums, nv = ctx.saved_tensors
        g = torch.empty_like(p)
        c = lambda t: None if t is None else t.reshape(1).contiguous().float()
        check(lib.mh_sdf_losses_bwd(ptr(p), ptr(ts), ptr(te), ptr(ray_idx), ptr(dep), ptr(msk), ctx.trunc, p.shape[0], ptr(nv),
                                    ptr(sums), ptr(c(g_fs)), ptr(c(g_sl)), ptr(g), stream()), "mh_sdf_losses_bwd")
        return g, None, None, None, None, None, None, None


def sdf_losses(pred_sdf, t_starts, t_ends, ray_idx, rays_depth, rays_mask, trunc: float, n_valid=None):
    """-> (fs_loss, sdf_loss) of utils.py:91-113 on packed samples; rays_depth / rays_mask are PER RAY ([N] or [N,1], mask may
    be None) and read through ray_idx (int32 [M]); the sample depth is (t_starts + t_ends) / 2.  n_valid: 0-dim device int
    tensor -- only the first n_valid packed entries are samples (march_rays_capped) -- or None."""
    return _SdfLosses.apply(pred_sdf, t_starts.contiguous(), t_ends.contiguous(), ray_idx.contiguous(), rays_depth, rays_mask,
                            trunc, n_valid)


class _SamplePositions(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, ray_idx, ts, te, ray_start, ray_cnt):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(rays_o, rays_d, ray_idx, ts, te)
        lib = _lib.load()
        o, d = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
        M = ts.shape[0]
        xyz = torch.empty(M, 3, device=o.device)
        check(lib.mh_sample_positions(ptr(o), ptr(d), ptr(ray_idx), ptr(ts), ptr(te), M, ptr(xyz), stream()), "mh_sample_positions")
        ctx.save_for_backward(ts, te, ray_start, ray_cnt)
        ctx.n = o.shape[0]
        return xyz

    @staticmethod
    def backward(ctx, g_xyz):
        lib = _lib.load()
        ts, te, ray_start, ray_cnt = ctx.saved_tensors
        g_o, g_d = torch.empty(ctx.n, 3, device=ts.device), torch.empty(ctx.n, 3, device=ts.device)
        check(lib.mh_sample_positions_bwd(ptr(g_xyz.contiguous()), ptr(ts), ptr(te), ptr(ray_start), ptr(ray_cnt), ctx.n, ptr(g_o),
                                          ptr(g_d), stream()), "mh_sample_positions_bwd")
        return g_o, g_d, None, None, None, None, None


def sample_positions(rays_o, rays_d, ray_idx, t_starts, t_ends, ray_start, ray_cnt):
    """xyz [M,3] = rays_o[ri] + rays_d[ri] * (t_starts + t_ends) / 2 (morpheus.py:644-647) for packed, ray-major samples;
    backward is a per-ray segment sum (no sort, no atomics).  ray_idx int32 [M]; ray_start / ray_cnt int32 [N]."""
    return _SamplePositions.apply(rays_o, rays_d, ray_idx.contiguous(), t_starts.contiguous(), t_ends.contiguous(),
                                  ray_start.contiguous(), ray_cnt.contiguous())


# ------------------------------------------------------------------------------------ fused MLPs
# A/B switch (tools/gpu/r6_pad_ab.sh): 0 = the weight-gradient kernels read every row of the padded tiles (needs a library built with
# -DMH_PARK_PAD_ROWS, whose forward / backward-data kernels write them)
WGRAD_LIVE_ROWS = os.environ.get("MORPHEUS_WGRAD_LIVE", "1") != "0"
# A/B and test switch: 0 = dPre4 of the warp nets is parked by backward-data and read by the weight gradients (the round-5 form)
REGEN_DPRE4 = os.environ.get("MORPHEUS_REGEN_DPRE4", "1") != "0"


def _wgrad(lib, acts, dpre, acts_tile, dpre_tile, act_off, dpre_off, in_pad, out_pad, n_tiles, dev, tag, b3=False, in_live=None,
           out_live=None):
    """in_live / out_live: rows of each layer's input / dPre tile that carry values (None: all; include/morpheus_hip.h)"""
    n_layers = len(act_off)
    if not WGRAD_LIVE_ROWS:
        in_live = out_live = None
    il_p = None if in_live is None else _i32arr(in_live)[1]
    ol_p = None if out_live is None else _i32arr(out_live)[1]
    dw_len = int(sum(i * o for i, o in zip(in_pad, out_pad)))
    db_len = int(sum(out_pad))
    a_np, a_p = _i32arr(act_off)
    d_np, d_p = _i32arr(dpre_off)
    i_np, i_p = _i32arr(in_pad)
    o_np, o_p = _i32arr(out_pad)
    ws = torch.empty(lib.mh_mlp_wgrad_workspace_floats(n_layers, i_p, o_p, n_tiles), device=dev)
    # an empty query (n_tiles == 0) returns MH_OK without writing: the token gradient must then be zeros, not heap contents
    raw = torch.empty(dw_len + db_len, device=dev) if n_tiles > 0 else torch.zeros(dw_len + db_len, device=dev)
    dw_raw, db_raw = raw[:dw_len], raw[dw_len:]
    _e = TIMER.start()
    fn = lib.mh_mlp_wgrad_b3 if b3 else lib.mh_mlp_wgrad
    check(fn(ptr(acts), ptr(dpre), acts_tile, dpre_tile, n_layers, a_p, d_p, i_p, o_p, il_p, ol_p, ptr(ws), ptr(dw_raw), ptr(db_raw),
             n_tiles, stream()), "mh_mlp_wgrad")
    TIMER.stop("mh_mlp_wgrad[" + tag + "]", _e)
    return raw          # dw_raw | db_raw, tile-row order (packing.JointPacker.unpack_grads maps it back)


WARP_ACT_ROWS, WARP_DPRE_ROWS = 64 + 2 * 640 + 40, 2 * 672     # csrc/mlp.hip: activations + 40 rows of ReLU masks
FIELD_ACT_ROWS = 96 + 64 * 5 + 8   # activations + 8 rows of ReLU masks


# A/B and test switch: False = every field query returns its own gradient tensors and autograd adds them up (the round-3 form).
# The in-place sums are keyed on autograd's graph-task id (torch >= 2.1); without it the per-query form is used.
_GRAPH_TASK_ID = getattr(torch._C, "_current_graph_task_id", None)
ACCUMULATE_IN_PLACE = _GRAPH_TASK_ID is not None and os.environ.get("MORPHEUS_ACCUMULATE_IN_PLACE", "1") != "0"


class _QueryAccumulator:
    """Running gradient sums of ONE backward pass, shared by the field queries that were given the same prepared operands
    (`MLPOperands.acc`; a real-view training step issues six: render, finite-difference taps, regularisers, point loss).

    Each query's backward produces (a) the raw weight / bias / beta gradient [24 929], (b) a full table gradient [419 640, 2] per
    hash table.  Handed to autograd one tensor per query, that is a zero fill per table and an add per tensor per extra query
    (about 30 launches of 0.1 - 3.4 MB per step).  Instead the queries ADD INTO ONE TENSOR EACH inside their own kernels
    (mh_field_bwd_fused's reduction launch with accumulate = 1; the brick kernel's flush is an atomic add anyway) and return None
    for token, beta and tables; `_PackOperands.backward` -- the node behind the token every query takes, which autograd runs only
    after ALL of them have run, None gradients included -- hands the finished sums to beta and the tables (its extra inputs) and
    maps the raw sum to the parameters.  Gradients that reach beta or a table along other paths are added by autograd as usual:
    the sums are complete when they are handed over, nothing is modified afterwards.

    A query joins only if it uses the very beta tensor and table parameters the operands were prepared with; any other query gets
    its own tensors as before.  State is keyed on the autograd graph task: a new backward pass starts empty, whatever the last
    one left behind."""

    def __init__(self, beta=None, tables=()):
        self.bound = beta is not None
        self.keep = (beta, tuple(tables))            # the ids below stay theirs for as long as the accumulator lives
        self.ids = (id(beta),) + tuple(id(t) for t in tables)
        self.task = None
        self.reset()

    def reset(self):
        self.raw = None                              # [raw_len + 1] running sum: weight / bias gradients, then d(loss)/d(beta)
        self.tab = [None] * (len(self.ids) - 1)      # running table-gradient sums
        self.gmax = None                             # pool of zeroed int32 pairs (max |feature gradient| words of the queries)
        self.gmax_used = 0

    def enter(self) -> bool:
        task = _GRAPH_TASK_ID()
        if task != self.task:
            self.reset()
            self.task = task
        return task >= 0           # -1: not inside a backward pass (a Function's backward called by hand): nobody would collect

    def joins(self, ids) -> bool:
        """ids = (id(beta), id(emb_s), id(emb_c) or id(None)) of a query: same beta, same tables (a colourless query: its one)"""
        return self.bound and ids[0] == self.ids[0] and ids[1] == self.ids[1] and (ids[2] == id(None) or ids[2:] == self.ids[2:])

    def gmax_words(self, dev):
        if self.gmax is None or self.gmax_used + 2 > self.gmax.numel():
            self.gmax, self.gmax_used = torch.zeros(64, dtype=torch.int32, device=dev), 0
        w = self.gmax[self.gmax_used:self.gmax_used + 2]
        self.gmax_used += 2
        return w

    def collect(self):
        """-> (raw or None, [table sums or None]) of the running backward pass; the accumulator is empty afterwards"""
        if _GRAPH_TASK_ID is None or self.task != _GRAPH_TASK_ID():
            self.reset()
        out = (self.raw, self.tab)
        self.reset()
        return out


class _PackOperands(torch.autograd.Function):
    """Natural (effective) weights + biases of the nets of one JointPacker -> the kernels' operands, ONCE per step:

        fwd pack  [A fragments | bias packs]         (non-differentiable: the kernels read them)
        bwd pack  [transposed A fragments]           (non-differentiable)
        token     [raw_len] uninitialised carrier    (differentiable)

    Every MLP call of the step takes the token as an input and returns `mh_mlp_wgrad`'s raw tile-order gradient
    (dW tiles | db tiles) as the token's gradient; autograd sums the calls' contributions (ONE add of one flat tensor per
    extra call instead of one per parameter per call) and this Function's backward maps the sum to the natural layout
    with one gather.  `zero_bias0`: the first-layer bias of the warp nets reaches its parameter through the per-frame
    bias0 (model.warp), so its slot in the raw gradient is dropped here."""

    @staticmethod
    def forward(ctx, jp, zero_bias0, b3, n_w, acc, notify, *params):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        ctx.notify = notify                   # callables run when this node's backward runs (model: the step cache went stale)
        # behind the n_w weights / biases: the tensors whose gradients the queries sum in `acc` (field nets: beta, the hash tables)
        params, extras = params[:n_w], params[n_w:]
        ctx.acc, ctx.n_extra = acc, len(extras)
        require_gpu(*params)
        weights, biases, o = [], [], 0
        for pk in jp.packers:
            weights.append([p.detach() for p in params[o:o + len(pk.specs)]])
            o += len(pk.specs)
        for pk in jp.packers:
            biases.append([p.detach() for p in params[o:o + len(pk.specs)]])
            o += len(pk.specs)
        assert o == len(params) == n_w
        flat = jp.flat(weights, biases)
        m = jp.on(flat.device)
        if b3:       # the four operand gathers as one (packing.JointPacker.all_index): sections are views of one buffer
            sect = jp.all_index()[1]
            gathered = flat[m["all"]]
            take = lambda k: gathered[sect[k][0]:sect[k][0] + sect[k][1]]
            fpack, bpack = take("fwd"), take("bwd")
        else:
            fpack, bpack = flat[m["fwd"]], flat[m["bwd"]]
        if b3:
            # bf16x3 forward fragments (csrc/mlp_b3.hip): the same weights gathered in the 32x32x16 fragment order, then cut
            # into [hi | mid | lo] bf16 planes per layer by one launch
            lib = _lib.load()
            w3 = torch.zeros((jp.fwd3_total_f4 + jp.bwd3_total_f4) * 4, device=flat.device)
            for key, layers, base in (("fwd3", jp.b3_layers, 0), ("bwd3", jp.b3T_layers, jp.fwd3_total_f4)):
                if key == "bwd3" and not jp.sliced_bwd_for(b3):
                    continue
                src = take(key)
                so, sp = _i32arr([l[0] for l in layers])
                no, np_ = _i32arr([l[1] for l in layers])
                do, dp = _i32arr([l[2] + base for l in layers])
                check(lib.mh_b3_slice(ptr(src), ptr(w3), len(layers), sp, np_, dp, stream()), "mh_b3_slice")
        else:
            w3 = fpack.new_empty(0)
        token = fpack.new_empty(jp.raw_len)
        ctx.jp, ctx.zero_bias0 = jp, zero_bias0
        ctx.mark_non_differentiable(fpack, bpack, w3)
        return fpack, bpack, w3, token

    @staticmethod
    def backward(ctx, _gf, _gb, _g3, g_token):
        for f in ctx.notify:
            f()
        n = ctx.jp.raw_len
        raw, tabs = ctx.acc.collect() if ctx.acc is not None else (None, [])
        g_extra = [None] * ctx.n_extra
        if raw is not None:                   # the joined queries' sum: [weights, biases | d beta]
            g_token = raw[:n] if g_token is None else g_token + raw[:n]
            g_extra = [raw[n].reshape(())] + list(tabs)
        if g_token is None:
            return (None,) * (6 + sum(2 * len(pk.specs) for pk in ctx.jp.packers)) + tuple(g_extra)
        nat_w, nat_b = ctx.jp.unpack_grads(g_token, zero_bias0=ctx.zero_bias0)
        flat = [g for net in nat_w for g in net] + [g for net in nat_b for g in net]
        return (None, None, None, None, None, None, *flat, *g_extra)


class MLPOperands:
    """Prepared operands of the warp nets (deform_net + topo_net) or of the field nets (sdf_net + color_net)."""

    def __init__(self, jp, fpack, bpack, w3, token, mode="", acc=None, notify=None):
        self.jp, self.fpack, self.bpack, self.token = jp, fpack, bpack, token
        self.notify = notify                               # list the pack node calls through when its backward runs
        self.acc = acc                                     # running gradient sums of the queries that share these operands
        self.mode = mode if w3.numel() else ""             # "b3": the sliced kernels the operands are cut for
        # slices per net (float32 storage, 4 floats per float4 unit), or None when the fp32-MFMA kernels serve
        self.w3 = [w3[4 * o:4 * (o + n)] for o, n in jp.w3] if w3.numel() else None
        self.wT3 = [w3[4 * (jp.fwd3_total_f4 + o):4 * (jp.fwd3_total_f4 + o + n)] for o, n in jp.wT3] if w3.numel() else None
        self.w = [jp.take(fpack, sl) for sl in jp.w]
        self.b = [jp.take(fpack, sl) for sl in jp.b]
        self.wT = [jp.take(bpack, sl) for sl in jp.wT]


def prepare_warp_operands(params_d: Sequence[torch.Tensor], params_t: Sequence[torch.Tensor], mode: Optional[str] = None) -> MLPOperands:
    """params_{d,t}: W0x [128,39], W1..W4 [128,128], W5 [n_out,128], b0 (its gradient travels through bias0), b1..b5.
    mode: the arithmetic the pack is cut for ("b3" / "f32"; None = the process default)."""
    jp = warp_joint_packer()
    flat = list(params_d[:6]) + list(params_t[:6]) + list(params_d[6:]) + list(params_t[6:])
    mode = _warp_mode(mode)
    notify = []
    return MLPOperands(jp, *_PackOperands.apply(jp, True, mode, len(flat), None, notify, *flat), mode=mode, notify=notify)


def prepare_field_operands(params: Sequence[torch.Tensor], mode: Optional[str] = None, beta=None, tables=()) -> MLPOperands:
    """params: Ws0 [64,73], Ws1, Ws2 [33,64], Wc0, Wc1, Wc2 [3,64], bs0, bs1, bs2, bc0, bc1, bc2 (natural, effective).
    beta (0-dim) and tables (the sdf and the colour hash table): when given, the field queries that are handed THESE tensors sum
    their beta / table / weight gradients in place and the pack hands the sums over once (_QueryAccumulator)."""
    jp = field_joint_packer()
    mode = _warp_mode(mode)    # the field FORWARD follows the mode; the fused backward picks its pack per pass (_field_wT)
    notify = []
    if beta is None:
        return MLPOperands(jp, *_PackOperands.apply(jp, False, mode, len(params), None, notify, *params), mode=mode, notify=notify)
    if len(tables) != 2:
        raise ValueError("prepare_field_operands: tables = (sdf table, colour table)")
    acc = _QueryAccumulator(beta, tables)
    return MLPOperands(jp, *_PackOperands.apply(jp, False, mode, len(params), acc, notify, *params, beta, *tables), mode=mode, acc=acc,
                       notify=notify)


class _WarpMLP(torch.autograd.Function):
    """deform_net + topo_net on [freq(x), per-slot code bias]  (model.py:412-437).

    bias0_{d,t} [n_slots,128] = code_slot @ W0[:,39:].T + b0, built by the caller in torch so that autograd carries the
    gradient on to the deform code, W0's code columns and b0.  `token` / `opnd`: see _PackOperands / MLPOperands.
    """

    @staticmethod
    def forward(ctx, x, slot, bias0_d, bias0_t, token, n_bands, opnd, slots_are_identity=False):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(x, bias0_d, bias0_t)
        lib = _lib.load()
        (wd, wt), (bd, bt), (wdT, wtT) = opnd.w, opnd.b, opnd.wT
        x = x.detach().contiguous().float()
        M, dev = x.shape[0], x.device
        need_grad = any(ctx.needs_input_grad)
        acts = _scratch(lib.mh_warp_acts_floats(M), dev) if need_grad else None
        deform, topo = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
        b0d, b0t = bias0_d.detach().contiguous(), bias0_t.detach().contiguous()
        slot_c = None if slot is None else slot.contiguous()
        _e = TIMER.start()
        if opnd.w3 is not None:
            check(lib.mh_warp_fwd_b3(ptr(x), ptr(slot_c), ptr(b0d), ptr(b0t), ptr(opnd.w3[0]), ptr(opnd.w3[1]), ptr(bd), ptr(bt),
                                     n_bands, ptr(deform), ptr(topo), ptr(acts), M, stream()), "mh_warp_fwd_b3")
        else:
            check(lib.mh_warp_fwd(ptr(x), ptr(slot_c), ptr(b0d), ptr(b0t), ptr(wd), ptr(wt), ptr(bd), ptr(bt), n_bands,
                                  ptr(deform), ptr(topo), ptr(acts), M, stream()), "mh_warp_fwd")
        TIMER.stop("mh_warp_fwd", _e)
        ctx.b3, ctx.mode = opnd.wT3 is not None, opnd.mode
        if ctx.b3:
            wdT, wtT = opnd.wT3
        ctx.save_for_backward(x, slot_c, wdT, wtT, acts)
        ctx.n_bands, ctx.n_slots, ctx.jp = n_bands, bias0_d.shape[0], opnd.jp
        # the caller SAYS when slot[i] == i (model._slots' one-slot-per-sample case); n_slots == M alone does not imply it
        # (B frames == M samples, or torch.unique's sorted inverse)
        ctx.identity = bool(slots_are_identity) and bias0_d.shape[0] == M
        return deform, topo

    @staticmethod
    def backward(ctx, g_deform, g_topo):
        lib = _lib.load()
        x, slot, wdT, wtT, acts = ctx.saved_tensors
        M, dev = x.shape[0], x.device
        n_tiles = lib.mh_mlp_tiles(M)
        dpre = _scratch(lib.mh_warp_dpre_floats(M), dev)
        g_x = torch.empty(M, 3, device=dev) if ctx.needs_input_grad[0] else None   # NULL -> the kernel skips the W0^T stage
        c = lambda t: None if t is None else t.contiguous()
        _e = TIMER.start()
        gd, gt = c(g_deform), c(g_topo)
        # b3, large batches: dPre4 (16 KB of a tile's 172) is not parked, the layer-4 weight-gradient launches make it again from
        # the incoming gradient, the ReLU sign words and the T5 slices -- the same bits (include/morpheus_hip.h: mh_warp_wgrad_b3)
        regen = bool(ctx.b3 and REGEN_DPRE4 and lib.mh_warp_regen_dpre4(M))
        if ctx.b3:
            check(lib.mh_warp_bwd_data_b3(ptr(x), ptr(gd), ptr(gt), ptr(wdT), ptr(wtT), ctx.n_bands, ptr(acts), ptr(dpre), ptr(g_x), M,
                                          int(regen), stream()), "mh_warp_bwd_data")
        else:
            check(lib.mh_warp_bwd_data(ptr(x), ptr(gd), ptr(gt), ptr(wdT), ptr(wtT), ctx.n_bands, ptr(acts), ptr(dpre), ptr(g_x), M,
                                       stream()), "mh_warp_bwd_data")
        TIMER.stop("mh_warp_bwd_data", _e)
        if ctx.b3:
            jp = ctx.jp
            ws = torch.empty(max(lib.mh_warp_wgrad_workspace_floats(M), 1), device=dev)
            dw_len = sum(i * o for i, o in zip(_WARP_WG[2], _WARP_WG[3]))
            raw_len = dw_len + sum(_WARP_WG[3])
            assert raw_len == jp.raw_len
            raw = torch.empty(raw_len, device=dev) if M > 0 else torch.zeros(raw_len, device=dev)      # (an empty call writes nothing)
            _e = TIMER.start()
            check(lib.mh_warp_wgrad_b3(ptr(acts), ptr(dpre), ptr(gd), ptr(gt), ptr(wdT), ptr(wtT), int(regen), ptr(ws), ptr(raw[:dw_len]),
                                       ptr(raw[dw_len:]), M, stream()), "mh_warp_wgrad_b3")
            TIMER.stop("mh_mlp_wgrad[warp]", _e)
        else:
            raw = _wgrad(lib, acts, dpre, WARP_ACT_ROWS * 32, WARP_DPRE_ROWS * 32, _WARP_WG[0], _WARP_WG[1], _WARP_WG[2],
                         _WARP_WG[3], n_tiles, dev, "warp", b3=False, in_live=_WARP_WG[4], out_live=_WARP_WG[5])
        # per-slot first-layer bias gradient: sum of dPre0 over the points of each slot
        if ctx.n_slots == 1:
            (od, ot) = ctx.jp.bias0_raw
            g_b0d, g_b0t = raw[od:od + 128][None], raw[ot:ot + 128][None]
        else:
            dp = dpre.view(n_tiles, WARP_DPRE_ROWS, 32)
            per_pt_d = dp[:, 0:128, :].permute(0, 2, 1).reshape(-1, 128)[:M]
            per_pt_t = dp[:, 672:800, :].permute(0, 2, 1).reshape(-1, 128)[:M]
            if ctx.identity:              # one slot per sample, slot[i] == i: the per-point rows ARE the answer
                g_b0d, g_b0t = per_pt_d, per_pt_t
            else:
                idx = slot.long()
                g_b0d = torch.zeros(ctx.n_slots, 128, device=dev).index_add_(0, idx, per_pt_d)
                g_b0t = torch.zeros(ctx.n_slots, 128, device=dev).index_add_(0, idx, per_pt_t)
        return (g_x, None, g_b0d, g_b0t, raw, None, None, None)


def _warp_wg_geometry():
    act_off, dpre_off, in_pad, out_pad, in_live, out_live = [], [], [], [], [], []
    for net in range(2):
        for l in range(6):
            act_off.append(0 if l == 0 else (64 + net * 640 + (l - 1) * 128) * 32)
            dpre_off.append((net * 672 + l * 128) * 32)
            in_pad.append(64 if l == 0 else 128)
            out_pad.append(32 if l == 5 else 128)
            # rows that carry values: 40 of H0's 64 (20 encoding k-steps x 2 halves), 3 | 2 of dPre5's 32 -- the b3 kernels write,
            # and every weight-gradient kernel reads, only those (csrc/mlp.hip: wg_row)
            in_live.append(40 if l == 0 else 128)
            out_live.append((3, 2)[net] if l == 5 else 128)
    return act_off, dpre_off, in_pad, out_pad, in_live, out_live


_WARP_WG = _warp_wg_geometry()

def warp_mlp(x, slot, bias0_d, bias0_t, n_bands, opnd: MLPOperands, slots_are_identity: bool = False):
    """slots_are_identity: slot[i] == i for every sample (bias0 rows are per sample): the first-layer bias gradient is then
    the per-point dPre0 rows themselves; any other slot map goes through index_add_."""
    return _WarpMLP.apply(x, slot, bias0_d, bias0_t, opnd.token, n_bands, opnd, slots_are_identity)


def _field_fwd(lib, xc, fs, fc, tp, beta_c, n_bands, with_color, opnd, need_grad):
    """-> sdf, sigma, albedo|None, acts|None (parked activations for backward)."""
    w, b = opnd.w[0], opnd.b[0]
    M, dev = xc.shape[0], xc.device
    acts = _scratch(lib.mh_field_acts_floats(M), dev) if need_grad else None
    sdf, sigma = torch.empty(M, device=dev), torch.empty(M, device=dev)
    albedo = torch.empty(M, 3, device=dev) if with_color else None
    _e = TIMER.start()
    if opnd.w3 is not None:
        check(lib.mh_field_fwd_b3(ptr(xc), ptr(fs), ptr(fc), ptr(tp), ptr(opnd.w3[0]), ptr(b), ptr(beta_c), n_bands,
                  int(bool(with_color)), ptr(sdf), ptr(sigma), ptr(albedo), ptr(acts), M, stream()), "mh_field_fwd_b3")
    else:
        check(lib.mh_field_fwd(ptr(xc), ptr(fs), ptr(fc), ptr(tp), ptr(w), ptr(b), ptr(beta_c), n_bands, int(bool(with_color)),
                               ptr(sdf), ptr(sigma), ptr(albedo), ptr(acts), M, stream()), "mh_field_fwd")
    TIMER.stop("mh_field_fwd", _e)
    return sdf, sigma, albedo, acts


# Arithmetic of the fused field backward in the b3 mode: the bf16 pipe like every other MLP kernel of the mode (mh_field_bwd_fused_b3),
# both passes -- colour + sdf (cfg3's: 2.13 -> 1.78 ms) and sdf-only (the finite-difference taps, 12 of every 13 points of a
# training step: virtual-view 72 x 72 2.56 -> 1.88 ms).  Same-box A/Bs: profiles/r04_ab_field_bwd_b3.txt.  The colour + sdf pass
# only won once the weight gradient of the sdf net's geo rows had moved into the colour launch: before, the sliced sdf launch
# spilled 91 registers beside its 192 accumulators and measured equal to the fp32 form.
# MORPHEUS_FIELD_BWD = "f32" keeps the native fp32 MFMA form (mh_field_bwd_fused; what the f32 mode always uses), "sdf" the
# sliced form for the sdf-only pass alone (A/B).
FIELD_BWD = os.environ.get("MORPHEUS_FIELD_BWD", "b3")
if FIELD_BWD not in ("b3", "sdf", "f32"):
    raise ValueError(f"MORPHEUS_FIELD_BWD={FIELD_BWD!r}: expected 'b3', 'sdf' or 'f32'")


def _field_wT(opnd, with_color: bool):
    """-> (transposed weight operand of the fused field backward, is it the bf16x3 pack?)"""
    sliced = FIELD_BWD == "b3" or (FIELD_BWD == "sdf" and not with_color)
    if sliced and opnd.mode == "b3" and opnd.wT3 is not None:
        return opnd.wT3[0], True
    return opnd.wT[0], False


def _field_bwd(lib, xc, wT, beta_c, acts, sdf, albedo, g_sdf, g_sigma, g_albedo, n_bands, with_color, has_topo, has_fc,
               need_dx, jp, raw_into=None, gmax=None, b3=False):
    """mh_field_bwd_fused: backward-data and weight gradients of the field nets in one pass per net (the pre-activation
    gradients stay on the chip).  raw_into: device ADDRESS of a running [raw_len + 1] sum this call adds its weight / bias /
    beta gradients to (the reduction launch adds instead of writing), or None -> a fresh tensor.  gmax: 2 zeroed int32 words.
    -> g_xc|None, g_fs, g_fc|None, g_tp|None, raw|None (tile-order weight gradient followed by d(loss)/d(beta); None when it
    went into raw_into), gmax (int32[2]: max|g_fs|, max|g_fc| as float bits, reduced on the fly by the kernel for the hash-grid
    backward's fixed point)."""
    M, dev = xc.shape[0], xc.device
    if not with_color:
        g_albedo = None
    g_xc = torch.empty(M, 3, device=dev) if need_dx else None
    g_fs = _scratch(M * 32, dev).view(M, 32)
    g_fc = _scratch(M * 32, dev).view(M, 32) if (with_color and has_fc) else None
    g_tp = torch.empty(M, 2, device=dev) if has_topo else None
    if gmax is None:
        gmax = torch.zeros(2, dtype=torch.int32, device=dev)
    dgeo = _scratch(lib.mh_field_dgeo_floats(M), dev) if with_color else None
    ws = torch.empty(lib.mh_field_bwd_fused_workspace_floats(M), device=dev)
    raw = None
    if raw_into is None:      # an empty query returns MH_OK without writing: the gradient must then be zeros, not heap contents
        raw = torch.empty(jp.raw_len + 1, device=dev) if M > 0 else torch.zeros(jp.raw_len + 1, device=dev)
    c = lambda t: None if t is None else t.contiguous()
    _e = TIMER.start()
    fused = lib.mh_field_bwd_fused_b3 if b3 else lib.mh_field_bwd_fused
    check(fused(ptr(xc), ptr(sdf), ptr(albedo if with_color else None), ptr(c(g_sdf)), ptr(c(g_sigma)), ptr(c(g_albedo)), ptr(wT),
                                 ptr(beta_c), n_bands, int(with_color), ptr(acts), ptr(dgeo), ptr(ws),
                                 ptr(raw) if raw_into is None else ctypes.c_void_p(raw_into), int(raw_into is not None), ptr(g_xc),
                                 ptr(g_fs), ptr(g_fc), ptr(g_tp), ptr(gmax), M, stream()), "mh_field_bwd_fused")
    TIMER.stop("mh_field_bwd_fused", _e)
    return g_xc, g_fs, g_fc, g_tp, raw, gmax


class _FieldMLP(torch.autograd.Function):
    """sdf_net (+Laplace density) and color_net on [freq(xc), hash, topo] / [hash_c, geo] with the hash features GIVEN
    (model.py:273-307, density.py:22-31).  The model itself goes through _FieldQuery (hash grids + nets in one node).

    beta: 0-dim device tensor = |beta_param| + 1e-4 (stays on the device: no host sync).  `token` / `opnd`: the
    prepared operands of prepare_field_operands.
    """

    @staticmethod
    def forward(ctx, xc, feat_s, feat_c, topo, beta, token, n_bands, with_color, opnd):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        require_gpu(xc, feat_s, feat_c, topo, beta)
        lib = _lib.load()
        xc = xc.detach().contiguous().float()
        fs = feat_s.detach().contiguous()
        fc = None if feat_c is None else feat_c.detach().contiguous()
        tp = None if topo is None else topo.detach().contiguous()
        beta_c = beta.detach().reshape(1).contiguous().float()
        sdf, sigma, albedo, acts = _field_fwd(lib, xc, fs, fc, tp, beta_c, n_bands, with_color, opnd, any(ctx.needs_input_grad))
        wT_sel, ctx.bwd_b3 = _field_wT(opnd, bool(with_color))
        ctx.save_for_backward(xc, wT_sel, beta_c, acts, sdf, albedo)
        ctx.cfg = (n_bands, bool(with_color), topo is not None, feat_c is not None)
        ctx.jp = opnd.jp
        if albedo is None:
            albedo = torch.empty(0, device=xc.device)        # (an empty allocation: no fill is dispatched)
            ctx.mark_non_differentiable(albedo)
        return sdf, sigma, albedo

    @staticmethod
    def backward(ctx, g_sdf, g_sigma, g_albedo):
        lib = _lib.load()
        xc, wT, beta_c, acts, sdf, albedo = ctx.saved_tensors
        n_bands, with_color, has_topo, has_fc = ctx.cfg
        g_xc, g_fs, g_fc, g_tp, raw, _ = _field_bwd(lib, xc, wT, beta_c, acts, sdf, albedo, g_sdf, g_sigma, g_albedo,
                                                    n_bands, with_color, has_topo, has_fc, ctx.needs_input_grad[0], ctx.jp,
                                                    b3=ctx.bwd_b3)
        n = ctx.jp.raw_len
        return (g_xc, g_fs, g_fc, g_tp, raw[n].reshape(()), raw[:n], None, None, None)


def field_mlp(xc, feat_s, feat_c, topo, beta, n_bands, with_color, opnd: MLPOperands):
    """-> sdf [M], sigma [M], albedo [M,3] (empty when with_color is False)."""
    return _FieldMLP.apply(xc, feat_s, feat_c, topo, beta, opnd.token, n_bands, with_color, opnd)


class _FieldQuery(torch.autograd.Function):
    """get_sigma_albedo (model.py:273-307) as ONE autograd node: hash-grid encode of the sdf (and colour) table at xc ->
    sdf_net + Laplace density (+ color_net).  Backward runs the field backward, then the brick backward of the tables with
    max|feature gradient| handed over directly (the field backward-data kernel reduces it on the fly; the brick kernel's
    fixed-point accumulation needs it) -- the features and their gradients never surface as autograd tensors."""

    @staticmethod
    def forward(ctx, xc, topo, beta, token, emb_s, emb_c, offsets_np, res_np, n_levels, bound, group, n_bands, opnd):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        with_color = emb_c is not None
        require_gpu(xc, topo, beta, emb_s, emb_c)
        lib = _lib.load()
        xc = xc.detach().contiguous().float()
        L = len(res_np)
        o_np, o_p = _i32arr(offsets_np)
        r_np, r_p = _i32arr(res_np)
        embs = [emb_s.detach().contiguous()] + ([emb_c.detach().contiguous()] if with_color else [])
        feats, ctx.binned = _grid_fwd(lib, xc, embs, o_p, r_p, L, n_levels, bound, group)
        tp = None if topo is None else topo.detach().contiguous()
        beta_c = beta.detach().reshape(1).contiguous().float()
        sdf, sigma, albedo, acts = _field_fwd(lib, xc, feats[0], feats[1] if with_color else None, tp, beta_c, n_bands,
                                              with_color, opnd, any(ctx.needs_input_grad))
        wT_sel, ctx.bwd_b3 = _field_wT(opnd, bool(with_color))
        ctx.save_for_backward(xc, wT_sel, beta_c, acts, sdf, albedo, *embs)
        ctx.cfg = (n_bands, with_color, topo is not None, o_np, r_np, n_levels, float(bound), L)
        ctx.jp, ctx.acc = opnd.jp, opnd.acc
        ctx.ids = (id(beta), id(emb_s), id(emb_c))       # identities of the shared inputs (see _QueryAccumulator.joins)
        if albedo is None:
            albedo = torch.empty(0, device=xc.device)        # (an empty allocation: no fill is dispatched)
            ctx.mark_non_differentiable(albedo)
        return sdf, sigma, albedo

    @staticmethod
    def backward(ctx, g_sdf, g_sigma, g_albedo):
        lib = _lib.load()
        xc, wT, beta_c, acts, sdf, albedo, *embs = ctx.saved_tensors
        n_bands, with_color, has_topo, o_np, r_np, n_levels, bound, L = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        acc, n = ctx.acc, ctx.jp.raw_len
        # joined: this query's weight / beta / table gradients go into the pass's running sums (handed over by _PackOperands.backward)
        joined = (ACCUMULATE_IN_PLACE and acc is not None and xc.shape[0] > 0 and ctx.needs_input_grad[3] and acc.joins(ctx.ids)
                  and acc.enter())
        g_xc, g_fs, g_fc, g_tp, raw, gmax = _field_bwd(
            lib, xc, wT, beta_c, acts, sdf, albedo, g_sdf, g_sigma, g_albedo, n_bands, with_color, has_topo, with_color, need_dx, ctx.jp,
            raw_into=acc.raw.data_ptr() if joined and acc.raw is not None else None, gmax=acc.gmax_words(xc.device) if joined else None,
            b3=ctx.bwd_b3)
        o_p, r_p = o_np.ctypes.data_as(ctypes.c_void_p), r_np.ctypes.data_as(ctypes.c_void_p)
        grads = [g_fs] + ([g_fc] if with_color else [])
        gptrs = [gmax.data_ptr(), gmax.data_ptr() + 4][:len(grads)]
        sums = [acc.tab[k].data_ptr() if acc.tab[k] is not None else None for k in range(len(grads))] if joined else None
        g_x, g_embs = _grid_bwd(lib, xc, embs, grads, o_p, r_p, L, n_levels, bound, need_dx, gptrs, sums=sums, g_x_into=g_xc,
                                binned=ctx.binned)
        if joined:
            if raw is not None:
                acc.raw = raw
            for k, g in enumerate(g_embs):
                if g is not None:
                    acc.tab[k] = g
            return (g_x, g_tp) + (None,) * 11
        return (g_x, g_tp, raw[n].reshape(()), raw[:n], g_embs[0], g_embs[1] if with_color else None, None, None, None, None, None, None, None)


def field_query(xc, topo, beta, emb_s, emb_c, offsets_np, res_np, bound, max_level, group, n_bands, opnd: MLPOperands):
    """-> sdf [M], sigma [M], albedo [M,3] (empty when emb_c is None: the sdf-only pass of the finite-difference taps)."""
    n_levels = effective_levels(max_level, len(res_np))
    return _FieldQuery.apply(xc, topo, beta, opnd.token, emb_s, emb_c, offsets_np, res_np, n_levels, float(bound), int(group),
                             n_bands, opnd)


class _WeightNormAll(torch.autograd.Function):
    """W_l = g_l * v_l / ||v_l||_row for a list of weight-normed layers (decoders.py:51-52), one launch forward and one
    backward (csrc/wnorm.hip) instead of norm / divide / multiply and their autograd graph per layer."""

    @staticmethod
    def forward(ctx, n, *vg):
        ctx.set_materialize_grads(False)      # unused outputs hand backward None, not a zero-filled tensor (one launch each)
        vs, gs = vg[:n], vg[n:]
        require_gpu(*vg)
        lib = _lib.load()
        vs_c = [v.detach().contiguous() for v in vs]
        gs_c = [g.detach().contiguous() for g in gs]
        ws = [torch.empty_like(v) for v in vs_c]
        PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
        rows, cols = IA(*[v.shape[0] for v in vs_c]), IA(*[v.shape[1] for v in vs_c])
        check(lib.mh_weight_norm_fwd(n, PA(*[ptr(v) for v in vs_c]), PA(*[ptr(g) for g in gs_c]), PA(*[ptr(w) for w in ws]),
                                     rows, cols, stream()), "mh_weight_norm_fwd")
        ctx.save_for_backward(*vs_c, *gs_c)
        ctx.n = n
        return tuple(ws)

    @staticmethod
    def backward(ctx, *gw):
        n = ctx.n
        lib = _lib.load()
        saved = ctx.saved_tensors
        vs, gs = saved[:n], saved[n:]
        gw_c = [None if g is None else g.contiguous() for g in gw]
        dvs = [torch.empty_like(v) for v in vs]
        dgs = [torch.empty_like(g) for g in gs]
        PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
        rows, cols = IA(*[v.shape[0] for v in vs]), IA(*[v.shape[1] for v in vs])
        check(lib.mh_weight_norm_bwd(n, PA(*[ptr(v) for v in vs]), PA(*[ptr(g) for g in gs]), PA(*[ptr(g) for g in gw_c]),
                                     PA(*[ptr(t) for t in dvs]), PA(*[ptr(t) for t in dgs]), rows, cols, stream()),
              "mh_weight_norm_bwd")
        return (None, *dvs, *dgs)


def weight_norm_all(vs: Sequence[torch.Tensor], gs: Sequence[torch.Tensor]):
    """Effective weights of weight-normed layers: vs[l] [out,in], gs[l] [out,1] -> list of [out,in]."""
    assert len(vs) == len(gs) and all(v.dim() == 2 for v in vs)
    return list(_WeightNormAll.apply(len(vs), *vs, *gs))


# ------------------------------------------------------------------------------------------ caller-side glue, one launch each
MEAN_KINDS = {"identity": 0, "square": 1, "abs": 2, "entropy": 3, "eikonal": 4, "absdiff": 5, "sqdiff": 6}


class _MaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, a, b, w_row, n_valid):
        ctx.set_materialize_grads(False)
        require_gpu(a, b, w_row, n_valid)
        lib = _lib.load()
        a_c = a.detach().contiguous().float()
        b_c = None if b is None else b.detach().contiguous().float()
        assert a_c.dim() >= 1 and (b_c is None or b_c.shape == a_c.shape)
        M = a_c.shape[0]
        C = max(a_c.numel() // max(M, 1), 1)
        w_c = None if w_row is None else w_row.detach().reshape(-1).contiguous().float()
        assert w_c is None or w_c.numel() == M
        nv = None if n_valid is None else n_valid.detach().reshape(1).to(torch.int32).contiguous()
        ws = torch.empty(lib.mh_masked_mean_workspace_floats(), device=a_c.device)
        out = torch.empty(2, device=a_c.device)
        check(lib.mh_masked_mean_fwd(kind, ptr(a_c), ptr(b_c), ptr(w_c), M, C, ptr(nv), ptr(ws), ptr(out), stream()), "mh_masked_mean_fwd")
        ctx.save_for_backward(a_c, b_c, w_c, nv, out)
        ctx.kind, ctx.shape = kind, a.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        lib = _lib.load()
        a_c, b_c, w_c, nv, out = ctx.saved_tensors
        M = a_c.shape[0]
        C = max(a_c.numel() // max(M, 1), 1)
        want_a, want_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2] and b_c is not None
        g_a = torch.empty_like(a_c) if want_a else None
        g_b = torch.empty_like(b_c) if want_b else None
        if want_a or want_b:
            check(lib.mh_masked_mean_bwd(ctx.kind, ptr(a_c), ptr(b_c), ptr(w_c), M, C, ptr(nv), ptr(out), ptr(g.reshape(1).contiguous().float()),
                                         ptr(g_a), ptr(g_b), stream()), "mh_masked_mean_bwd")
        return None, g_a, g_b, None, None


def masked_mean(kind: str, a, b=None, *, n_valid=None, row_weight=None):
    """Mean over the samples of f(a[, b]) for a per-sample tensor [M, ...] (include/morpheus_hip.h: mh_masked_mean_*): the
    reference's `f(a).mean()` when every row counts; with n_valid (0-dim device int tensor, fixed-capacity sampling) the rows behind
    it are left out; with row_weight [M] the value is sum(f * w) / max(per_row * sum(w), 1) (morpheus.py:556).  kind: see
    MEAN_KINDS ("eikonal": a is [M,3], f = (|row| - 1)^2; "absdiff" / "sqdiff": f(a - b))."""
    return _MaskedMean.apply(MEAN_KINDS[kind], a, b, row_weight, n_valid)


class _OrthoPerturb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, normals, phi, scale):
        ctx.set_materialize_grads(False)
        require_gpu(x, normals, phi)
        lib = _lib.load()
        x_c, n_c = x.detach().contiguous().float(), normals.detach().contiguous().float()
        p_c = phi.detach().reshape(-1).contiguous().float()
        M = x_c.shape[0]
        assert x_c.shape == (M, 3) and n_c.shape == (M, 3) and p_c.numel() == M
        out = torch.empty_like(x_c)
        check(lib.mh_ortho_perturb_fwd(ptr(x_c), ptr(n_c), ptr(p_c), float(scale), M, ptr(out), stream()), "mh_ortho_perturb_fwd")
        ctx.save_for_backward(n_c, p_c)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        n_c, p_c = ctx.saved_tensors
        g_n = None
        if ctx.needs_input_grad[1]:
            lib = _lib.load()
            g_c = g.contiguous().float()
            g_n = torch.empty_like(n_c)
            check(lib.mh_ortho_perturb_bwd(ptr(n_c), ptr(p_c), ptr(g_c), ctx.scale, n_c.shape[0], ptr(g_n), stream()), "mh_ortho_perturb_bwd")
        return (g if ctx.needs_input_grad[0] else None), g_n, None, None


def ortho_perturb(x, normals, phi, scale: float):
    """x + scale * get_ortho_normal_dir(normals)  (morpheus.py:518-528 applied at :549 / :766) in one launch; phi [M] or [M,1] is the
    caller's uniform draw times 2 pi.  Gradients reach x (identity) and the normals."""
    return _OrthoPerturb.apply(x, normals, phi, scale)


class _SmoothPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, off, rays_o, rays_d):
        ctx.set_materialize_grads(False)
        require_gpu(depth, off, rays_o, rays_d)
        lib = _lib.load()
        d_c, f_c = depth.detach().reshape(-1).contiguous().float(), off.detach().reshape(-1).contiguous().float()
        o_c, r_c = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
        N, K = d_c.shape[0], f_c.shape[0]
        assert o_c.shape == (N, 3) and r_c.shape == (N, 3)
        pts, keep = torch.empty(K * N, 3, device=d_c.device), torch.empty(K * N, device=d_c.device)
        check(lib.mh_smooth_points_fwd(ptr(d_c), ptr(f_c), ptr(o_c), ptr(r_c), N, K, ptr(pts), ptr(keep), stream()), "mh_smooth_points_fwd")
        ctx.save_for_backward(d_c, f_c, r_c)
        ctx.depth_shape = depth.shape
        ctx.mark_non_differentiable(keep)
        return pts, keep

    @staticmethod
    def backward(ctx, g, _gk):
        if g is None:
            return None, None, None, None
        d_c, f_c, r_c = ctx.saved_tensors
        N, K = d_c.shape[0], f_c.shape[0]
        need_depth, _, need_o, need_d = ctx.needs_input_grad
        g_depth = torch.empty(N, device=g.device) if need_depth else None
        g_o = torch.empty(N, 3, device=g.device) if need_o else None
        g_d = torch.empty(N, 3, device=g.device) if need_d else None
        if need_depth or need_o or need_d:
            check(_lib.load().mh_smooth_points_bwd(ptr(g.contiguous().float()), ptr(d_c), ptr(f_c), ptr(r_c), N, K, ptr(g_depth), ptr(g_o),
                                                   ptr(g_d), stream()), "mh_smooth_points_bwd")
        return (None if g_depth is None else g_depth.view(ctx.depth_shape)), None, g_o, g_d


def smooth_points(depth, off, rays_o, rays_d):
    """-> pts [K*N,3] = (depth[n] + off[k]) * rays_d[n] + rays_o[n] (point k*N + n), keep [K*N] = (|pts| < 1.1) as 0 / 1: the points
    of get_normal_smoothness_loss (morpheus.py:530-547) in one launch each way; gradients to depth, rays_o, rays_d."""
    return _SmoothPoints.apply(depth, off, rays_o, rays_d)


class _BgBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, opacity, bg):
        ctx.set_materialize_grads(False)
        require_gpu(color, opacity, bg)
        c_c, o_c, b_c = color.detach().contiguous().float(), opacity.detach().reshape(-1).contiguous().float(), bg.detach().contiguous().float()
        N = c_c.shape[0]
        assert c_c.shape == (N, 3) and b_c.shape == (N, 3) and o_c.shape[0] == N
        image = torch.empty_like(c_c)
        check(_lib.load().mh_bg_blend_fwd(ptr(c_c), ptr(o_c), ptr(b_c), N, ptr(image), stream()), "mh_bg_blend_fwd")
        ctx.save_for_backward(o_c, b_c)
        ctx.opacity_shape = opacity.shape
        return image

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        o_c, b_c = ctx.saved_tensors
        N = o_c.shape[0]
        need_c, need_o, need_b = ctx.needs_input_grad
        g_c = g.contiguous().float()
        g_o = torch.empty(N, device=g.device) if need_o else None
        g_b = torch.empty(N, 3, device=g.device) if need_b else None
        if need_o or need_b:
            check(_lib.load().mh_bg_blend_bwd(ptr(g_c), ptr(o_c), ptr(b_c), N, ptr(g_o), ptr(g_b), stream()), "mh_bg_blend_bwd")
        return (g_c if need_c else None), (None if g_o is None else g_o.view(ctx.opacity_shape)), g_b


def bg_blend(color, opacity, bg):
    """color + (1 - opacity) * bg for a per-ray background [N,3] (morpheus.py:686-694) in one launch each way, the chain's rounding."""
    return _BgBlend.apply(color, opacity, bg)


class _PoseApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, pose, frame_of_row, n_per_row):
        ctx.set_materialize_grads(False)
        require_gpu(rays_o, rays_d, pose, frame_of_row)
        lib = _lib.load()
        o_c, d_c = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
        p_c = pose.detach().contiguous().float()
        f_c = frame_of_row.detach().reshape(-1).long().contiguous()
        B = f_c.numel()
        assert o_c.shape == (B * n_per_row, 3) and d_c.shape == o_c.shape and p_c.dim() == 2 and p_c.shape[1] == 6
        o_out, d_out = torch.empty_like(o_c), torch.empty_like(d_c)
        check(lib.mh_pose_apply_fwd(ptr(o_c), ptr(d_c), ptr(p_c), ptr(f_c), B, n_per_row, ptr(o_out), ptr(d_out), stream()), "mh_pose_apply_fwd")
        ctx.save_for_backward(d_c, p_c, f_c)
        ctx.n_per_row = n_per_row
        return o_out, d_out

    @staticmethod
    def backward(ctx, g_o, g_d):
        d_c, p_c, f_c = ctx.saved_tensors
        if (g_o is None and g_d is None) or not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        lib = _lib.load()
        B = f_c.numel()
        g_o = None if g_o is None else g_o.contiguous().float()
        g_d = None if g_d is None else g_d.contiguous().float()
        ws = torch.empty(max(lib.mh_pose_bwd_workspace_floats(B, ctx.n_per_row), 1), device=d_c.device)
        g_pose = torch.empty_like(p_c)
        check(lib.mh_pose_apply_bwd(ptr(d_c), ptr(p_c), ptr(f_c), B, ctx.n_per_row, p_c.shape[0], ptr(g_o), ptr(g_d), ptr(ws), ptr(g_pose),
                                    stream()), "mh_pose_apply_bwd")
        return None, None, g_pose, None, None


def pose_apply(rays_o, rays_d, pose, frame_of_row, n_per_row: int):
    """(rays_o + t_f, R_f rays_d) for a batch of B rows of n_per_row rays, one frame per row (include/morpheus_hip.h:
    mh_pose_apply_*; models/pose.py:4-64 + model.py:335-346): pose [n_frames, 6], frame_of_row [B] frame ids.  The rays are data
    (no gradient); the gradient reaches `pose`."""
    return _PoseApply.apply(rays_o, rays_d, pose, frame_of_row, int(n_per_row))


class _RenderLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_rgb, pred_depth, opacity, image, depth, mask, bg, rays_o, rays_d, w_rgb, w_mask, w_depth):
        ctx.set_materialize_grads(False)
        require_gpu(pred_rgb, pred_depth, opacity, image, depth, mask, bg, rays_o, rays_d)
        lib = _lib.load()
        c = lambda t, *shape: t.detach().reshape(*shape).contiguous().float()
        N = pred_depth.numel()
        pr, pd, op = c(pred_rgb, N, 3), c(pred_depth, N), c(opacity, N)
        im, dp, mk, bgc = c(image, 3, N), c(depth, N), c(mask, N), c(bg, N, 3)
        ro, rd = c(rays_o, N, 3), c(rays_d, N, 3)
        gt_rgb, valid, out = torch.empty(3, N, device=pr.device), torch.empty(N, device=pr.device), torch.empty(4, device=pr.device)
        check(lib.mh_render_loss_fwd(ptr(pr), ptr(pd), ptr(op), ptr(im), ptr(dp), ptr(mk), ptr(bgc), ptr(ro), ptr(rd), N, w_rgb, w_mask,
                                     w_depth, ptr(gt_rgb), ptr(valid), ptr(out), stream()), "mh_render_loss_fwd")
        ctx.save_for_backward(pr, pd, op, gt_rgb, dp, mk, valid)
        ctx.w = (float(w_rgb), float(w_mask), float(w_depth))
        ctx.shapes = (pred_rgb.shape, pred_depth.shape, opacity.shape)
        terms = out[1:]
        ctx.mark_non_differentiable(terms, gt_rgb, valid)     # the three terms are VALUES for logging: the gradient travels through out[0] only
        return out[0], terms, gt_rgb, valid

    @staticmethod
    def backward(ctx, g, _g_terms, _g_gt, _g_valid):
        if g is None:
            return (None,) * 12
        lib = _lib.load()
        pr, pd, op, gt_rgb, dp, mk, valid = ctx.saved_tensors
        N = pd.numel()
        need = ctx.needs_input_grad
        g_rgb = torch.empty_like(pr) if need[0] else None
        g_dep = torch.empty_like(pd) if need[1] else None
        g_op = torch.empty_like(op) if need[2] else None
        if need[0] or need[1] or need[2]:
            check(lib.mh_render_loss_bwd(ptr(pr), ptr(pd), ptr(op), ptr(gt_rgb), ptr(dp), ptr(mk), ptr(valid), N, *ctx.w,
                                         ptr(g.reshape(1).contiguous().float()), ptr(g_rgb), ptr(g_dep), ptr(g_op), stream()),
                  "mh_render_loss_bwd")
        s = ctx.shapes
        return (None if g_rgb is None else g_rgb.view(s[0]), None if g_dep is None else g_dep.view(s[1]),
                None if g_op is None else g_op.view(s[2]), None, None, None, None, None, None, None, None, None)


def real_view_render_loss(pred_rgb, pred_depth, opacity, image, depth, mask, bg, rays_o, rays_d, w_rgb, w_mask, w_depth):
    """get_gt_from_data + get_real_view_render_loss (morpheus.py:930-983) in one launch each way (include/morpheus_hip.h:
    mh_render_loss_*).  pred_rgb [..., 3] per ray (the renderer's `image`), pred_depth / opacity per ray; image: the dataset's
    [B,3,H,W] batch (B = 1: channel-major [3,N]); bg [N,3].  -> (weighted loss, the three terms [rgb, mask, depth] (detached
    values), gt_rgb [3,N], valid-depth mask [N])."""
    return _RenderLoss.apply(pred_rgb, pred_depth, opacity, image, depth, mask, bg, rays_o, rays_d, float(w_rgb), float(w_mask),
                             float(w_depth))


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.set_materialize_grads(False)
        stacked = torch.stack([t.detach().reshape(()) for t in terms])
        ctx.save_for_backward(weights)
        ctx.n = len(terms)
        return (stacked * weights).sum()

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * (ctx.n + 1)
        (weights,) = ctx.saved_tensors
        gs = (weights * g).unbind(0)
        return (None, *[gs[k] if ctx.needs_input_grad[k + 1] else None for k in range(ctx.n)])


_WEIGHT_CACHE: dict = {}


def weighted_sum(pairs):
    """sum_k w_k * term_k for 0-dim device tensors term_k and host floats w_k: three launches forward (stack, multiply, add) and
    one backward, whatever the number of terms -- the reference's `loss = loss + w * term` chains (morpheus.py:946-1145) cost two
    launches per term each way.  The weight vector is built once per distinct tuple of weights (no host->device copy per step:
    that would be a sync in the eager step and illegal inside a captured graph)."""
    pairs = [(float(w), t) for w, t in pairs]
    if not pairs:
        return 0
    dev = pairs[0][1].device
    key = (tuple(w for w, _ in pairs), str(dev))
    wt = _WEIGHT_CACHE.get(key)
    if wt is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("weighted_sum: a new set of loss weights inside a captured graph (run the step once eagerly first)")
        if len(_WEIGHT_CACHE) >= 64:        # weights that change every step do not belong here (multiply the term instead)
            _WEIGHT_CACHE.clear()
        wt = _WEIGHT_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
    return _WeightedSum.apply(wt, *[t for _, t in pairs])


# ------------------------------------------------------------------------------------------ HIP-graph hygiene
def graph_memset_nodes(graph: "torch.cuda.CUDAGraph"):
    """(nodes, memset nodes, smallest memset in bytes) of a captured, not yet instantiated graph (`CUDAGraph(keep_graph=True)`)."""
    lib = _lib.load()
    n, m, b = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    check(lib.mh_graph_count_memset_nodes(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(n), ctypes.byref(m), ctypes.byref(b)),
          "mh_graph_count_memset_nodes")
    return n.value, m.value, b.value


def graph_replace_memset_nodes(graph: "torch.cuda.CUDAGraph") -> int:
    """Every memset node of the captured graph becomes a fill-kernel node with the same edges (csrc/graph.hip: small memset
    nodes replay wrongly on ROCm 7.2, and PyTorch's multi-block reductions memset their semaphores).  Call between the end of
    the capture and the first replay, on a graph made with `torch.cuda.CUDAGraph(keep_graph=True)`.  -> nodes replaced."""
    lib = _lib.load()
    n = ctypes.c_int64()
    check(lib.mh_graph_replace_memset_nodes(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(n)), "mh_graph_replace_memset_nodes")
    return n.value
