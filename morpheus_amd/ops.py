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
    stream, which is the one handed to the library).  Used by bench.py for the roofline numbers."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self, enabled: bool):
        self.enabled, self.records = enabled, {}

    def start(self):
        if not self.enabled:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name: str, e0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.setdefault(name, []).append((e0, e1))

    def summary(self):
        """name -> (calls, total_ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.records.items()}


TIMER = KernelTimer()


def _i32arr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------ hash grid
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
    perm = torch.empty(M, dtype=torch.int32, device=dev)
    bstart = torch.empty(lib.mh_grid_bin_index_ints(), dtype=torch.int32, device=dev)
    _e = TIMER.start()
    check(lib.mh_grid_bin_points(ptr(x), M, bound, ptr(ws), ptr(perm), ptr(bstart), stream()), "mh_grid_bin_points")
    TIMER.stop("mh_grid_bin_points", _e)
    return perm, bstart


class _GradMaxHint:
    """max |grad| of a feature-gradient tensor, computed on the fly by the kernel that produced it (mh_field_bwd_data)
    and handed to the hash-grid backward, which needs it for its fixed-point accumulation -- autograd only carries the
    tensor, so the device word travels beside it keyed by the tensor's address; a miss just means the grid backward
    reduces max |grad| itself (one more pass over the gradient)."""

    def __init__(self):
        self._by_ptr = {}

    def put(self, tensor, words, index):
        if tensor is not None:
            self._by_ptr = {k: v for k, v in self._by_ptr.items() if v[0] is words}   # keep only the current producer's
            self._by_ptr[tensor.data_ptr()] = (words, index, tensor.numel(), torch._C._current_graph_task_id())

    def take(self, grad):
        """device address of the word, or None; valid only inside the backward pass that produced the hint"""
        hit = self._by_ptr.pop(grad.data_ptr(), None)
        if hit is None or hit[2] != grad.numel() or hit[3] != torch._C._current_graph_task_id() or hit[3] < 0:
            return None
        return hit[0].data_ptr() + 4 * hit[1]


_GMAX = _GradMaxHint()


GRID_BWD_NAIVE = os.environ.get("MORPHEUS_GRID_BWD", "") == "naive"   # A/B switch: per-point global atomics


class _GridEncode(torch.autograd.Function):
    """grid.py:25-96 (_grid_encode) on the HIP kernels, for one or several tables evaluated at the
    same points (sdf + colour encoders share x: one index computation pattern, one binning).
    No dy_dx tensor is materialised; backward recomputes corner weights."""

    @staticmethod
    def forward(ctx, x, offsets_np, res_np, n_levels, bound, group, *embs):
        require_gpu(x, *embs)
        lib = _lib.load()
        x = x.detach().contiguous().float()
        M, L = x.shape[0], len(res_np)
        o_np, o_p = _i32arr(offsets_np)
        r_np, r_p = _i32arr(res_np)
        outs, saved = [], []
        for emb in embs:
            embc = emb.detach().contiguous()
            out = torch.empty(M, L * 2, device=x.device, dtype=torch.float32)
            _e = TIMER.start()
            check(lib.mh_grid_encode_fwd(ptr(x), ptr(embc), o_p, r_p, ptr(out), M, L, n_levels, float(bound), int(group),
                                         stream()),
                  "mh_grid_encode_fwd")
            TIMER.stop("mh_grid_encode_fwd", _e)
            outs.append(out)
            saved.append(embc)
        ctx.save_for_backward(x, *saved)
        ctx.meta = (o_np, r_np, n_levels, float(bound), L)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        x, *embs = ctx.saved_tensors
        o_np, r_np, n_levels, bound, L = ctx.meta
        o_p, r_p = o_np.ctypes.data_as(ctypes.c_void_p), r_np.ctypes.data_as(ctypes.c_void_p)
        M = x.shape[0]
        need_dx = ctx.needs_input_grad[0]
        g_x_total, g_embs = None, []
        binned = None
        for emb, grad in zip(embs, grads):
            if grad is None:
                g_embs.append(None)
                continue
            grad = grad.contiguous()
            g_emb = torch.zeros_like(emb)
            g_x = torch.empty_like(x) if need_dx else None
            if M > 0 and not GRID_BWD_NAIVE and L == 16:
                if binned is None:
                    binned = _bin_points(lib, x, bound)
                _e = TIMER.start()
                check(lib.mh_grid_encode_bwd_binned(ptr(grad), ptr(x), ptr(emb), o_p, r_p, ptr(binned[0]), ptr(binned[1]),
                                                    ptr(g_emb), ptr(g_x), M, L, n_levels, bound, _GMAX.take(grad), stream()),
                      "mh_grid_encode_bwd_binned")
                TIMER.stop("mh_grid_encode_bwd_binned", _e)
            else:
                _e = TIMER.start()
                check(lib.mh_grid_encode_bwd(ptr(grad), ptr(x), ptr(emb), o_p, r_p, ptr(g_emb), ptr(g_x), M, L, n_levels,
                                             bound, stream()), "mh_grid_encode_bwd")
                TIMER.stop("mh_grid_encode_bwd", _e)
            g_embs.append(g_emb)
            if need_dx:
                g_x_total = g_x if g_x_total is None else g_x_total + g_x
        return (g_x_total, None, None, None, None, None, *g_embs)


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
    def forward(ctx, sigma, t_starts, t_ends, rgb, ray_start, ray_cnt):
        require_gpu(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt)
        lib = _lib.load()
        sigma, t_starts, t_ends = sigma.detach().contiguous(), t_starts.contiguous(), t_ends.contiguous()
        rgbc = rgb.detach().contiguous()
        N, M = ray_start.shape[0], sigma.shape[0]
        dev = sigma.device
        weights = torch.empty(M, device=dev)
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
        d_sigma = torch.empty_like(sigma)
        d_rgb = torch.empty_like(rgb)
        _e = TIMER.start()
        check(lib.mh_composite_bwd(ptr(sigma), ptr(ts), ptr(te), ptr(rgb), ptr(ray_start), ptr(ray_cnt), ptr(weights),
                                   ptr(c(g_w)), ptr(c(g_o)), ptr(c(g_d)), ptr(c(g_c)), ptr(d_sigma), ptr(d_rgb), N,
                                   stream()), "mh_composite_bwd")
        TIMER.stop("mh_composite_bwd", _e)
        return d_sigma, None, None, d_rgb, None, None


def composite(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt):
    """-> weights [M], opacity [N], depth [N], color [N,3]  (morpheus.py:675-685)."""
    return _Composite.apply(sigma, t_starts, t_ends, rgb, ray_start, ray_cnt)


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
    o, d, j = rays_o.detach().contiguous(), rays_d.detach().contiguous(), jitter.contiguous()
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


MARCH_TWO_PASS = os.environ.get("MORPHEUS_MARCH", "") == "two_pass"   # A/B switch: thread-per-ray count + fill


def march_rays(rays_o, rays_d, jitter, step: float, bound: float, binary: torch.Tensor):
    """Occupancy-grid marcher -> (ray_idx int32 [M], t_starts [M], t_ends [M], ray_start [N], ray_cnt [N]).
    One host sync on the path: M = total sample count sizes the packed arrays (nerfacc synchronises at the same point).
    Single pass, one wavefront per ray (mh_march_slots + mh_march_pack); a batch with rays that overflow the per-ray slot
    row (directions much shorter than unit length) is re-run through the un-capped thread-per-ray count/fill pair."""
    require_gpu(rays_o, rays_d, jitter, binary)
    lib = _lib.load()
    o, d = rays_o.detach().contiguous(), rays_d.detach().contiguous()
    j = None if jitter is None else jitter.contiguous()
    assert binary.dtype == torch.uint8 and binary.is_contiguous() and binary.dim() == 3
    N, R, dev = o.shape[0], binary.shape[0], o.device
    if N == 0:
        e = torch.empty(0, device=dev)
        z = torch.empty(0, dtype=torch.int32, device=dev)
        return z, e, e.clone(), z.clone(), z.clone()
    if not MARCH_TWO_PASS:
        cap = int(lib.mh_march_cap(float(step), float(bound)))
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
            ri = torch.empty(M, dtype=torch.int32, device=dev)
            ts, te = torch.empty(M, device=dev), torch.empty(M, device=dev)
            if M > 0:
                _e = TIMER.start()
                check(lib.mh_march_pack(ptr(start), ptr(cnt), ptr(slots[0]), ptr(slots[1]), N, cap, ptr(ri), ptr(ts), ptr(te),
                                        stream()), "mh_march_pack")
                TIMER.stop("mh_march_pack", _e)
            return ri, ts, te, start, cnt.contiguous()
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    _e = TIMER.start()
    check(lib.mh_march_count(ptr(o), ptr(d), ptr(j), N, float(step), float(bound), R, ptr(binary), ptr(cnt), stream()),
          "mh_march_count")
    TIMER.stop("mh_march_count", _e)
    csum = torch.cumsum(cnt, 0, dtype=torch.int32)
    start = (csum - cnt).contiguous()
    M = int(csum[-1].item())
    ri = torch.empty(M, dtype=torch.int32, device=dev)
    ts, te = torch.empty(M, device=dev), torch.empty(M, device=dev)
    if M > 0:
        _e = TIMER.start()
        check(lib.mh_march_fill(ptr(o), ptr(d), ptr(j), N, float(step), float(bound), R, ptr(binary), ptr(start), ptr(ri),
                                ptr(ts), ptr(te), stream()), "mh_march_fill")
        TIMER.stop("mh_march_fill", _e)
    return ri, ts, te, start, cnt


# ------------------------------------------------------------------------------------ fused MLPs
def _wgrad(lib, acts, dpre, acts_tile, dpre_tile, act_off, dpre_off, in_pad, out_pad, n_tiles, dev, tag):
    n_layers = len(act_off)
    dw_len = int(sum(i * o for i, o in zip(in_pad, out_pad)))
    db_len = int(sum(out_pad))
    a_np, a_p = _i32arr(act_off)
    d_np, d_p = _i32arr(dpre_off)
    i_np, i_p = _i32arr(in_pad)
    o_np, o_p = _i32arr(out_pad)
    ws = torch.empty(lib.mh_mlp_wgrad_workspace_floats(n_layers, i_p, o_p, n_tiles), device=dev)
    raw = torch.empty(dw_len + db_len, device=dev)
    dw_raw, db_raw = raw[:dw_len], raw[dw_len:]
    _e = TIMER.start()
    check(lib.mh_mlp_wgrad(ptr(acts), ptr(dpre), acts_tile, dpre_tile, n_layers, a_p, d_p, i_p, o_p, ptr(ws), ptr(dw_raw),
                           ptr(db_raw), n_tiles, stream()), "mh_mlp_wgrad")
    TIMER.stop("mh_mlp_wgrad[" + tag + "]", _e)
    return raw          # dw_raw | db_raw, tile-row order (packing.JointPacker.unpack_grads maps it back)


WARP_ACT_ROWS, WARP_DPRE_ROWS = 64 + 2 * 640 + 40, 2 * 672     # csrc/mlp.hip: activations + 40 rows of ReLU masks
FIELD_ACT_ROWS, FIELD_DPRE_ROWS = 96 + 64 * 5 + 8, 64 * 5 + 32   # activations + 8 rows of ReLU masks


class _WarpMLP(torch.autograd.Function):
    """deform_net + topo_net on [freq(x), per-slot code bias]  (model.py:412-437).

    params = 12 tensors per net, deform first:  W0x [128,39], W1..W4 [128,128], W5 [n_out,128],
    b0 (unused here: it lives in bias0), b1..b4 [128], b5 [n_out]  -- natural, effective weights.
    bias0_{d,t} [n_slots,128] = code_slot @ W0[:,39:].T + b0, built by the caller in torch so that
    autograd carries the gradient on to the deform code, W0's code columns and b0.
    """

    @staticmethod
    def forward(ctx, x, slot, bias0_d, bias0_t, n_bands, *params):
        require_gpu(x, bias0_d, bias0_t, *params)
        lib = _lib.load()
        assert len(params) == 24
        pd, pt_ = params[:12], params[12:]
        jp = warp_joint_packer()       # both nets' fragments, transposed fragments and bias packs: 2 gathers in all
        det = lambda ts: [p.detach() for p in ts]
        fpack, bpack = jp.pack([det(pd[:6]), det(pt_[:6])], [det(pd[6:]), det(pt_[6:])])
        wd, wt, bd, bt = jp.take(fpack, jp.w[0]), jp.take(fpack, jp.w[1]), jp.take(fpack, jp.b[0]), jp.take(fpack, jp.b[1])
        wdT, wtT = jp.take(bpack, jp.wT[0]), jp.take(bpack, jp.wT[1])
        x = x.detach().contiguous().float()
        M, dev = x.shape[0], x.device
        need_grad = any(ctx.needs_input_grad)
        acts = torch.empty(lib.mh_warp_acts_floats(M), device=dev) if need_grad else None
        deform, topo = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
        b0d, b0t = bias0_d.detach().contiguous(), bias0_t.detach().contiguous()
        slot_c = None if slot is None else slot.contiguous()
        _e = TIMER.start()
        check(lib.mh_warp_fwd(ptr(x), ptr(slot_c), ptr(b0d), ptr(b0t), ptr(wd), ptr(wt), ptr(bd), ptr(bt), n_bands,
                              ptr(deform), ptr(topo), ptr(acts), M, stream()), "mh_warp_fwd")
        TIMER.stop("mh_warp_fwd", _e)
        ctx.save_for_backward(x, slot_c, wdT, wtT, acts)
        ctx.n_bands, ctx.n_slots = n_bands, bias0_d.shape[0]
        return deform, topo

    @staticmethod
    def backward(ctx, g_deform, g_topo):
        lib = _lib.load()
        x, slot, wdT, wtT, acts = ctx.saved_tensors
        M, dev = x.shape[0], x.device
        n_tiles = lib.mh_mlp_tiles(M)
        dpre = torch.empty(lib.mh_warp_dpre_floats(M), device=dev)
        g_x = torch.empty(M, 3, device=dev) if ctx.needs_input_grad[0] else None   # NULL -> the kernel skips the W0^T stage
        c = lambda t: None if t is None else t.contiguous()
        _e = TIMER.start()
        check(lib.mh_warp_bwd_data(ptr(x), ptr(c(g_deform)), ptr(c(g_topo)), ptr(wdT), ptr(wtT), ctx.n_bands, ptr(acts),
                                   ptr(dpre), ptr(g_x), M, stream()), "mh_warp_bwd_data")
        TIMER.stop("mh_warp_bwd_data", _e)
        act_off, dpre_off, in_pad, out_pad = [], [], [], []
        for net in range(2):
            for l in range(6):
                act_off.append(0 if l == 0 else (64 + net * 640 + (l - 1) * 128) * 32)
                dpre_off.append((net * 672 + l * 128) * 32)
                in_pad.append(64 if l == 0 else 128)
                out_pad.append(32 if l == 5 else 128)
        raw = _wgrad(lib, acts, dpre, WARP_ACT_ROWS * 32, WARP_DPRE_ROWS * 32, act_off, dpre_off, in_pad, out_pad, n_tiles,
                     dev, "warp")
        (gw_d, gw_t), (gb_d, gb_t) = warp_joint_packer().unpack_grads(raw)
        # per-slot first-layer bias gradient: sum of dPre0 over the points of each slot
        if ctx.n_slots == 1:
            g_b0d, g_b0t = gb_d[0][None], gb_t[0][None]
        else:
            dp = dpre.view(n_tiles, WARP_DPRE_ROWS, 32)
            per_pt_d = dp[:, 0:128, :].permute(0, 2, 1).reshape(-1, 128)[:M]
            per_pt_t = dp[:, 672:800, :].permute(0, 2, 1).reshape(-1, 128)[:M]
            idx = slot.long()
            g_b0d = torch.zeros(ctx.n_slots, 128, device=dev).index_add_(0, idx, per_pt_d)
            g_b0t = torch.zeros(ctx.n_slots, 128, device=dev).index_add_(0, idx, per_pt_t)
        zero_b0 = lambda g: torch.zeros_like(g)   # b0 itself gets its gradient through bias0
        grads_d = gw_d + [zero_b0(gb_d[0])] + gb_d[1:]
        grads_t = gw_t + [zero_b0(gb_t[0])] + gb_t[1:]
        return (g_x, None, g_b0d, g_b0t, None, *grads_d, *grads_t)


def warp_mlp(x, slot, bias0_d, bias0_t, n_bands, params_d: Sequence[torch.Tensor], params_t: Sequence[torch.Tensor]):
    return _WarpMLP.apply(x, slot, bias0_d, bias0_t, n_bands, *params_d, *params_t)


class _FieldMLP(torch.autograd.Function):
    """sdf_net (+Laplace density) and color_net on [freq(xc), hash, topo] / [hash_c, geo]
    (model.py:273-307, density.py:22-31).

    params: Ws0 [64,73], Ws1 [64,64], Ws2 [33,64], Wc0 [64,64], Wc1 [64,64], Wc2 [3,64],
            bs0, bs1, bs2, bc0, bc1, bc2   (natural, effective weights)
    beta: 0-dim device tensor = |beta_param| + 1e-4 (stays on the device: no host sync).
    """

    @staticmethod
    def forward(ctx, xc, feat_s, feat_c, topo, beta, n_bands, with_color, *params):
        require_gpu(xc, feat_s, feat_c, topo, beta, *params)
        lib = _lib.load()
        assert len(params) == 12
        jp = field_joint_packer()
        fpack, bpack = jp.pack([[p.detach() for p in params[:6]]], [[p.detach() for p in params[6:]]])
        w, b, wT = jp.take(fpack, jp.w[0]), jp.take(fpack, jp.b[0]), jp.take(bpack, jp.wT[0])
        xc = xc.detach().contiguous().float()
        M, dev = xc.shape[0], xc.device
        fs = feat_s.detach().contiguous()
        fc = None if feat_c is None else feat_c.detach().contiguous()
        tp = None if topo is None else topo.detach().contiguous()
        beta_c = beta.detach().reshape(1).contiguous().float()
        need_grad = any(ctx.needs_input_grad)
        acts = torch.empty(lib.mh_field_acts_floats(M), device=dev) if need_grad else None
        sdf, sigma = torch.empty(M, device=dev), torch.empty(M, device=dev)
        albedo = torch.empty(M, 3, device=dev) if with_color else None
        _e = TIMER.start()
        check(lib.mh_field_fwd(ptr(xc), ptr(fs), ptr(fc), ptr(tp), ptr(w), ptr(b), ptr(beta_c), n_bands,
                               int(bool(with_color)), ptr(sdf), ptr(sigma), ptr(albedo), ptr(acts), M, stream()), "mh_field_fwd")
        TIMER.stop("mh_field_fwd", _e)
        ctx.save_for_backward(xc, wT, beta_c, acts, sdf, albedo)
        ctx.cfg = (n_bands, bool(with_color), topo is not None, feat_c is not None)
        if albedo is None:
            albedo = torch.zeros(0, device=dev)
            ctx.mark_non_differentiable(albedo)
        return sdf, sigma, albedo

    @staticmethod
    def backward(ctx, g_sdf, g_sigma, g_albedo):
        lib = _lib.load()
        xc, wT, beta_c, acts, sdf, albedo = ctx.saved_tensors
        n_bands, with_color, has_topo, has_fc = ctx.cfg
        M, dev = xc.shape[0], xc.device
        n_tiles = lib.mh_mlp_tiles(M)
        dpre = torch.empty(lib.mh_field_dpre_floats(M), device=dev)
        if not with_color:
            g_albedo = None   # colour rows of the scratch are neither written nor read on this path
        g_xc = torch.empty(M, 3, device=dev) if ctx.needs_input_grad[0] else None   # NULL: the kernel skips the d/dx stage
        g_fs = torch.empty(M, 32, device=dev)
        g_fc = torch.empty(M, 32, device=dev) if (with_color and has_fc) else None
        g_tp = torch.empty(M, 2, device=dev)
        g_bp = torch.empty(n_tiles, device=dev)
        gmax = torch.zeros(2, dtype=torch.int32, device=dev)     # max |g_fs|, max |g_fc| as float bits
        c = lambda t: None if t is None else t.contiguous()
        _e = TIMER.start()
        check(lib.mh_field_bwd_data(ptr(xc), ptr(sdf), ptr(albedo if with_color else None), ptr(c(g_sdf)),
                                    ptr(c(g_sigma)), ptr(c(g_albedo)), ptr(wT), ptr(beta_c), n_bands, int(with_color),
                                    ptr(acts), ptr(dpre), ptr(g_xc), ptr(g_fs), ptr(g_fc), ptr(g_tp), ptr(g_bp), ptr(gmax),
                                    M, stream()), "mh_field_bwd_data")
        TIMER.stop("mh_field_bwd_data", _e)
        _GMAX.put(g_fs, gmax, 0)
        _GMAX.put(g_fc, gmax, 1)
        pk = field_packer()
        act_rows = [0, 96, 160, 224, 288, 352]
        dpre_rows = [0, 64, 128, 192, 256, 320]
        if with_color:
            raw = _wgrad(lib, acts, dpre, FIELD_ACT_ROWS * 32, FIELD_DPRE_ROWS * 32, [r * 32 for r in act_rows],
                         [r * 32 for r in dpre_rows], pk.wg_in, pk.wg_out, n_tiles, dev, "field")
            (gw,), (gb,) = field_joint_packer().unpack_grads(raw)
        else:   # FD-normal taps: the sdf net only; the colour net's gradients are zero
            # dP2 has one non-zero row (the sdf output, first row of its second 32-row tile): the kernel parked only that
            # tile, so layer 2's weight gradient is a 32-row launch on it; the geo rows' gradients are zero
            wg_out = [pk.wg_out[0], pk.wg_out[1], 32]
            raw = _wgrad(lib, acts, dpre, FIELD_ACT_ROWS * 32, FIELD_DPRE_ROWS * 32, [r * 32 for r in act_rows[:3]],
                         [dpre_rows[0] * 32, dpre_rows[1] * 32, (dpre_rows[2] + 32) * 32], pk.wg_in[:3], wg_out, n_tiles, dev,
                         "field")
            n01 = pk.wg_in[0] * pk.wg_out[0] + pk.wg_in[1] * pk.wg_out[1]
            n2 = pk.wg_in[2] * 32
            dw01, dw2, db = raw[:n01], raw[n01:n01 + n2], raw[n01 + n2:]
            nb01 = pk.wg_out[0] + pk.wg_out[1]
            z = raw.new_zeros
            dw_raw = torch.cat([dw01, z(n2), dw2, z(pk.raw_dw - n01 - 2 * n2)])          # [.. | L2 tile 0 = 0 | L2 tile 1 | colour = 0]
            db_raw = torch.cat([db[:nb01], z(32), db[nb01:], z(pk.raw_db - nb01 - 64)])
            gw, gb = pk.unpack_grads(dw_raw, db_raw)
        g_beta = g_bp.sum().reshape(())
        return (g_xc, g_fs, g_fc, g_tp if has_topo else None, g_beta, None, None, *gw, *gb)


def field_mlp(xc, feat_s, feat_c, topo, beta, n_bands, with_color, params: Sequence[torch.Tensor]):
    """-> sdf [M], sigma [M], albedo [M,3] (empty when with_color is False)."""
    return _FieldMLP.apply(xc, feat_s, feat_c, topo, beta, n_bands, with_color, *params)


class _WeightNormAll(torch.autograd.Function):
    """W_l = g_l * v_l / ||v_l||_row for a list of weight-normed layers (decoders.py:51-52), one launch forward and one
    backward (csrc/wnorm.hip) instead of norm / divide / multiply and their autograd graph per layer."""

    @staticmethod
    def forward(ctx, n, *vg):
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
