"""GPU-box micro-benchmark of the hash-grid kernels (forward: gathered / brick-binned; backward: naive / brick-binned with
the d/dx rows gathered or staged in LDS)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morpheus_amd import synth, ops, _lib
from morpheus_amd.ops import level_resolutions
import ctypes, numpy as np

dev = "cuda"
offs, s = synth.grid_offsets()
res = level_resolutions(16, s, 16)
emb = synth.hash_tensor((int(offs[-1]), 2), 9001, 0.1).to(dev)
lib = _lib.load()
o_p = np.ascontiguousarray(offs, np.int32); r_p = np.ascontiguousarray(res, np.int32)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def run(x, tag):
    M = x.shape[0]
    grad = torch.randn(M, 32, device=dev)
    g_emb = torch.zeros_like(emb); g_x = torch.empty(M, 3, device=dev)
    ws = torch.empty(lib.mh_grid_bin_workspace_ints(), dtype=torch.int32, device=dev)
    perm = torch.empty(M, dtype=torch.int32, device=dev); bs = torch.empty(lib.mh_grid_bin_index_ints(), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t_bin = timeit(lambda: lib.mh_grid_bin_points(x.data_ptr(), M, 1.01, ws.data_ptr(), perm.data_ptr(), bs.data_ptr(), st))
    cnt = (bs[1:4098] - bs[:4097]).cpu()
    print(f"[{tag}] M={M} bin {t_bin:.3f} ms; bricks nonempty {(cnt[:4096] > 0).sum().item()}, max {cnt[:4096].max().item()}, mean(nonempty) {cnt[:4096][cnt[:4096] > 0].float().mean().item():.0f}, oob {cnt[4096].item()}")
    knob = lib.mh_grid_stage_min_points(-1)
    for nl in (16,):
        for dx, staged in ((False, False), (True, False), (True, True)):
            lib.mh_grid_stage_min_points(0 if staged else 1 << 40)        # d/dx forms: a brick's rows staged in LDS / gathered
            t = timeit(lambda: lib.mh_grid_encode_bwd_binned(grad.data_ptr(), x.data_ptr(), emb.data_ptr(), P(o_p), P(r_p), perm.data_ptr(), bs.data_ptr(), g_emb.data_ptr(), g_x.data_ptr() if dx else None, 0, M, 16, nl, 1.01, None, st))
            print(f"   binned n_levels={nl:2d} dx={int(dx)}{' rows staged in LDS' if staged else (' rows gathered' if dx else '')}: {t:.3f} ms")
    lib.mh_grid_stage_min_points(knob)
    t = timeit(lambda: lib.mh_grid_encode_bwd(grad.data_ptr(), x.data_ptr(), emb.data_ptr(), P(o_p), P(r_p), g_emb.data_ptr(), None, M, 16, 16, 1.01, st), 2)
    print(f"   naive  n_levels=16 dx=0: {t:.3f} ms")
    out = torch.empty(M, 32, device=dev)
    t = timeit(lambda: lib.mh_grid_encode_fwd(x.data_ptr(), emb.data_ptr(), P(o_p), P(r_p), out.data_ptr(), M, 16, 16, 1.01, 1, st))
    print(f"   fwd: {t:.3f} ms")
    ref = out.clone()
    t = timeit(lambda: lib.mh_grid_encode_fwd_binned(x.data_ptr(), emb.data_ptr(), P(o_p), P(r_p), perm.data_ptr(), bs.data_ptr(), out.data_ptr(), M, 16, 16, 1.01, st))
    print(f"   fwd, brick-binned (rows staged in LDS; binning not included): {t:.3f} ms; same bits: {bool(torch.equal(out, ref))}")
    # share of the LDS-sized levels: levels 0..3 are dense tables of 32 / 55 / 85 / 125 KB (level 4 = 176 KB exceeds the
    # 160 KB of LDS); levels >= n_levels are written as zeros by idle lanes, so the launch / store cost is the same in
    # every row and the differences are the gather cost of the added levels
    for nl in (1, 4, 5, 6, 10, 16):
        t = timeit(lambda: lib.mh_grid_encode_fwd(x.data_ptr(), emb.data_ptr(), P(o_p), P(r_p), out.data_ptr(), M, 16, nl, 1.01, 1, st), 10)
        print(f"   fwd, levels 0..{nl - 1:2d} active: {t:.3f} ms")


o, d, t, rid = [v.to(dev) for v in synth.frame_rays(0, 128, 128)]
ri, ts, te, xyz, rs, rc = ops.sample_uniform(o[0], d[0], synth.ray_jitter(16384).to(dev), 128, 1.01, with_xyz=True)
run(xyz, "rays 16384x128")
run((torch.rand(2097152, 3, device=dev) * 2 - 1), "uniform random")
