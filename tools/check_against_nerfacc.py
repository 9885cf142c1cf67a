"""One command from "unpinned" to "pinned": this build's occupancy marcher and compositor against nerfacc itself.

nerfacc is third-party, un-vendored and version un-pinned in the reference (docs/INSTALL.md:21) and is NOT in this image, so
`morpheus_amd.occgrid.OccupancyGrid` (interval placement, packed layout, the 'estimator' checkpoint buffers) and
`mh_composite_fwd` (nerfacc.render_weight_from_density + accumulate_along_rays) follow nerfacc's published 0.5.x behaviour as
recalled (occgrid.py:1-15, csrc/composite.hip:1-6) -- the parity tests compare them with this build's own oracle.  On a box WITH
nerfacc and an MI355X, this script feeds both the closed-form rays of `morpheus_amd.synth` through the reference's own call
sites and prints the differences:

    python tools/check_against_nerfacc.py            # exit code 0: intervals identical / within 1e-6, weights within 1e-5

  1. sampling -- morpheus.py:629-638: `OccGridEstimator(roi_aabb, resolution=128).sampling(rays_o, rays_d, sigma_fn=None,
     render_step_size=step, alpha_thre=0, stratified=False, cone_angle=0.0, early_stop_eps=0)` on a closed-form binary grid (a
     sphere of radius 0.6 and an off-centre box) against `OccupancyGrid.sampling` with the jitter pinned at 0: per-ray sample
     counts, ray indices, t_starts / t_ends.  (stratified=True draws nerfacc's own random offsets; pinning them needs the same
     generator, so the comparison is made un-jittered; the jitter is one per-ray shift of the same lattice in both.)
  2. compositing -- morpheus.py:675-685: `render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=, n_rays=)` and
     `accumulate_along_rays(weights, values, ray_indices, n_rays)` against `ops.composite` on the SAME packed samples (nerfacc's)
     with a closed-form density and colour: weights, opacity, depth, colour.
  3. the estimator's state_dict keys / shapes / dtypes against `OccupancyGrid.state_dict()` (checkpoint entry 'estimator',
     morpheus.py:341,355).

It cannot run in the build container (no nerfacc, no GPU); tests/test_host.py only checks that it imports and fails with its own
message there.
"""
from __future__ import annotations

import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def closed_form_binary(R: int, bound: float):
    """occupied cells: a sphere of radius 0.6 around the origin or a box [0.3, 0.8] x [-0.2, 0.4] x [-0.9, -0.5] (cell centres)"""
    import torch
    c = (torch.arange(R, dtype=torch.float32) + 0.5) / R * (2 * bound) - bound
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    sphere = (x * x + y * y + z * z) < 0.36
    box = (x > 0.3) & (x < 0.8) & (y > -0.2) & (y < 0.4) & (z > -0.9) & (z < -0.5)
    return sphere | box


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--res", type=int, default=64, help="the view is res x res rays of synth frame 25")
    ap.add_argument("--step", type=float, default=0.01)
    args = ap.parse_args(argv)
    try:
        import nerfacc
    except ImportError:
        print("check_against_nerfacc: nerfacc is not installed here -- run this on a box with the reference's environment "
              "(docs/INSTALL.md:21) and an MI355X", file=sys.stderr)
        return 2
    import torch
    from morpheus_amd import ops, synth
    from morpheus_amd.occgrid import OccupancyGrid
    if not torch.cuda.is_available():
        print("check_against_nerfacc: needs the GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    dev = torch.device("cuda", 0)
    bound, R = 1.01, 128
    aabb = [-bound] * 3 + [bound] * 3
    o, d, _, _ = synth.frame_rays(25, args.res, args.res)
    o, d = o[0].to(dev), d[0].to(dev)
    N = o.shape[0]
    binary = closed_form_binary(R, bound).to(dev)
    ok = True

    # ---- 1. sampling (morpheus.py:196-202, 629-638)
    est = nerfacc.OccGridEstimator(roi_aabb=aabb, resolution=R).to(dev)
    est.binaries.copy_(binary.view_as(est.binaries))
    est.occs.copy_(binary.reshape(-1).float())
    ri_n, ts_n, te_n = est.sampling(o, d, sigma_fn=None, render_step_size=args.step, alpha_thre=0, stratified=False, cone_angle=0.0,
                                    early_stop_eps=0)
    grid = OccupancyGrid(aabb, R).to(dev)
    grid.set_binary(binary)
    grid.fixed_jitter = 0.0
    ri_h, ts_h, te_h = grid.sampling(o, d, sigma_fn=None, render_step_size=args.step, alpha_thre=0, stratified=False, cone_angle=0.0,
                                     early_stop_eps=0)
    cnt_n = torch.bincount(ri_n.long(), minlength=N)
    cnt_h = torch.bincount(ri_h.long(), minlength=N)
    print(f"sampling: {N} rays, nerfacc {ri_n.numel()} samples, this build {ri_h.numel()}; rays whose count differs: "
          f"{int((cnt_n != cnt_h).sum())} (max |difference| {int((cnt_n - cnt_h).abs().max())})")
    if ri_n.numel() == ri_h.numel() and torch.equal(ri_n.long(), ri_h.long()):
        e0, e1 = float((ts_n - ts_h).abs().max()), float((te_n - te_h).abs().max())
        print(f"          same packed layout; max |t_starts difference| {e0:.3e}, max |t_ends difference| {e1:.3e}")
        ok &= e0 <= 1e-6 and e1 <= 1e-6
    else:
        # different counts: say WHERE the lattices differ -- first interval of the first differing ray
        bad = int(torch.nonzero(cnt_n != cnt_h)[0]) if bool((cnt_n != cnt_h).any()) else 0
        sn, sh = ts_n[ri_n.long() == bad][:4].tolist(), ts_h[ri_h.long() == bad][:4].tolist()
        print(f"          ray {bad}: nerfacc t_starts {sn} ..., this build {sh} ...   (interval placement differs: occgrid.py:11-13)")
        ok = False

    # ---- 2. compositing on nerfacc's own samples (morpheus.py:675-685)
    tm = 0.5 * (ts_n + te_n)
    x = o[ri_n.long()] + d[ri_n.long()] * tm[:, None]
    sig = 40.0 * torch.exp(-8.0 * (x.norm(dim=-1) - 0.55).abs())             # a shell of density around the sphere's surface
    rgb = 0.5 + 0.5 * torch.sin(3.0 * x + torch.tensor([0.0, 1.0, 2.0], device=dev))
    w_n, _, _ = nerfacc.render_weight_from_density(ts_n, te_n, sig, ray_indices=ri_n, n_rays=N)
    op_n = nerfacc.accumulate_along_rays(w_n, values=None, ray_indices=ri_n, n_rays=N)
    dp_n = nerfacc.accumulate_along_rays(w_n, values=tm[:, None], ray_indices=ri_n, n_rays=N)
    cl_n = nerfacc.accumulate_along_rays(w_n, values=rgb, ray_indices=ri_n, n_rays=N)
    start, cnt = ops.packed_info(ri_n.long(), N)
    w_h, op_h, dp_h, cl_h = ops.composite(sig.contiguous(), ts_n.contiguous(), te_n.contiguous(), rgb.contiguous(), start, cnt)
    for name, a, b in (("weights", w_h, w_n), ("opacity", op_h, op_n.reshape(-1)), ("depth", dp_h, dp_n.reshape(-1)), ("colour", cl_h, cl_n)):
        err = float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
        print(f"compositing: {name:8s} max |difference| / max |nerfacc| = {err:.3e}")
        ok &= err <= 1e-5

    # ---- 3. the checkpoint entry 'estimator' (morpheus.py:341,355)
    sd_n, sd_h = est.state_dict(), grid.state_dict()
    for k in sorted(set(sd_n) | set(sd_h)):
        a, b = sd_n.get(k), sd_h.get(k)
        same = a is not None and b is not None and tuple(a.shape) == tuple(b.shape) and a.dtype == b.dtype
        print(f"state_dict: {k:12s} nerfacc {None if a is None else (tuple(a.shape), a.dtype)}  this build "
              f"{None if b is None else (tuple(b.shape), b.dtype)}  {'ok' if same else 'DIFFERS'}")
        ok &= same
    print("PINNED: nerfacc and this build agree" if ok else "DIFFERENCES FOUND (see above)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
