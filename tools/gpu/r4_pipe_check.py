"""GPU-box A/B of the pipelined b3 warp kernels (MH_B3_PIPE=1, round 4) against the single-stage ones (MH_B3_PIPE=0) on the same
operands: outputs, parked tiles and mask words BIT FOR BIT (every accumulator sees the same sequence of slice products), timing at
the benchmark size, at a ragged mid size and at the small-batch shape; forward and backward-data."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops, _lib
from morpheus_amd.ops import ptr, stream, check

dev = "cuda"
lib = _lib.load()
torch.manual_seed(0)
WHAT = sys.argv[1] if len(sys.argv) > 1 else "fwd,bwd"


def params(scale=0.15):
    out = []
    for nout in (3, 2):
        W = [torch.randn(128, 39, device=dev) * scale] + [torch.randn(128, 128, device=dev) * scale * 0.6 for _ in range(4)] + \
            [torch.randn(nout, 128, device=dev) * scale]
        b = [torch.randn(128, device=dev) * 0.1 for _ in range(5)] + [torch.randn(nout, device=dev) * 0.1]
        out.append(W + b)
    return out


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def run(M, n_slots, n_bands=6, want_gx=True, scale=0.15):
    pd, pt_ = params(scale)
    ops.set_mlp_mode("b3")
    op = ops.prepare_warp_operands(pd, pt_)
    x = (torch.rand(M, 3, device=dev) * 2 - 1) * (0.0 if scale == 0.0 else 1.0)
    slot = (torch.arange(M, device=dev) % n_slots).int() if n_slots > 1 else None
    z = 0.0 if scale == 0.0 else 0.3
    b0d, b0t = torch.randn(n_slots, 128, device=dev) * z, torch.randn(n_slots, 128, device=dev) * z
    if scale == 0.0:
        pd, pt_ = [p * 0 for p in pd], [p * 0 for p in pt_]
        op = ops.prepare_warp_operands(pd, pt_)
    (bd, bt) = op.b
    gd, gt = torch.randn(M, 3, device=dev), torch.randn(M, 2, device=dev)
    res = {}
    n = 10 if M > 500000 else 20
    for pipe in ("0", "1"):
        os.environ["MH_B3_PIPE"] = pipe
        acts = torch.zeros(lib.mh_warp_acts_floats(M), device=dev)
        d, t = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
        fwd = lambda: check(lib.mh_warp_fwd_b3(ptr(x), ptr(slot), ptr(b0d), ptr(b0t), ptr(op.w3[0]), ptr(op.w3[1]), ptr(bd), ptr(bt),
                                               n_bands, ptr(d), ptr(t), ptr(acts), M, stream()), "fwd")
        ms_f = timed(fwd, n)
        fwd_ng = lambda: check(lib.mh_warp_fwd_b3(ptr(x), ptr(slot), ptr(b0d), ptr(b0t), ptr(op.w3[0]), ptr(op.w3[1]), ptr(bd), ptr(bt),
                                                  n_bands, ptr(d), ptr(t), None, M, stream()), "fwd no parking")
        ms_f0 = timed(fwd_ng, n)
        fwd(); torch.cuda.synchronize()
        out = dict(d=d.clone(), t=t.clone(), acts=acts.clone(), ms_fwd=ms_f, ms_fwd_noparking=ms_f0)
        if "bwd" in WHAT:
            dpre = torch.zeros(lib.mh_warp_dpre_floats(M), device=dev)
            gx = torch.empty(M, 3, device=dev) if want_gx else None
            bwd = lambda: check(lib.mh_warp_bwd_data_b3(ptr(x), ptr(gd), ptr(gt), ptr(op.wT3[0]), ptr(op.wT3[1]), n_bands, ptr(acts),
                                                        ptr(dpre), ptr(gx), M, stream()), "bwd")
            out["ms_bwd"] = timed(bwd, n)
            out.update(dpre=dpre.clone(), gx=None if gx is None else gx.clone())
        res[pipe] = out
    a, b = res["0"], res["1"]
    bits = lambda v: v.view(torch.int32)                 # the parked tiles carry ReLU mask WORDS: compare bit patterns, not floats
    eq = {k: (a[k] is None and b[k] is None) or bool(torch.equal(bits(a[k]), bits(b[k]))) for k in ("d", "t", "acts", "dpre", "gx") if k in a}
    worst = {k: (0 if eq[k] or a[k] is None else int((bits(a[k]) != bits(b[k])).sum())) for k in eq}
    print(f"M={M} slots={n_slots} bands={n_bands} gx={want_gx} scale={scale}: bit-identical {eq} differing words {worst}")
    print("   ms  old -> pipelined: " + ", ".join(f"{k[3:]} {a[k]:.3f} -> {b[k]:.3f}" for k in a if k.startswith("ms_")))
    assert all(eq.values()), eq


run(16384 * 128, 1)
run(16384 * 128, 1, scale=0.0)            # all-zero operands: the same instruction stream at low matrix-pipe power
run(140000 + 37, 1)
run(21000, 3, n_bands=4)
run(2048, 2048, want_gx=False)
run(96, 1)
print("OK")
