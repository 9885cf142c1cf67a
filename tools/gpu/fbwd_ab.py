#!/usr/bin/env python
"""GPU-box A/B of the fused field backward between two builds of the library (same box, same inputs).

    python tools/gpu/fbwd_ab.py                      # driver: runs itself once per library, compares the outputs, prints the timings
    MORPHEUS_HIP_LIB=... python tools/gpu/fbwd_ab.py --one out.pt

Cases: colour + sdf pass with d/dx (cfg3's call), colour + sdf without d/dx, sdf-only pass with and without d/dx (the finite-difference
taps), M = 2 097 152 points each.  Compared: every output of mh_field_bwd_fused_b3 (g_xc, g_feat_s, g_feat_c, g_topo, raw weight / bias
/ beta gradients, gmax words) -- bit for bit where possible, else max relative difference."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(out_path, M):
    import torch
    from morpheus_amd import _lib, ops
    lib = _lib.load()
    dev = "cuda"
    torch.manual_seed(0)
    ops.set_mlp_mode("b3")
    Ws = [torch.randn(64, 73, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(33, 64, device=dev) * 0.2]
    Wc = [torch.randn(64, 64, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(3, 64, device=dev) * 0.2]
    bs = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(33, device=dev) * 0.1]
    bc = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(3, device=dev) * 0.1]
    fop = ops.prepare_field_operands([p.requires_grad_() for p in Ws + Wc + bs + bc])
    x = (torch.rand(M, 3, device=dev) * 2 - 1).contiguous()
    fs, fc = torch.randn(M, 32, device=dev) * 0.1, torch.randn(M, 32, device=dev) * 0.1
    tp = torch.randn(M, 2, device=dev) * 0.1
    beta = torch.tensor([0.1], device=dev)
    g_sdf, g_sig, g_alb = torch.randn(M, device=dev), torch.randn(M, device=dev) * 0.01, torch.randn(M, 3, device=dev)
    res, times = {}, {}
    for with_color in (True, False):
        sdf, sigma, albedo, acts = ops._field_fwd(lib, x, fs, fc if with_color else None, tp, beta, 6, with_color, fop, True)
        wT, b3 = ops._field_wT(fop, with_color)
        for need_dx in (True, False):
            tag = f"color{int(with_color)}_dx{int(need_dx)}"
            for it in range(4):
                if it == 1:
                    ops.TIMER.reset(enabled=True)
                out = ops._field_bwd(lib, x, wT, beta, acts, sdf, albedo, g_sdf, g_sig, g_alb if with_color else None, 6, with_color, True,
                                     with_color, need_dx, fop.jp, b3=b3)
            torch.cuda.synchronize()
            t = ops.TIMER.summary()
            times[tag] = t["mh_field_bwd_fused"][1] / t["mh_field_bwd_fused"][0]
            ops.TIMER.reset(False)
            names = ("g_xc", "g_fs", "g_fc", "g_tp", "raw", "gmax")
            for n, v in zip(names, out):
                if v is not None:
                    res[tag + "." + n] = v.detach().float().cpu() if v.dtype != torch.int32 else v.detach().cpu()
    torch.save(dict(res=res, times=times), out_path)
    print({k: round(v, 4) for k, v in times.items()})


def main():
    if "--one" in sys.argv:
        M = int(os.environ.get("FBWD_POINTS", str(128 * 128 * 128)))
        return one(sys.argv[sys.argv.index("--one") + 1], M)
    import torch
    libs = {"head": os.path.join(ROOT, "morpheus_amd", "_build", "libmorpheus_head.so"), "new": None}
    outs = {}
    for rep in range(2):
        for name, path in libs.items():
            env = dict(os.environ)
            if path:
                env["MORPHEUS_HIP_LIB"] = path
            else:
                env.pop("MORPHEUS_HIP_LIB", None)
            o = f"/tmp/fbwd_{name}.pt"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", o], env=env, capture_output=True, text=True, timeout=600)
            print(name, rep, (r.stdout.strip().splitlines() or ["<no output>"])[-1], r.stderr.strip()[-300:] if r.returncode else "")
            if r.returncode == 0:
                outs[name] = torch.load(o)
    if len(outs) == 2:
        a, b = outs["head"]["res"], outs["new"]["res"]
        for k in sorted(a):
            if k not in b:
                print("missing in new:", k)
                continue
            x, y = a[k], b[k]
            if x.dtype == torch.int32:
                print(f"{k:22s} equal={bool(torch.equal(x, y))} head={x.tolist()} new={y.tolist()}")
                continue
            x, y = x.double(), y.double()
            if k.endswith(".raw"):          # per segment: dW s0 s1 s2 c0 c1 c2 | db s0 .. c2 | d(beta)
                sizes = [64 * 96, 64 * 64, 64 * 64, 64 * 64, 64 * 64, 32 * 64, 64, 64, 64, 64, 64, 32, 1]
                names = ["dW_s0", "dW_s1", "dW_s2", "dW_c0", "dW_c1", "dW_c2", "db_s0", "db_s1", "db_s2", "db_c0", "db_c1", "db_c2", "dbeta"]
                o = 0
                for nm, sz in zip(names, sizes):
                    xs, ys = x[o:o + sz], y[o:o + sz]
                    o += sz
                    sc = float(xs.abs().max().clamp_min(1e-30))
                    print(f"    {k}.{nm:6s} equal={bool(torch.equal(xs, ys))} max|d|/max|x|={float((xs - ys).abs().max()) / sc:.3e} nan={int(torch.isnan(ys).sum())}")
            scale = float(x.abs().max().clamp_min(1e-30))
            d = float((x - y).abs().max())
            rl2 = float((x - y).norm() / x.norm().clamp_min(1e-30))
            print(f"{k:22s} bit-equal={bool(torch.equal(a[k], b[k]))} max|d|/max|x|={d / scale:.3e} rel-L2={rl2:.3e} nan={bool(torch.isnan(y).any())}")


if __name__ == "__main__":
    main()
