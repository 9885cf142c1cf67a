"""GPU-box check of the bf16x3 warp kernels against the fp32-MFMA ones on the same operands: outputs, parked tiles, timing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import synth, ops, _lib
from morpheus_amd.ops import ptr, stream, check

dev = "cuda"
lib = _lib.load()
torch.manual_seed(0)


def params(scale):
    out = []
    for nout in (3, 2):
        W = [torch.randn(128, 39, device=dev) * scale] + [torch.randn(128, 128, device=dev) * scale * 0.6 for _ in range(4)] + \
            [torch.randn(nout, 128, device=dev) * scale]
        b = [torch.randn(128, device=dev) * 0.1 for _ in range(5)] + [torch.randn(nout, device=dev) * 0.1]
        out.append(W + b)
    return out


def run(M, n_slots, with_acts, scale=0.15):
    pd, pt_ = params(scale)
    ops.set_mlp_mode("b3")
    op = ops.prepare_warp_operands(pd, pt_)
    assert op.w3 is not None
    x = (torch.rand(M, 3, device=dev) * 2 - 1) * (0.0 if scale == 0.0 else 1.0)
    slot = (torch.arange(M, device=dev) % n_slots).int()
    b0d, b0t = torch.randn(n_slots, 128, device=dev) * 0.3 * (scale != 0.0), torch.randn(n_slots, 128, device=dev) * 0.3 * (scale != 0.0)
    (wd, wt), (bd, bt) = op.w, op.b
    res = {}
    for mode in ("f32", "b3"):
        acts = torch.zeros(lib.mh_warp_acts_floats(M), device=dev) if with_acts else None
        d, t = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)

        def call():
            if mode == "b3":
                check(lib.mh_warp_fwd_b3(ptr(x), ptr(slot), ptr(b0d), ptr(b0t), ptr(op.w3[0]), ptr(op.w3[1]), ptr(bd), ptr(bt), 6,
                                         ptr(d), ptr(t), ptr(acts), M, stream()), "b3")
            else:
                check(lib.mh_warp_fwd(ptr(x), ptr(slot), ptr(b0d), ptr(b0t), ptr(wd), ptr(wt), ptr(bd), ptr(bt), 6, ptr(d), ptr(t),
                                      ptr(acts), M, stream()), "f32")
        call(); torch.cuda.synchronize()
        n = 10 if M > 500000 else 3
        t0 = time.perf_counter()
        for _ in range(n): call()
        torch.cuda.synchronize()
        res[mode] = (d.clone(), t.clone(), None if acts is None else acts.clone(), (time.perf_counter() - t0) / n * 1e3)
    # float64 reference of the network on a sample of points
    idx = torch.arange(0, M, max(1, M // 4096), device=dev)
    xs = x[idx].double()
    enc = [xs]
    for b in range(6):
        enc += [torch.sin(xs * 2 ** b), torch.cos(xs * 2 ** b)]
    e = torch.cat(enc, -1)
    errs = {}
    for k, (P, b0, nout) in enumerate(((pd, b0d, 3), (pt_, b0t, 2))):
        hcur = torch.relu(e @ P[0].double().t() + b0[slot[idx].long()].double())
        for l in range(1, 5):
            hcur = torch.relu(hcur @ P[l].double().t() + P[6 + l].double())
        o = hcur @ P[5].double().t() + P[11].double()
        for mode in ("f32", "b3"):
            errs[(k, mode)] = float((res[mode][k][idx].double() - o).abs().max()), float(o.abs().max())
    dd = float((res["f32"][0] - res["b3"][0]).abs().max()); dt_ = float((res["f32"][1] - res["b3"][1]).abs().max())
    line = f"M={M} slots={n_slots} acts={int(with_acts)}: f32 {res['f32'][3]:.3f} ms  b3 {res['b3'][3]:.3f} ms  |f32-b3| deform {dd:.2e} topo {dt_:.2e}"
    line += "  vs f64: " + " ".join(f"{'dt'[k]}/{m} {v[0]:.2e}(of {v[1]:.2f})" for (k, m), v in errs.items())
    if with_acts:
        nt = lib.mh_mlp_tiles(M)
        a0 = res["f32"][2].view(nt, -1, 32); a1 = res["b3"][2].view(nt, -1, 32)
        hid = 64 + 2 * 640
        line += f"  parked max diff {float((a0[:, :hid] - a1[:, :hid]).abs().max()):.2e}"
        m0 = a0[:, hid:].contiguous().view(torch.int32); m1 = a1[:, hid:].contiguous().view(torch.int32)
        x_ = (m0 ^ m1)
        bits = sum(int(((x_ >> k) & 1).sum()) for k in range(32))
        line += f"  mask bits differing {bits} of {m0.numel() * 32}"
    print(line, flush=True)


run(1000, 3, True)
run(1000, 1, False)
run(128 * 7 + 5, 2, True)
run(2097152, 1, True)
run(2097152, 1, False)
run(2097152, 1, True, scale=0.3)
print('--- all-zero weights and inputs (data-dependent power: is the kernel running against the power budget?)')
run(2097152, 1, True, scale=0.0)
run(2097152, 1, False, scale=0.0)
