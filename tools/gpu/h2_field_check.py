"""GPU check of mh_field_fwd_h2: outputs against float64 for the three forward forms, and timings at 2 M points."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops, _lib
from morpheus_amd._lib import ptr, stream
DEV = "cuda"
torch.manual_seed(5)
lib = _lib.load()
Ws = [torch.randn(64, 73, device=DEV) * 0.15, torch.randn(64, 64, device=DEV) * 0.15, torch.randn(33, 64, device=DEV) * 0.15]
Wc = [torch.randn(64, 64, device=DEV) * 0.15, torch.randn(64, 64, device=DEV) * 0.15, torch.randn(3, 64, device=DEV) * 0.15]
bs = [torch.randn(64, device=DEV) * 0.1, torch.randn(64, device=DEV) * 0.1, torch.randn(33, device=DEV) * 0.1]
bc = [torch.randn(64, device=DEV) * 0.1, torch.randn(64, device=DEV) * 0.1, torch.randn(3, device=DEV) * 0.1]
beta = torch.tensor([0.05], device=DEV)


def run(mode, M, x, fs, fc, tp, with_color=True, reps=1):
    ops.set_mlp_mode(mode)
    opnd = ops.prepare_field_operands(Ws + Wc + bs + bc)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sdf, sigma, albedo, acts = ops._field_fwd(lib, x, fs, fc, tp, beta, 6, with_color, opnd, True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sdf, albedo, best


M = 5000
x = torch.rand(M, 3, device=DEV) * 2 - 1
fs, fc, tp = torch.randn(M, 32, device=DEV) * 0.1, torch.randn(M, 32, device=DEV) * 0.1, torch.randn(M, 2, device=DEV) * 0.3
# float64 reference (model.py:273-307: sdf net on [enc(x) | hash features | topo], colour net on [hash features | geo])
xd = x.double()
enc = torch.cat([xd] + [f(xd * 2 ** k) for k in range(6) for f in (torch.sin, torch.cos)], -1)
hcur = torch.cat([enc, fs.double(), tp.double()], -1)
hcur = torch.relu(hcur @ Ws[0].double().t() + bs[0].double())
hcur = torch.relu(hcur @ Ws[1].double().t() + bs[1].double())
o = hcur @ Ws[2].double().t() + bs[2].double()
sdf64, geo = o[:, 0], o[:, 1:]
c = torch.cat([fc.double(), geo], -1)
c = torch.relu(c @ Wc[0].double().t() + bc[0].double())
c = torch.relu(c @ Wc[1].double().t() + bc[1].double())
alb64 = torch.sigmoid(c @ Wc[2].double().t() + bc[2].double())
for mode in ("f32", "b3", "h2"):
    sdf, alb, _ = run(mode, M, x, fs, fc, tp)
    print(f"{mode}: sdf max err / max |sdf| {float((sdf.double() - sdf64).abs().max() / sdf64.abs().max()):.2e}, albedo max err {float((alb.double() - alb64).abs().max()):.2e}")
    sdf2, _, _ = run(mode, M, x, fs, fc, tp, with_color=False)
    print(f"     sdf-only pass equals the full pass: {bool(torch.equal(sdf, sdf2))}")
Mb = 16384 * 128
x = torch.rand(Mb, 3, device=DEV) * 2 - 1
fs, fc, tp = torch.randn(Mb, 32, device=DEV) * 0.1, torch.randn(Mb, 32, device=DEV) * 0.1, torch.randn(Mb, 2, device=DEV) * 0.3
for mode in ("f32", "b3", "h2", "b3", "h2"):
    _, _, t1 = run(mode, Mb, x, fs, fc, tp, True, 4)
    _, _, t0 = run(mode, Mb, x, fs, fc, tp, False, 4)
    print(f"{mode}: field forward {t1:.3f} ms with colour, {t0:.3f} ms sdf only, at {Mb} points")
