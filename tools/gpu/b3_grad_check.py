import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops
DEV = "cuda"
torch.manual_seed(7)
M = 6000
nets = []
for nout in (3, 2):
    W = [torch.randn(128, 39, device=DEV) * 0.15] + [torch.randn(128, 128, device=DEV) * 0.1 for _ in range(4)] + [torch.randn(nout, 128, device=DEV) * 0.15]
    b = [torch.randn(128, device=DEV) * 0.1 for _ in range(5)] + [torch.randn(nout, device=DEV) * 0.1]
    nets.append(W + b)
x = torch.rand(M, 3, device=DEV) * 2 - 1
slot = (torch.arange(M, device=DEV) % 3).int()
b0 = [torch.randn(3, 128, device=DEV) * 0.3 for _ in range(2)]
wd_, wt_ = torch.randn(M, 3, device=DEV), torch.randn(M, 2, device=DEV)
def run(b3):
    ops.set_mlp_mode("b3" if b3 else "f32")
    ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
    xg = x.clone().requires_grad_(True)
    bb = [t.clone().requires_grad_(True) for t in b0]
    d, t = ops.warp_mlp(xg, slot, bb[0], bb[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
    ((d * wd_).sum() + (t * wt_).sum()).backward()
    return d.detach(), t.detach(), xg.grad, [p.grad for net in ps for p in net]
ps64 = [[p.double().clone().requires_grad_(True) for p in net] for net in nets]
x64 = x.double().clone().requires_grad_(True)
enc = [x64] + [f(x64 * 2 ** k) for k in range(6) for f in (torch.sin, torch.cos)]
e = torch.cat(enc, -1)
outs = []
for k, P in enumerate(ps64):
    h = torch.relu(e @ P[0].t() + b0[k].double()[slot.long()])
    for l in range(1, 5):
        h = torch.relu(h @ P[l].t() + P[6 + l])
    outs.append(h @ P[5].t() + P[11])
((outs[0] * wd_.double()).sum() + (outs[1] * wt_.double()).sum()).backward()
r32, r3 = run(False), run(True)
for name, r in (("f32", r32), ("b3", r3)):
    err = (r[2].double() - x64.grad).abs().max(dim=1).values
    print(name, "d/dx abs err: max %.3e  median %.3e  p99 %.3e  count>1e-3: %d  count>1e-4: %d  (|grad| max %.2f)" % (
        err.max(), err.median(), err.quantile(0.99), int((err > 1e-3).sum()), int((err > 1e-4).sum()), float(x64.grad.abs().max())))
    bad = torch.nonzero(err > 1e-4).flatten()[:8].tolist()
    print("   worst points", bad, [f"{float(err[i]):.2e}" for i in bad])
g64 = [p.grad for net in ps64 for p in net]
for name, r in (("f32", r32), ("b3", r3)):
    print(name, "param grad rel-L2:", " ".join("-" if b is None else f"{float((a.double()-b).norm()/b.norm()):.1e}" for a, b in zip(r[3], g64)))
