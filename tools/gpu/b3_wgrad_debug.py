import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops
DEV = "cuda"
torch.manual_seed(7)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nets = []
for nout in (3, 2):
    W = [torch.randn(128, 39, device=DEV) * 0.15] + [torch.randn(128, 128, device=DEV) * 0.1 for _ in range(4)] + [torch.randn(nout, 128, device=DEV) * 0.15]
    b = [torch.randn(128, device=DEV) * 0.1 for _ in range(5)] + [torch.randn(nout, device=DEV) * 0.1]
    nets.append(W + b)
x = torch.rand(M, 3, device=DEV) * 2 - 1
b0 = [torch.randn(1, 128, device=DEV) * 0.3 for _ in range(2)]
wd_, wt_ = torch.randn(M, 3, device=DEV), torch.randn(M, 2, device=DEV)
orig = ops._wgrad
cap = {}
def both(lib, acts, dpre, *a, **kw):
    kw2 = dict(kw); kw2["b3"] = False
    r32 = orig(lib, acts, dpre, *a, **kw2)
    kw2["b3"] = True
    r3 = orig(lib, acts, dpre, *a, **kw2)
    cap["r32"], cap["r3"], cap["acts"], cap["dpre"] = r32, r3, acts, dpre
    return r3
ops._wgrad = both
ops.MLP_B3 = True
ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
d, t = ops.warp_mlp(x, None, b0[0], b0[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
((d * wd_).sum() + (t * wt_).sum()).backward()
r32, r3 = cap["r32"], cap["r3"]
print("acts finite", bool(torch.isfinite(cap["acts"].view(-1, 1384, 32)[:, :1344]).all()), "dpre finite", bool(torch.isfinite(cap["dpre"]).all()))
print("raw len", r3.numel(), "nan in b3:", int(torch.isnan(r3).sum()), "nan in f32:", int(torch.isnan(r32).sum()))
off = 0
geo = ops._WARP_WG
for l, (i, o) in enumerate(zip(geo[2], geo[3])):
    n = i * o
    a, b = r3[off:off + n], r32[off:off + n]
    print(f"layer {l}: in {i} out {o}: nan {int(torch.isnan(a).sum())}  max |b3-f32| {float((a - b).abs().nan_to_num(0).max()):.3e}  max|f32| {float(b.abs().max()):.3e}")
    off += n
