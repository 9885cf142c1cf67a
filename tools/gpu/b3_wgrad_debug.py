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
ops.set_mlp_mode("b3")
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
print("---- error pattern of layer 1 (raw [out 128][in 128]) by 32 x 32 block, and of its bias gradient")
off = geo[2][0] * geo[3][0]
a = r3[off:off + 16384].view(128, 128); b = r32[off:off + 16384].view(128, 128)
for ot in range(4):
    print("out tile", ot, [f"{float((a[32*ot:32*ot+32, 32*it:32*it+32] - b[32*ot:32*ot+32, 32*it:32*it+32]).abs().max()):.2e}" for it in range(4)])
dw_total = sum(i * o for i, o in zip(geo[2], geo[3]))
db3, db32 = r3[dw_total:], r32[dw_total:]
print("db layer0..1 max err", float((db3[:256] - db32[:256]).abs().max()), "max", float(db32[:256].abs().max()))
# ratio test: is b3 a multiple of f32 somewhere?
ratio = (a / b)[b.abs() > 1.0]
print("ratio b3/f32 quantiles", [float(ratio.quantile(q)) for q in (0.01, 0.25, 0.5, 0.75, 0.99)])
