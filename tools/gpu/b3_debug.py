import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops, _lib
from morpheus_amd.packing import warp_joint_packer
dev = "cuda"; lib = _lib.load(); torch.manual_seed(0)
ps = []
for nout in (3, 2):
    W = [torch.randn(128, 39, device=dev) * 0.15] + [torch.randn(128, 128, device=dev) * 0.1 for _ in range(4)] + [torch.randn(nout, 128, device=dev) * 0.15]
    b = [torch.randn(128, device=dev) * 0.1 for _ in range(5)] + [torch.randn(nout, device=dev) * 0.1]
    ps.append(W + b)
ops.set_mlp_mode("b3")
op = ops.prepare_warp_operands(ps[0], ps[1])
jp = warp_joint_packer()
flat = jp.flat([p[:6] for p in ps], [p[6:] for p in ps])
src = flat[jp.on(flat.device)["fwd3"]]
w3 = torch.cat(op.w3).view(torch.int16).view(-1, 8)      # f4 units x 8 bf16
def trunc(x): return (x.view(torch.int32) & -65536).view(torch.float32)
bad = 0
for (so, n, do) in jp.b3_layers:
    s = src[so:so + n]
    hi = trunc(s); r = s - hi; mid = trunc(r); lo = r - mid
    for k, pl in enumerate((hi, mid, lo)):
        exp = (pl.view(torch.int32) >> 16).to(torch.int16).view(-1, 8)
        got = w3[do + k * (n // 8): do + (k + 1) * (n // 8)]
        nb = int((exp != got).sum()); bad += nb
        if nb: print("layer at", so, "plane", k, "mismatches", nb)
print("slice mismatches total", bad)
from morpheus_amd.ops import ptr, stream, check
M = 256
x = torch.rand(M, 3, device=dev) * 2 - 1
b0d, b0t = torch.randn(1, 128, device=dev) * 0.3, torch.randn(1, 128, device=dev) * 0.3
(wd, wt), (bd, bt) = op.w, op.b
outs = {}
for mode in ("f32", "b3"):
    acts = torch.zeros(lib.mh_warp_acts_floats(M), device=dev)
    d, t = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
    if mode == "b3":
        check(lib.mh_warp_fwd_b3(ptr(x), None, ptr(b0d), ptr(b0t), ptr(op.w3[0]), ptr(op.w3[1]), ptr(bd), ptr(bt), 6, ptr(d), ptr(t), ptr(acts), M, stream()), "b3")
    else:
        check(lib.mh_warp_fwd(ptr(x), None, ptr(b0d), ptr(b0t), ptr(wd), ptr(wt), ptr(bd), ptr(bt), 6, ptr(d), ptr(t), ptr(acts), M, stream()), "f32")
    torch.cuda.synchronize()
    outs[mode] = (d, t, acts.view(lib.mh_mlp_tiles(M), -1, 32))
a0, a1 = outs["f32"][2], outs["b3"][2]
print("H0 diff", float((a0[:, :64] - a1[:, :64]).abs().max()))
for net in range(2):
    for l in range(5):
        r0 = 64 + net * 640 + l * 128
        df = (a0[:, r0:r0 + 128] - a1[:, r0:r0 + 128]).abs()
        print(f"net {net} H{l+1}: max diff {float(df.max()):.3e} (max val {float(a0[:, r0:r0+128].abs().max()):.2f}); per out-tile:",
              [f"{float(df[:, 32*t:32*t+32].max()):.1e}" for t in range(4)], "per wave-tile:", [f"{float(df[k].max()):.1e}" for k in range(a0.shape[0])])
print("out diff", float((outs['f32'][0] - outs['b3'][0]).abs().max()), float((outs['f32'][1] - outs['b3'][1]).abs().max()))
print("---- repeat runs, M=1024")
M = 1024
x = torch.rand(M, 3, device=dev) * 2 - 1
def go(mode):
    acts = torch.zeros(lib.mh_warp_acts_floats(M), device=dev)
    d, t = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
    if mode == "b3":
        check(lib.mh_warp_fwd_b3(ptr(x), None, ptr(b0d), ptr(b0t), ptr(op.w3[0]), ptr(op.w3[1]), ptr(bd), ptr(bt), 6, ptr(d), ptr(t), ptr(acts), M, stream()), "b3")
    else:
        check(lib.mh_warp_fwd(ptr(x), None, ptr(b0d), ptr(b0t), ptr(wd), ptr(wt), ptr(bd), ptr(bt), 6, ptr(d), ptr(t), ptr(acts), M, stream()), "f32")
    torch.cuda.synchronize()
    return acts.view(lib.mh_mlp_tiles(M), -1, 32)
ref = go("f32")
for rep in range(3):
    a = go("b3")
    df = (ref[:, 64:192] - a[:, 64:192]).abs()
    print("rep", rep, "H1 tile0 wrong in wave-tiles:", [k for k in range(a.shape[0]) if float(df[k, :32].max()) > 1e-3], " rows wrong within tile 0 of wave-tile 0:",
          [r for r in range(32) if float(df[0, r].max()) > 1e-3], "pts wrong:", [p for p in range(32) if float(df[0, :32, p].max()) > 1e-3])
