"""GPU check of the fp16x2 warp kernels (csrc/mlp_h2.hip): slice planes against the weights, values / d/dx / parameter
gradients of f32, b3 and h2 against float64, and forward / backward timings at the benchmark's 2 M points."""
import os, sys, numpy as np, torch
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


def set_mode(m):
    ops.set_mlp_mode(m)


# ---- 1. slices -----------------------------------------------------------------------------------------------------
set_mode("h2")
opnd = ops.prepare_warp_operands([p for p in nets[0]], [p for p in nets[1]])
jp = opnd.jp
flat = jp.flat([[p for p in nets[0][:6]], [p for p in nets[1][:6]]], [[p for p in nets[0][6:]], [p for p in nets[1][6:]]])
for key, blocks, table, views in (("fwd3", jp.h2_blocks, jp.h2_table, opnd.w3), ("bwd3", jp.h2T_blocks, jp.h2T_table, opnd.wT3)):
    src = flat[jp.on(flat.device)[key]].cpu().numpy()
    whole = torch.cat(views).cpu().numpy()
    words = whole.view(np.uint32)
    worst = 0.0
    for (so, n, d4, ly) in blocks:
        amax_bits = int(words[table[ly]])
        k = min(141 - (amax_bits >> 23), 60)
        ref = src[so:so + n].astype(np.float64)
        halves = whole[4 * d4:4 * d4 + 2 * n // 2].view(np.float16).astype(np.float64)   # 2 planes x n fp16
        hpl, lpl = halves[:n], halves[n:2 * n]
        rec = (hpl + lpl) / 2.0 ** k
        err = np.abs(rec - ref).max() / max(np.abs(ref).max(), 1e-30)
        worst = max(worst, err)
        layer_amax = np.float32(np.uint32(amax_bits).view(np.float32))
    print(f"{key}: slices reconstruct the weights to {worst:.2e} of the layer maximum (want <= 2.4e-7); last table entry {layer_amax:.4f}")

# ---- 2. accuracy ---------------------------------------------------------------------------------------------------
def run(mode):
    set_mode(mode)
    ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
    xg = x.clone().requires_grad_(True)
    bb = [t.clone().requires_grad_(True) for t in b0]
    d, t = ops.warp_mlp(xg, slot, bb[0], bb[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
    ((d * wd_s).sum() + (t * wt_s).sum()).backward()
    return d.detach(), t.detach(), xg.grad, [p.grad for net in ps for p in net]


ps64 = [[p.double().clone().requires_grad_(True) for p in net] for net in nets]
x64 = x.double().clone().requires_grad_(True)
enc = [x64] + [f(x64 * 2 ** k) for k in range(6) for f in (torch.sin, torch.cos)]
e = torch.cat(enc, -1)
outs, safe = [], torch.ones(M, dtype=torch.bool, device=DEV)
for k, P in enumerate(ps64):
    z = e @ P[0].t() + b0[k].double()[slot.long()]
    safe &= (z.detach().abs() > 1e-4).all(dim=1)
    h = torch.relu(z)
    for l in range(1, 5):
        z = h @ P[l].t() + P[6 + l]
        safe &= (z.detach().abs() > 1e-4).all(dim=1)
        h = torch.relu(z)
    outs.append(h @ P[5].t() + P[11])
wd_s, wt_s = wd_ * safe[:, None], wt_ * safe[:, None]
((outs[0] * wd_s.double()).sum() + (outs[1] * wt_s.double()).sum()).backward()
g64 = [p.grad for net in ps64 for p in net]
res = {m: run(m) for m in ("f32", "b3", "h2")}
for name, r in res.items():
    fe = [float((r[k].double() - outs[k]).abs().max() / outs[k].abs().max()) for k in (0, 1)]
    ge = float((r[2].double() - x64.grad).norm() / x64.grad.norm())
    print(f"{name}: fwd max err / max |out| deform {fe[0]:.2e} topo {fe[1]:.2e};  d/dx rel-L2 {ge:.2e};  nan {bool(torch.isnan(r[0]).any())}")
    print("    param grad rel-L2:", " ".join("-" if b is None else f"{float((a.double() - b).norm() / b.norm()):.1e}" for a, b in zip(r[3], g64)))

# ---- 2b. large batch: the weight-gradient kernels with per-tensor scales, against float64 ---------------------------------
if os.environ.get("H2_BIG", "1") == "1":
    Mq = 600_000
    xq = torch.rand(Mq, 3, device=DEV) * 2 - 1
    bq = [t[:1].contiguous() for t in b0]
    gq = (torch.randn(Mq, 3, device=DEV) * torch.rand(Mq, 1, device=DEV) ** 8, torch.randn(Mq, 2, device=DEV) * torch.rand(Mq, 1, device=DEV) ** 8)
    p64 = [[p.double().clone().requires_grad_(True) for p in net] for net in nets]
    e64 = torch.cat([xq.double()] + [f(xq.double() * 2 ** k) for k in range(6) for f in (torch.sin, torch.cos)], -1)
    o64 = []
    for k, P in enumerate(p64):
        hh = torch.relu(e64 @ P[0].t() + bq[k].double())
        for l in range(1, 5):
            hh = torch.relu(hh @ P[l].t() + P[6 + l])
        o64.append(hh @ P[5].t() + P[11])
    ((o64[0] * gq[0].double()).sum() + (o64[1] * gq[1].double()).sum()).backward()
    gq64 = [p.grad for net in p64 for p in net]
    del e64, o64, hh
    for mode, wg in (("f32", False), ("b3", False), ("h2", False), ("h2", True)):
        set_mode(mode)
        ops.WGRAD_H2 = wg
        ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
        d, t = ops.warp_mlp(xq, None, bq[0], bq[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
        torch.autograd.backward([d, t], [gq[0], gq[1]])
        gs = [p.grad for net in ps for p in net]
        if mode == "h2" and not wg:
            g_ref = gs
        if mode == "h2" and wg:
            print("   h2 weight-gradient kernel against the bf16x3 one on the same parked tensors, rel-L2:",
                  " ".join("-" if b is None else f"{float((a - b).norm() / b.norm()):.1e}" for a, b in zip(gs, g_ref)))
        print(f"{mode}{'+h2 wgrad' if wg else ''}: 600k points, heavy-tailed loss gradients; param grad rel-L2 vs float64:",
              " ".join("-" if b is None else f"{float((a.double() - b).norm() / b.norm()):.1e}" for a, b in zip(gs, gq64)))
    ops.WGRAD_H2 = True

# ---- 3. timing -----------------------------------------------------------------------------------------------------
if os.environ.get("H2_TIME", "1") == "1":
    Mb = 16384 * 128
    xb = torch.rand(Mb, 3, device=DEV) * 2 - 1
    gb = (torch.randn(Mb, 3, device=DEV) * 1e-6, torch.randn(Mb, 2, device=DEV) * 1e-6)
    b1 = [t[:1].contiguous() for t in b0]
    for mode in ("b3", "h2", "b3", "h2"):
        set_mode(mode)
        ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
        opnd = ops.prepare_warp_operands(ps[0], ps[1])
        tf, tb = [], []
        for it in range(4):
            xg = xb.clone().requires_grad_(True)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            d, t = ops.warp_mlp(xg, None, b1[0], b1[1], 6, opnd)
            ev[1].record()
            torch.autograd.backward([d, t], [gb[0], gb[1]])
            ev[2].record()
            torch.cuda.synchronize()
            tf.append(ev[0].elapsed_time(ev[1])); tb.append(ev[1].elapsed_time(ev[2]))
        print(f"{mode}: forward {min(tf):.3f} ms, backward (data + weight gradients) {min(tb):.3f} ms at {Mb} points")
