"""Forward of the warp nets with and without parking (inference / training), per arithmetic mode: how much of the training
forward is the 11.4 GB of parked activations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops
DEV = "cuda"
torch.manual_seed(7)
nets = []
for nout in (3, 2):
    W = [torch.randn(128, 39, device=DEV) * 0.15] + [torch.randn(128, 128, device=DEV) * 0.1 for _ in range(4)] + [torch.randn(nout, 128, device=DEV) * 0.15]
    b = [torch.randn(128, device=DEV) * 0.1 for _ in range(5)] + [torch.randn(nout, device=DEV) * 0.1]
    nets.append(W + b)
Mb = 16384 * 128
xb = torch.rand(Mb, 3, device=DEV) * 2 - 1
b1 = [torch.randn(1, 128, device=DEV) * 0.3 for _ in range(2)]
for mode in ("b3", "h2", "f32", "b3", "h2"):
    ops.MLP_B3, ops.MLP_H2 = mode == "b3", mode == "h2"
    ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
    opnd = ops.prepare_warp_operands(ps[0], ps[1])
    out = {}
    for park in (True, False):
        ts = []
        for it in range(5):
            xg = xb.clone().requires_grad_(park)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            with torch.set_grad_enabled(park):
                ev[0].record()
                d, t = ops.warp_mlp(xg, None, b1[0], b1[1], 6, opnd)
                ev[1].record()
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]))
            del d, t
        out[park] = min(ts)
    print(f"{mode}: forward with parking {out[True]:.3f} ms, without {out[False]:.3f} ms")
