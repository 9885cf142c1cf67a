"""GPU-box micro-benchmark of the fused MLP kernels (with / without activation parking)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morpheus_amd import harness, synth, ops

dev = "cuda"
model = harness.build_model("b", dev)
M = 2097152
x = (torch.rand(M, 3, device=dev) * 2 - 1)
t = torch.full((M, 1), 0.2, device=dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def run(tag, grad):
    ops.TIMER.reset(True)
    if grad:
        for _ in range(3):
            model.zero_grad()
            sdf, sig, col, _, dfm, _ = model(x, t, shading="albedo")
            (col.sum() + sdf.sum() + dfm.sum()).backward()
    else:
        with torch.no_grad():
            for _ in range(3):
                model(x, t, shading="albedo")
    torch.cuda.synchronize()
    for k, (c, tot) in sorted(ops.TIMER.summary().items(), key=lambda kv: -kv[1][1]):
        print(f"  [{tag}] {k:28s} avg {tot / c:8.3f} ms")
    ops.TIMER.reset(False)


run("train fwd+bwd", True)
run("inference (no acts stores)", False)
