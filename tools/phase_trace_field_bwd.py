#!/usr/bin/env python
"""Where does a wave of field_fused_sdf_kernel spend a tile?  Needs the trace build of the library:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DMH_PHASE_TRACE -o morpheus_amd/_build/libmorpheus_trace.so morpheus_amd/csrc/*.hip
and MORPHEUS_HIP_LIB pointing at it (tools/gpu/trace_field_bwd.sh); MORPHEUS_FIELD_BWD=f32|b3 selects the arithmetic, argv[1] = 1
for the colour + sdf pass (cfg3's call), 0 for the sdf-only pass (the finite-difference taps).  Wave 0 of every 8th workgroup stamps
s_memtime at the phase boundaries of its third tile."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morpheus_amd import _lib, ops
with_color = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
M = int(os.environ.get("MH_TRACE_POINTS", str(128 * 128 * 128)))
dev = "cuda"
torch.manual_seed(0)
Ws = [torch.randn(64, 73, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(33, 64, device=dev) * 0.2]
Wc = [torch.randn(64, 64, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(3, 64, device=dev) * 0.2]
bs = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(33, device=dev) * 0.1]
bc = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(3, device=dev) * 0.1]
params = [p.requires_grad_() for p in Ws + Wc + bs + bc]
ops.set_mlp_mode("b3")
x = (torch.rand(M, 3, device=dev) * 2 - 1).requires_grad_()
fs, fc = (torch.randn(M, 32, device=dev) * 0.1).requires_grad_(), (torch.randn(M, 32, device=dev) * 0.1).requires_grad_()
tp = (torch.randn(M, 2, device=dev) * 0.1).requires_grad_()
beta = torch.tensor(0.1, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    opnd = ops.prepare_field_operands(params)
    sdf, sig, alb = ops.field_mlp(x, fs, fc if with_color else None, tp, beta, 6, with_color, opnd)
    loss = (sdf ** 2).sum() + sig.mean() + ((alb ** 2).sum() if with_color else 0.0)
    ops.TIMER.reset(enabled=True)
    loss.backward()
    torch.cuda.synchronize()
    print({k: round(v[1] / max(v[0], 1), 4) for k, v in ops.TIMER.summary().items() if "field" in k})
lib = _lib.load()
lib.mh_fused_trace_read.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_longlong * (64 * 32))()
assert lib.mh_fused_trace_read(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(64, 32).astype(np.float64)
ok = (t[:, 31] - t[:, 30]) > 0
t = t[ok]
span, real = t[:, 14] - t[:, 0], t[:, 31] - t[:, 30]
print("arithmetic", ops.FIELD_BWD, "with_color", with_color, "traced waves", int(ok.sum()),
      "effective shader clock %.0f MHz" % (100.0 * (span / real).mean()))
names = ["prologue (gradient loads, d2)", "s2 scratch put + backward-data", "s2 wait for rows", "s2 slice rows", "s2 dW",
         "s1 put + backward-data", "s1 wait", "s1 slice", "s1 dW", "s0 put + backward-data", "s0 wait", "s0 slice", "s0 dW", "epilogue (enc deriv, stores)"]
d = np.diff(t[:, :15], axis=1).mean(0)
for n, v in zip(names, d):
    print(f"   {n:36s} {v:8.0f}  {100 * v / d.sum():5.1f} %")
print("   total ticks per tile", round(d.sum()))
