#!/usr/bin/env python
"""Where does a warp_fwd_kernel wave spend its time?  Needs the trace build of the library:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMH_PHASE_TRACE -o morpheus_amd/_build/libmorpheus_trace.so morpheus_amd/csrc/*.hip
Wave 0 of every 64th workgroup stamps s_memtime at the phase boundaries; this prints the mean duration of each phase."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "morpheus_amd", "_build", "libmorpheus_trace.so"))
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.mh_warp_fwd.argtypes = [P] * 8 + [I32, P, P, P, I64, P]
lib.mh_warp_acts_floats.restype = I64
lib.mh_warp_acts_floats.argtypes = [I64]
M = 128 * 128 * 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(M, 3, device=dev, generator=g) * 2 - 1)
wd = torch.randn(74752, device=dev, generator=g) * 0.05
wt = torch.randn(74752, device=dev, generator=g) * 0.05
bd, bt = torch.zeros(544, device=dev), torch.zeros(544, device=dev)
b0d, b0t = torch.zeros(1, 128, device=dev), torch.zeros(1, 128, device=dev)
deform, topo = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
acts = None if os.environ.get("MH_TRACE_NOPARK") else torch.empty(lib.mh_warp_acts_floats(M), device=dev)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(int(os.environ.get('MH_TRACE_ITERS', '3'))):
    e0.record()
    rc = lib.mh_warp_fwd(x.data_ptr(), None, b0d.data_ptr(), b0t.data_ptr(), wd.data_ptr(), wt.data_ptr(), bd.data_ptr(),
                         bt.data_ptr(), 6, deform.data_ptr(), topo.data_ptr(), None if acts is None else acts.data_ptr(), M, st)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0
print("kernel ms", e0.elapsed_time(e1))
buf = (ctypes.c_longlong * (256 * 64))()
assert lib.mh_trace_read(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(256, 64).astype(np.float64)
span = t[:, 60] - t[:, 0]
print("ticks per traced workgroup: mean %.0f  (min %.0f max %.0f)" % (span.mean(), span.min(), span.max()))
real = t[:, 63] - t[:, 62]          # the same span on the constant 100 MHz counter
ok = real > 0
print("effective shader clock while the traced workgroups ran: %.0f MHz (s_memtime ticks per s_memrealtime tick x 100 MHz)"
      % (100.0 * (span[ok] / real[ok]).mean()))
# calibrate: 16384 workgroups / 512 resident => 32 rounds per kernel
tick_us = None
names = ["wait+barrier -> burst start", "MFMA burst", "barrier after burst", "DMA issue", "epilogue (ReLU + stores)"]
tot = np.zeros(5)
for net in range(2):
    for l in range(6):
        b = 1 + (net * 6 + l) * 5
        prev = t[:, b - 1] if not (net == 0 and l == 0) else t[:, 0]
        d = [t[:, b] - prev, t[:, b + 1] - t[:, b], t[:, b + 2] - t[:, b + 1], t[:, b + 3] - t[:, b + 2], t[:, b + 4] - t[:, b + 3]]
        if l == 5:
            d[3] = np.zeros(256)
            d[4] = t[:, b + 4] - t[:, b + 2]
        if 1 <= l <= 4 and net == 0 and l == 2:
            print("layer 2 of net 0 (ticks):", " | ".join(f"{n}: {v.mean():.0f}" for n, v in zip(names, d)))
        tot += np.array([v.mean() for v in d])
print("sum over the 12 layers (ticks):")
for n, v in zip(names, tot):
    print(f"   {n:32s} {v:9.0f}  {100 * v / tot.sum():5.1f} %")
print("   total", tot.sum())
