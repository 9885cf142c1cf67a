#!/usr/bin/env python
"""Where does a warp_fwd_b3_kernel wave spend a hidden layer?  Needs the trace build of the library:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMH_PHASE_TRACE -o morpheus_amd/_build/libmorpheus_trace.so morpheus_amd/csrc/*.hip
Wave 0 of every 32nd workgroup stamps s_memtime at the phase boundaries of net 0's layers 1..4."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morpheus_amd import ops
lib = ctypes.CDLL(os.path.join(ROOT, "morpheus_amd", "_build", "libmorpheus_trace.so"))
P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
lib.mh_warp_fwd_b3.argtypes = [P] * 8 + [I32, P, P, P, I64, P]
lib.mh_warp_acts_floats.restype = I64
lib.mh_warp_acts_floats.argtypes = [I64]
M = 128 * 128 * 128
dev = "cuda"
torch.manual_seed(0)
ps = []
for nout in (3, 2):
    W = [torch.randn(128, 39, device=dev) * 0.15] + [torch.randn(128, 128, device=dev) * 0.1 for _ in range(4)] + [torch.randn(nout, 128, device=dev) * 0.15]
    b = [torch.randn(128, device=dev) * 0.1 for _ in range(5)] + [torch.randn(nout, device=dev) * 0.1]
    ps.append(W + b)
ops.set_mlp_mode("b3")
op = ops.prepare_warp_operands(ps[0], ps[1])
x = torch.rand(M, 3, device=dev) * 2 - 1
b0d, b0t = torch.randn(1, 128, device=dev) * 0.3, torch.randn(1, 128, device=dev) * 0.3
deform, topo = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
acts = None if os.environ.get("MH_TRACE_NOPARK") else torch.empty(lib.mh_warp_acts_floats(M), device=dev)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(int(os.environ.get("MH_TRACE_ITERS", "4"))):
    e0.record()
    rc = lib.mh_warp_fwd_b3(x.data_ptr(), None, b0d.data_ptr(), b0t.data_ptr(), op.w3[0].data_ptr(), op.w3[1].data_ptr(),
                            op.b[0].data_ptr(), op.b[1].data_ptr(), 6, deform.data_ptr(), topo.data_ptr(),
                            None if acts is None else acts.data_ptr(), M, st)
    e1.record(); torch.cuda.synchronize(); assert rc == 0
print("kernel ms", e0.elapsed_time(e1), "(stamped build)")
buf = (ctypes.c_longlong * (256 * 64))()
assert lib.mh_b3_trace_read(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(256, 64).astype(np.float64)
span = t[:, 3 * 8 + 6] - t[:, 0]; real = t[:, 63] - t[:, 62]; ok = real > 0
print("effective shader clock over the traced layers: %.0f MHz" % (100.0 * (span[ok] / real[ok]).mean()))
names = ["stage wait (vmcnt 0 + barrier)", "quarters 1-3 (144 MFMAs)", "quarter 4 + hidden half epilogue", "reach barrier stamp", "barrier", "DMA issue",
         "exposed half epilogue", "bias loads -> next stamp 0"]
tot = np.zeros(8)
for l in range(4):
    b = l * 8
    d = [t[:, b + 1] - t[:, b], t[:, b + 2] - t[:, b + 1], t[:, b + 3] - t[:, b + 2], np.zeros(256), t[:, b + 4] - t[:, b + 3], t[:, b + 5] - t[:, b + 4],
         t[:, b + 6] - t[:, b + 5], (t[:, b + 8] - t[:, b + 6]) if l < 3 else np.zeros(256)]
    tot += np.array([v.mean() for v in d])
print("mean ticks per hidden layer of one wave (net 0, layers 1..4):")
for n, v in zip(names, tot / 4):
    print(f"   {n:36s} {v:8.0f}  {100 * v / tot.sum() * 4:5.1f} %")
print("   total per layer", tot.sum() / 4)
