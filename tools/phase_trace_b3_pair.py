#!/usr/bin/env python
"""The two waves of ONE SIMD (waves 0 and 4 of a workgroup) through a hidden layer of warp_fwd_b3_kernel, on a common clock: who
issues MFMAs when, and when does neither?  Needs the trace build (tools/gpu/trace_b3.sh builds it).  Rows: every 64th workgroup,
even row = wave 0, odd row = wave 4; stamps per layer (the pipelined kernel of round 5): 0 layer start, 1 Q1 (+ previous layer's tiles 2,3
epilogue) done, 2 Q2 done, 3 past M (barrier + DMA issue), 4 Q3 done, 5 Q4 (+ tiles 0,1 epilogue) done, 6 past E (barrier + DMA issue).
(profiles/r05_phase_trace_warp_fwd_pair_before.txt is the round-4 kernel under its own stamp set: 0 before the stage wait, 1 accumulators
ready, 2 quarters 1-3 done, 3 quarter 4 done, 4 past the mid barrier, 5 DMA issued, 6 tiles 2,3 epilogue done.)"""
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
for it in range(4):
    e0.record()
    rc = lib.mh_warp_fwd_b3(x.data_ptr(), None, b0d.data_ptr(), b0t.data_ptr(), op.w3[0].data_ptr(), op.w3[1].data_ptr(),
                            op.b[0].data_ptr(), op.b[1].data_ptr(), 6, deform.data_ptr(), topo.data_ptr(),
                            None if acts is None else acts.data_ptr(), M, st)
    e1.record(); torch.cuda.synchronize(); assert rc == 0
print("kernel ms %.3f (stamped build)%s" % (e0.elapsed_time(e1), " WITHOUT parking" if acts is None else ""))
buf = (ctypes.c_longlong * (256 * 64))()
assert lib.mh_b3_trace_read(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(128, 2, 64).astype(np.float64)
ok = (t[:, 0, 63] - t[:, 0, 62]) > 0
t = t[ok]
span = t[:, 0, 3 * 8 + 6] - t[:, 0, 0]; real = t[:, 0, 63] - t[:, 0, 62]
print("effective shader clock over the traced layers: %.0f MHz; %d workgroups traced" % (100.0 * (span / real).mean(), ok.sum()))
names = ["0 layer start", "1 Q1 + E23(prev) done", "2 Q2 done", "3 past M + DMA issue", "4 Q3 done", "5 Q4 + E01 done", "6 past E + DMA issue"]
print("stamps relative to wave 0's stamp 0 of the same layer, mean over layers 2..3 (net 0) and workgroups [ticks]:")
print("   %-28s %10s %10s" % ("", "wave 0", "wave 4"))
rel = np.zeros((7, 2))
for l in (1, 2):
    base = t[:, 0, l * 8][:, None]
    for k in range(7):
        rel[k] += (t[:, :, l * 8 + k] - base).mean(axis=0) / 2
for k in range(7):
    print("   %-28s %10.0f %10.0f" % (names[k], rel[k, 0], rel[k, 1]))
nxt = ((t[:, :, 2 * 8] - t[:, 0, 1 * 8][:, None]).mean(axis=0) + (t[:, :, 3 * 8] - t[:, 0, 2 * 8][:, None]).mean(axis=0)) / 2
print("   %-28s %10.0f %10.0f" % ("next layer's stamp 0", nxt[0], nxt[1]))
print("per-wave phase durations [ticks] (mean over layers 1..4):")
dn = ["Q1+E23", "-", "Q2", "M", "Q3", "Q4+E01", "E", "to next stamp 0"]
for w in (0, 1):
    d = np.zeros(8)
    for l in range(4):
        b = l * 8
        seg = [t[:, w, b + 1] - t[:, w, b], None, t[:, w, b + 2] - t[:, w, b + 1], t[:, w, b + 3] - t[:, w, b + 2], t[:, w, b + 4] - t[:, w, b + 3],
               t[:, w, b + 5] - t[:, w, b + 4], t[:, w, b + 6] - t[:, w, b + 5], (t[:, w, b + 8] - t[:, w, b + 6]) if l < 3 else None]
        for i, v in enumerate(seg):
            if v is not None:
                d[i] += v.mean() / (4 if i != 7 else 3)
    print("   wave %d: " % (4 * w) + "  ".join(f"{n} {v:.0f}" for n, v in zip(dn, d) if v) + f"   total {d.sum():.0f}")
# spread of the workgroup at the two points: how far apart do waves 0 and 4 arrive
for nm, k in (("M", 2), ("E", 5)):
    arr = t[:, :, [l * 8 + k for l in range(4)]]
    print("%s: |arrival wave 0 - wave 4| mean %.0f ticks" % (nm, np.abs(arr[:, 0] - arr[:, 1]).mean()))
