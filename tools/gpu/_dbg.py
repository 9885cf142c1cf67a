import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from morpheus_amd import ops
x = torch.zeros(1, device="cuda")
ops.TIMER.reset(True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000):
        e = ops.TIMER.start(); ops.TIMER.stop("k", e)
    torch.cuda.synchronize(); print("timer pair us", (time.perf_counter() - t0) / 2000 * 1e6, len(ops.TIMER._pool))
    ops.TIMER.reset(True)
ev = torch.cuda.Event(enable_timing=True)
s = torch.cuda.current_stream()
t0 = time.perf_counter()
for _ in range(2000): ev.record()
print("record() us", (time.perf_counter() - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): ev.record(s)
print("record(stream) us", (time.perf_counter() - t0) / 2000 * 1e6)
t0 = time.perf_counter()
for _ in range(2000): torch.cuda.current_stream()
print("current_stream us", (time.perf_counter() - t0) / 2000 * 1e6)
