"""What the box's HBM sustains for pure writes, pure reads and copies (torch kernels), to put parking traffic in context."""
import time, torch
dev = "cuda"
n = 2 * 1024 ** 3            # 2 Gi floats = 8 GiB
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
w = t(lambda: a.fill_(1.5)); print(f"fill   8 GiB: {w*1e3:.2f} ms  {8*1.0737/w/1e3:.2f} TB/s written")
r = t(lambda: a.sum());      print(f"sum    8 GiB: {r*1e3:.2f} ms  {8*1.0737/r/1e3:.2f} TB/s read")
c = t(lambda: b.copy_(a));   print(f"copy   8 GiB: {c*1e3:.2f} ms  {16*1.0737/c/1e3:.2f} TB/s read+written")
