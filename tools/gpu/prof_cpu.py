"""GPU-box helper: where does the HOST time of a bench step go?  cProfile over a few steps of one workload."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
args = bench.parse_args(["--workload", wl, "--no-cpu-baseline"])
dev = torch.device("cuda", 0)
builder = {"train_real": bench.build_train_real, "density128": bench.build_density128}.get(wl, bench.build_render_workload)
w = builder(args, 0, 1, dev)
for _ in range(3):
    w["step"]()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    w["step"]()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{wl}: host issue time per step {t_issue / 10 * 1e3:.2f} ms, wall per step {t_all / 10 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    w["step"]()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
