"""Where the HOST spends a real-view training step (round 6): cProfile over N eager steps, cumulative and own time of the top entries.
The eager step is host-bound (6.3 ms against 4.9 ms of kernels): every microsecond of Python per C-ABI call is step time.

    python tools/gpu/host_profile.py [--glue fused|reference] [--steps 40]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glue", default="fused")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--timers", action="store_true", help="with the per-kernel HIP-event timers on (bench.py's kernel tables)")
    a = ap.parse_args()
    import bench
    args = bench.parse_args(["--workload", "train_real", "--glue", a.glue, "--no-cpu-baseline", "--no-kernel-timers"])
    args.rays = args.rays or 2048
    wl = bench.build_train_real(args, 0, 1, torch.device("cuda", 0))
    step = wl["step"]
    if a.timers:
        from morpheus_amd import ops
        ops.TIMER.reset(True)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(a.top)
        txt = s.getvalue()
        print(f"==== sorted by {key} ({a.steps} steps) ====")
        print("\n".join(l[:170] for l in txt.splitlines()[4:]))


if __name__ == "__main__":
    main()
