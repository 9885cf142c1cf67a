"""In-process A/B of a switch on the real-view training step (round 6): ONE process, one model, the switch flipped between blocks of
steps, so that host, clocks and allocator state are shared -- separate bench.py runs of this host-bound step differ by +-0.5 ms
between (and within) boxes.  Prints per block the median and the minimum step time.

    python tools/gpu/glue_ab.py --glue reference --switch implicit      (model step cache on / off)
"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--glue", default="reference", choices=["fused", "reference", "reference_scoped"])
    ap.add_argument("--switch", default="implicit", choices=["implicit", "none"])
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--rays", type=int, default=2048)
    args = ap.parse_args()
    import bench
    from morpheus_amd import model as mm
    bargs = argparse.Namespace(rays=args.rays, glue=args.glue, graph=False, no_overlap=True, steps=args.steps)
    dev = torch.device("cuda", 0)
    wl = bench.build_train_real(bargs, 0, 1, dev)
    step = wl["step"]
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    res = {}
    for b in range(args.blocks):
        on = (b % 2 == 0)
        if args.switch == "implicit":
            mm.IMPLICIT_OPERANDS = on
        times = []
        for _ in range(args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        res.setdefault(on, []).append((statistics.median(times), min(times)))
        print(f"block {b} {args.switch}={'on' if on else 'off'}: median {statistics.median(times):.3f} ms  min {min(times):.3f}  max {max(times):.3f}", flush=True)
    for on, v in res.items():
        print(f"{args.glue} {args.switch}={'on' if on else 'off'}: median of medians {statistics.median([m for m, _ in v]):.3f} ms, min {min(mi for _, mi in v):.3f}")


if __name__ == "__main__":
    main()
