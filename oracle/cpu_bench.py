"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU baseline leg of bench.py ("kind": "port").

Times the CPU oracle (oracle/field.py + oracle/hashgrid.c) forward+backward on a bounded sample of
the benchmark workload and prints one JSON line.  Run as a subprocess so that the thread count
(OMP_NUM_THREADS + torch.set_num_threads) is fixed before any OpenMP runtime starts.

    python -m oracle.cpu_bench --workload cfg3 --rays 1024 --samples 128 --threads 32 --reps 2
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = str(a.threads)
    os.environ["OMP_WAIT_POLICY"] = "PASSIVE"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    torch.set_num_threads(a.threads)
    from morpheus_amd import synth
    from oracle import field as of
    o, d, t, rid = synth.frame_rays(0, 128, 128)
    n = a.rays
    o, d, t, rid = o[:, :n], d[:, :n], t[:, :n], rid[:, :n]
    samples = of.uniform_samples(o[0], d[0], synth.ray_jitter(128 * 128)[:n], a.samples, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(128 * 128)
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in synth.make_state("b").items()}
    f = of.OracleField(p, 1.01, None)
    best = float("inf")
    for i in range(a.reps + 1):
        for v in p.values():
            if v.is_floating_point():
                v.grad = None
        t0 = time.perf_counter()
        res = of.render_rays(f, o, d, t, rid, samples, ambient_ratio=1.0, light_d=light, shading="albedo",
                             cano=(a.workload == "cfg2"))
        loss = ((res["image"][0] - timg[:n]) ** 2).mean() + ((res["depth"][0] - tdep[:n]) ** 2).mean()
        loss.backward()
        dt = time.perf_counter() - t0
        if i > 0:
            best = min(best, dt)
    print(json.dumps(dict(rays_per_s=n / best, seconds=best, threads=a.threads, rays=n, samples=a.samples)))


if __name__ == "__main__":
    main()
