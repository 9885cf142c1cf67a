"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Build-container cross-check (needs /root/reference):
times the IMPORTED reference (models.model.scene_representation + the unmodified MorpheuS.render_rays behind
the shims of oracle/make_golden.py) against the oracle port on the same inputs, same thread count, so that the
"port" CPU baseline bench.py reports on the GPU box can be related to the reference's own Python.

    python -m oracle.crosscheck_cpu [--rays 1024] [--samples 128]
"""
from __future__ import annotations

import argparse
import sys
import time
import types

sys.dont_write_bytecode = True
import torch

from morpheus_amd import synth
from oracle import field as of
from oracle import make_golden as mg


def timed(fn, reps=3):
    fn()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=128)
    a = ap.parse_args()
    mg.install_shims()
    import morpheus as ref_morpheus
    o, d, t, rid = synth.frame_rays(0, 128, 128)
    n = a.rays
    o, d, t, rid = o[:, :n], d[:, :n], t[:, :n], rid[:, :n]
    samples = of.uniform_samples(o[0], d[0], synth.ray_jitter(128 * 128)[:n], a.samples, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(128 * 128)
    st = synth.make_state("b")
    # reference
    m, cfg = mg.build_ref_model(st, None)
    m.eval()
    sampler = mg._PresetSampler()
    sampler.samples = samples
    fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200))

    def ref_step():
        m.zero_grad()
        res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, 32, 32, ambient_ratio=1.0, light_d=light, shading="albedo")
        (((res["image"][0] - timg[:n]) ** 2).mean() + ((res["depth"][0] - tdep[:n]) ** 2).mean()).backward()

    # oracle port
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in st.items()}
    f = of.OracleField(p, 1.01, None)

    def port_step():
        for v in p.values():
            if v.is_floating_point():
                v.grad = None
        res = of.render_rays(f, o, d, t, rid, samples, ambient_ratio=1.0, light_d=light, shading="albedo")
        (((res["image"][0] - timg[:n]) ** 2).mean() + ((res["depth"][0] - tdep[:n]) ** 2).mean()).backward()

    tr, tp = timed(ref_step), timed(port_step)
    print(f"threads={torch.get_num_threads()}  rays={n} x {a.samples}")
    print(f"imported reference (oracle hash grid stubbed in): {tr * 1e3:8.1f} ms  -> {n / tr:8.1f} rays/s")
    print(f"oracle port                                      : {tp * 1e3:8.1f} ms  -> {n / tp:8.1f} rays/s")
    print(f"port / reference time ratio: {tp / tr:.3f}")


if __name__ == "__main__":
    main()
