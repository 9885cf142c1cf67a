#!/usr/bin/env python
"""Headline benchmark: rays/sec, forward + backward, 128 samples/ray (BASELINE.json metric).

One "step" = one pass of the render_rays hot path over one batch of synthetic rays:
  sampler -> warp (deform/topo MLPs) -> hash grids -> sdf/colour MLPs + Laplace density ->
  transmittance compositor -> loss = MSE(image) + MSE(depth) -> backward to every parameter group
  (both hash tables included) -> [N>1: RCCL all-reduce of the flat gradient bucket] -> Adam step.
Workload at N=1: BASELINE configs[2] ("cfg3": snoopy.yaml, full deform field, 16384 rays x 128 samples);
N>1 = configs[4] ("cfg5"): one frame per rank (frames 0,25,...,175; weak scaling), no data-path collective other than
the gradient all-reduce (the hash tables' 6.4 of 7.45 MB go out on a side stream under the rest of backward).

`python bench.py --gpus N` launches itself: without torchrun's RANK/WORLD_SIZE in the environment and N > 1 it re-execs
through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per
GPU over RCCL; if the box has fewer GPUs than ranks the ranks share devices over gloo and the line says so).  Under
torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE as given.

Other workloads (parity / tier cases, not the headline): cfg2 (canonical field only), cfg3b (deform + FD normals),
train_real (the reference's real-view training step, morpheus.py:1147-1236: 2048 rays of one frame, occupancy-marched
ragged samples, albedo_normal, shipped regularisers, point loss, pose optimisation, occupancy refresh every 16 steps;
--glue reference keeps the reference's own caller-side loss code), train_virtual (its virtual-view step, :1393-1408: a whole
72^2 .. 180^2 novel view, random shading, SDS replaced by an injected pred_rgb gradient), density128 (forward-only
model.density on a 128^3 grid: export_mesh / update_occ_grid's query, morpheus.py:367-408).  The default run
(`python bench.py`) carries train_real (eager, HIP-graph replay, reference glue) and train_virtual (72^2, 180^2) beside the
cfg3 headline under the keys `train_real` / `train_virtual`, per-kernel timers off.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the step's LARGEST time item (the weight-gradient group included), SURVEY 8(d): top-level frac = algorithmic
                  FLOPs / HIP-event time / (dense peak of the matrix pipe the kernels issue on / slice products per fp32 MAC);
                  beside it the HBM view (parked bytes, algorithmic bytes, PMC traffic) and `per_kernel` with the same numbers
                  for every MLP entry of the step
  cpu_baseline -- the CPU oracle ("port") timed on a bounded sample of the same workload
  modes        -- N=1: the two arithmetic modes (b3 / f32, both fp32-faithful), each timed in its own process; the top-level
                  numbers are the faster one's
plus roofline_hashgrid (HBM-bound hash-grid stage, as north_star asks) and a per-kernel time table.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic MACs per sample point
MACS = dict(deform=77056, topo=76928, sdf=10880, color=8384)
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (the bf16x3 warp kernels issue 6 per fp32 MAC)
# what the bf16 matrix pipe SUSTAINS chip-wide on live operand slices in the kernels' configuration (A fragments from LDS, 8 waves per
# 96 KB copy), measured by tools/micro/mfma_bf16_rate.hip: 1575 TFLOP/s at a power-limited 1.68 GHz (profiles/r04_micro_mfma_bf16_rate.txt;
# 1788 from registers, 2223 on all-zero operands).  Reported BESIDE the nominal peak, never instead of it.
BF16_MFMA_SUSTAINED_TFLOPS = 1575.0
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
L2_PEAK_GBS = 34500.0             # MI355X_MICROARCH.md: aggregate L2 bandwidth
GRID_FWD_BYTES, GRID_BWD_BYTES = 1164, 2188   # per point per encoder
GRID_GATHERS_PER_POINT = 16 * 8   # 8 corners x 16 levels, one 8-byte row each -> one 64-byte L2 sector each (worst case)
WORKLOADS = ["cfg3", "cfg2", "cfg3b", "train_real", "train_virtual", "train_loop", "density128"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (32 for train_real: two occupancy refreshes; 60 for "
                                                            "cfg2, whose 5 ms steps need a longer timed region)")
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (15 for cfg2)")
    ap.add_argument("--workload", default="cfg3", choices=WORKLOADS,
                    help="cfg3: deform field, albedo (headline); cfg2: canonical only; cfg3b: deform + albedo_normal "
                         "shading (FD normals); train_real: the reference's real-view training step; train_virtual: its "
                         "virtual-view step (whole novel view, SDS replaced by an injected pred_rgb gradient); density128: "
                         "forward-only field query on a 128^3 grid")
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU per step (default 16384; 2048 for train_real; "
                                                           "--virtual-res squared for train_virtual)")
    ap.add_argument("--virtual-res", type=int, default=72,
                    help="train_virtual: side of the rendered novel view (datasets/dataset.py:538-541: novel_view_scale x 360 "
                         "= 72 at the start of training, 180 at novel_view_scale_final)")
    ap.add_argument("--glue", default="fused", choices=["fused", "reference", "reference_scoped"],
                    help="train_real: caller-side loss code.  fused = this build's rewrite (one launch per loss group, one operand "
                         "scope per step); reference = the reference's own operator chains and a loss.item() per step around the "
                         "swapped-in render_rays -- what INTEGRATION.md's three edits alone give")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run only: skip the extra workloads (train_real eager / replayed / reference glue, train_virtual) "
                         "that ride beside the cfg3 headline")
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one all-reduce after backward instead of the early "
                                                              "side-stream exchange of the hash-table gradients")
    ap.add_argument("--mode", default="auto", choices=["auto", "b3", "f32"],
                    help="arithmetic of the MLP kernels (morpheus_amd.ops).  auto at N=1 on a headline workload: every mode is "
                         "timed in its own process and reported in `modes`, the top-level line is the faster one "
                         "(b3 / f32); auto elsewhere = MORPHEUS_MLP or the library default (b3)")
    ap.add_argument("--modes", default="b3,f32", help="the modes `--mode auto` times, in this order")
    ap.add_argument("--detail-out", default=None,
                    help="where the FULL result object goes (kernel tables, notes, per-mode lines, allocator statistics); default "
                         "bench_detail.json beside this file.  stdout's LAST line is the compact object (< 6 KB) the driver parses")
    ap.add_argument("--graph", action="store_true",
                    help="capture one whole step (render fwd+bwd, all-reduce excluded, Adam) in a HIP graph and replay it; "
                         "disables the per-kernel event timers")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = {"train_real": 32, "train_virtual": 32, "train_loop": 6, "cfg2": 60}.get(args.workload, 20)
    if args.warmup is None:
        args.warmup = {"cfg2": 15, "train_loop": 2}.get(args.workload, 5)
    if args.rays is None:
        args.rays = {"train_real": 2048, "train_loop": 2048, "train_virtual": args.virtual_res ** 2}.get(args.workload, 128 * 128)
    if args.workload == "train_virtual":
        args.rays = args.virtual_res ** 2
    return args


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a torchrun environment: spawn N ranks of this script, one per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env.get("MORPHEUS_BENCH_STUB"):
        env["MORPHEUS_DIST_BACKEND"] = "gloo"
    elif "MORPHEUS_DIST_BACKEND" not in env:
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            # fewer GPUs than ranks (a 1-GPU box): the ranks share devices and exchange over gloo -- exercises the N>1
            # path, is NOT a scaling data point (the JSON line carries devices_visible and the backend)
            env["MORPHEUS_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(workload: str, n_rays: int, S: int, reps: int = 3):
    """Oracle (CPU restatement, kind 'port') fwd+bwd on a bounded sample of the same workload, in a
    subprocess per thread count (all-core runs of these small GEMMs are slower than 16-64 threads,
    so a few settings are tried and the best is reported with the thread count actually used)."""
    ncpu = os.cpu_count() or 1
    tried, best = [], None
    wl = workload
    for th in sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}):
        try:
            out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--workload", wl, "--rays", str(n_rays),
                                  "--samples", str(S), "--threads", str(th), "--reps", str(reps)], cwd=ROOT,
                                 capture_output=True, text=True, timeout=240)
            r = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:   # noqa: BLE001 - the baseline is informational; never fail the bench on it
            tried.append(dict(threads=th, error=str(e)[:80]))
            continue
        tried.append(dict(threads=th, rays_per_s=round(r["rays_per_s"], 1)))
        if best is None or r["rays_per_s"] > best["rays_per_s"]:
            best = r
    if best is None:
        return dict(value=None, unit="rays/s", cores=0, kind="port", sample="failed", tried=tried, cpu=_cpu_model())
    return dict(value=round(best["rays_per_s"], 1), unit="rays/s", cores=best["threads"], kind="port",
                sample=f"{n_rays} rays x {S} samples of the same frame/weights ({wl} render fwd+bwd), min of {reps} after 1 "
                       f"warm-up (oracle/field.py + oracle/hashgrid.c, OpenMP + torch CPU); host has {ncpu} logical CPUs",
                tried=tried, cpu=_cpu_model())


# ------------------------------------------------------------------------------------------------ workloads
def build_stub(args, rank, world):
    """MORPHEUS_BENCH_STUB=1 (CPU test of the launcher / timing / JSON plumbing, tests/test_bench_launcher.py): a toy
    step on the CPU with the same bucket + all-reduce + barrier structure.  Never used for a reported number."""
    import torch
    from morpheus_amd import dist as mdist
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 3))
    table = torch.nn.Parameter(torch.randn(256, 2))
    params = [table] + list(net.parameters())
    bucket = mdist.GradBucket(params)
    if not args.no_overlap:
        bucket.overlap_early([table])
    opt = torch.optim.Adam(params, lr=1e-3)
    x = torch.randn(args.rays, 6, generator=torch.Generator().manual_seed(rank))
    idx = torch.arange(args.rays) % 256

    def step():
        bucket.zero()
        loss = ((net(x) + table[idx].sum(-1, keepdim=True)) ** 2).mean()
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        return loss

    return dict(step=step, rays_per_step=args.rays, bucket=bucket, desc="stub (CPU toy step, plumbing test only)",
                samples=lambda: args.rays * args.samples)


def build_render_workload(args, rank, world, dev):
    import torch
    from morpheus_amd import dist as mdist
    from morpheus_amd import harness, synth
    from morpheus_amd.optim import FlatAdam
    cano = args.workload == "cfg2"
    model = harness.build_model("b", dev).train()
    cfg = model.config
    for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
        cfg["train"][k] = 0.0            # bare render path, as in BASELINE.md section 2
    frame = (25 * rank) % 200            # cfg5: frames 0,25,...,175, one per rank
    hw = int(round(args.rays ** 0.5))
    o, d, t, rid = [v.to(dev) for v in synth.frame_rays(frame, hw, hw)]
    N, S = o.shape[1], args.samples
    jitter = synth.ray_jitter(N).to(dev)
    rend = harness.make_renderer(model, S, jitter=jitter)
    light = torch.nn.functional.normalize(o[0] + torch.tensor([0.3, -0.2, 0.5], device=dev), dim=-1)
    timg, tdep = [v.to(dev) for v in synth.targets(N)]
    if args.graph:
        # a captured step freezes by-value kernel arguments (step count, learning rates): keep torch's capturable Adam
        opt = torch.optim.Adam(model.get_params_all(cfg["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15, fused=True,
                               capturable=True)
        bucket = mdist.GradBucket(model.parameters())
    else:
        # Adam of morpheus.py:154-155 over one flat bucket: one mh_adam_step launch per step, gradients in opt.bucket
        opt = FlatAdam(model.get_params_all(cfg["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
        bucket = opt.bucket
    if world > 1 and not args.no_overlap:
        bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings])
    shading = "albedo_normal" if args.workload == "cfg3b" else "albedo"

    def step():
        bucket.zero()
        res = rend.render_rays(o, d, t, rid, hw, hw, ambient_ratio=1.0, light_d=light, shading=shading, cano=cano)
        loss = harness.bench_loss(res, timg, tdep)
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        return loss

    desc = ("snoopy.yaml full deform field" + (" + FD normals (albedo_normal)" if args.workload == "cfg3b" else "")
            if not cano else "snoopy.yaml canonical field only") + \
        f", {N} rays x {S} samples per GPU, fwd+bwd+Adam ({args.workload}" + (", one frame per rank = cfg5" if world > 1 else "") + ")"
    return dict(step=step, rays_per_step=N, bucket=bucket, desc=desc, samples=lambda: N * S, frame=frame)


def build_train_real(args, rank, world, dev):
    """The reference's real-view training step (bench_support/trainstep.py restates morpheus.py:1147-1236 around render_rays)."""
    import torch
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer
    model = harness.build_model("b", dev).train()
    cfg = model.config                                   # shipped snoopy.yaml values: every regulariser on
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(dev)
    rend = HotPathRenderer(model, cfg, grid, 200)
    frames = trainstep.make_frames([(25 * rank + 8 * k) % 200 for k in range(8)], 256, 256, dev)
    ts = trainstep.RealViewTrainStep(rend, frames, ray_num=args.rays, glue=args.glue)
    ts.epoch = 1000                                       # mid-training: progressive level 0.75
    opt = FlatAdam(model.get_params_all(cfg["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
    bucket = opt.bucket
    if world > 1 and not args.no_overlap:
        bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings])
    with torch.no_grad():
        trainstep.warm_up_occupancy(ts)                   # trained-like occupancy from the field's own density
    ts.global_step = 4096                                 # past the estimator's warm-up: partial refresh every 16 steps
    occ = float(grid.binaries.float().mean())
    sample_log = []

    graphed = None
    if args.graph:
        assert world == 1, "--graph captures the single-GPU step (the gradient exchange stays outside a graph)"
        graphed = trainstep.GraphedRealViewStep(ts, bucket)
        graphed.prepare()                    # capture the capacity buckets of the frames' batches up front (not in the timed region)

        assert args.glue == "fused", "--graph replays this build's own step; the reference's glue synchronises every step"

        def step():
            loss = graphed()                 # render + losses + backward + gradient gather: one graph replay
            opt.step()
            sample_log.append(ts.last_samples)
            return loss
    elif args.glue in ("reference", "reference_scoped"):
        def step():
            bucket.zero()
            loss = ts()
            loss.backward()
            bucket.allreduce_mean()
            opt.step()
            sample_log.append(ts.last_samples)
            loss.item()                      # the reference reads the loss back on the host (morpheus.py:1426)
            return loss
    else:
        def step():
            bucket.zero()
            loss = ts()
            loss.backward()
            bucket.allreduce_mean()
            opt.step()
            sample_log.append(ts.last_samples)
            return loss

    desc = (f"snoopy.yaml real-view training step (morpheus.py:1147-1236): {args.rays} random rays of one frame per GPU, "
            f"occupancy-marched ragged samples (step 0.01, {occ * 100:.1f}% of 128^3 cells occupied), albedo_normal, "
            "normal_smooth_3d + normal_smoothness + code_reg, depth/mask/sdf/surface-point losses, pose optimisation, "
            "occupancy refresh every 16 steps, Adam")
    desc += {"fused": "; caller-side losses: this build's fused glue (bench_support/trainstep.py)",
             "reference": "; caller-side losses: the reference's own operator chains + loss.item() per step (INTEGRATION.md's three "
                          "edits only)",
             "reference_scoped": "; caller-side losses: the reference's own operator chains + loss.item() per step, plus ONE line: "
                                 "`with model.operand_scope():` around the step"}[args.glue]
    return dict(step=step, rays_per_step=args.rays, bucket=bucket, desc=desc + (", captured in a HIP graph" if graphed else ""),
                samples=lambda: (sum(sample_log[-args.steps:]) / max(len(sample_log[-args.steps:]), 1)), occupied=occ,
                graphed=graphed, glue=args.glue)


def pretouch_vram(dev, fraction: float = 0.85):
    """Touch most of the device's free memory once, then hand it back.  On a FRESH box the first process that reaches deep into
    the 288 GB pays for it inside its steps: a 180 x 180 virtual-view step (104 GB allocated at its peak, 243 GB reserved by the
    caching allocator) runs 270-350 ms in the first such process and 55.6 ms in every later one, whatever the allocator
    configuration (profiles/r04_first_touch_ab.txt) -- a property of the box's first use of that memory, not of the step, which a
    220 k-step run pays once.  So the workload touches the memory before its warm-up steps."""
    import torch
    free, _ = torch.cuda.mem_get_info(dev)
    n = int(free * fraction) // 4
    t0 = time.perf_counter()
    x = torch.empty(n, dtype=torch.float32, device=dev)
    x.fill_(0.0)
    torch.cuda.synchronize(dev)
    del x
    torch.cuda.empty_cache()
    return dict(touched_GB=round(n * 4 / 1e9, 1), seconds=round(time.perf_counter() - t0, 2))


def build_train_virtual(args, rank, world, dev):
    """The reference's VIRTUAL-view training step (morpheus.py:1393-1408: one step in eleven): all res x res rays of a novel view
    of a random frame, random lambertian / textureless shading at a random ambient ratio (:873-885), random / no background
    colour, orientation loss + normal_smooth_3d + normal_smoothness + code_reg, occupancy-marched ragged samples, no pose
    optimisation, no depth / mask terms; the SDS guidance is replaced by its interface, a fixed gradient on pred_rgb
    (trainstep.InjectedGuidance) -- the render forward + backward under it is what is timed.  With the deformation learning
    rates frozen for the virtual step (epochs <= freeze_epoch, :1394-1409) the reference steps the optimiser after it: so does
    this step, with those groups' learning rate at 0."""
    import torch
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer
    res = args.virtual_res
    touched = pretouch_vram(dev) if res * res > 20000 else None      # whole-view steps at the final resolution: ~100 GB per step
    # what the caching allocator may RESERVE: whole-view steps allocate ~105 GB and left 245 GB reserved after a dozen views
    # (morpheus_amd/chunking.py, profiles/r06_park_alloc.txt); 0.6 of the device bounds it at no cost.  The line INTEGRATION.md
    # section 5 recommends to the reference's trainer; MORPHEUS_BENCH_MEM_FRACTION=0 measures without it.
    mem_fraction = float(os.environ.get("MORPHEUS_BENCH_MEM_FRACTION", "0.6") or 0.6)
    if touched is not None and 0.0 < mem_fraction < 1.0:
        torch.cuda.memory.set_per_process_memory_fraction(mem_fraction, dev)
    model = harness.build_model("b", dev).train()
    cfg = model.config
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(dev)
    rend = HotPathRenderer(model, cfg, grid, 200)
    vs = trainstep.VirtualViewTrainStep(rend, res=res, seed=2024 + rank)
    vs.epoch = 1000                                      # mid-training: progressive level 0.75, random shading
    groups = model.get_params_all(cfg["train"]["lr"])
    for g in groups:                                      # freeze_lr_deform (morpheus.py:504-511)
        if g["name"] in ("code_deform", "decoder_deform", "decoder_topo"):
            g["lr"] = 0.0
    opt = FlatAdam(groups, betas=(0.9, 0.99), eps=1e-15)
    bucket = opt.bucket
    if world > 1 and not args.no_overlap:
        bucket.overlap_early([model.encoder.embeddings, model.encoder_c.embeddings])
    # trained-like occupancy from the field's own density (the real-view step's helper: same grid, same update rule)
    frames = trainstep.make_frames([(25 * rank) % 200], 64, 64, dev)
    warm = trainstep.RealViewTrainStep(rend, frames, ray_num=64)
    warm.epoch = 1000
    with torch.no_grad():
        trainstep.warm_up_occupancy(warm)
    vs.global_step = 4096
    occ = float(grid.binaries.float().mean())
    sample_log, shade_log = [], []
    inv_freq = 1.0 / cfg["train"]["virtual_freq"]

    def step():
        bucket.zero()
        loss = vs() * inv_freq                            # morpheus.py:1401
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        sample_log.append(vs.last_samples)
        shade_log.append(vs.last_shading[0])
        return loss

    desc = (f"snoopy.yaml virtual-view training step (morpheus.py:1393-1408): all {res} x {res} rays of one novel view per GPU, "
            f"occupancy-marched ragged samples (step 0.01, {occ * 100:.1f}% of 128^3 cells occupied), random lambertian / "
            "textureless shading + ambient ratio, orientation loss + normal_smooth_3d + normal_smoothness + code_reg, SDS replaced "
            "by an injected pred_rgb gradient (its interface), deform learning rates frozen, occupancy refresh every 16 steps, Adam")
    return dict(step=step, rays_per_step=res * res, bucket=bucket, desc=desc,
                samples=lambda: (sum(sample_log[-args.steps:]) / max(len(sample_log[-args.steps:]), 1)), occupied=occ,
                shadings=lambda: {k: shade_log[-args.steps:].count(k) for k in sorted(set(shade_log[-args.steps:]))},
                pretouch=touched)


def build_train_loop(args, rank, world, dev):
    """BASELINE configs[3] ("cfg4": teddy.yaml, the full optimisation loop, end-to-end iterations per second) with the one piece that
    is not available offline -- the Zero-1-to-3 UNet and its weights -- replaced by its interface (trainstep.InjectedGuidance: a fixed
    gradient on pred_rgb).  One "step" = one iteration of train_one_epoch's loop (morpheus.py:1390-1428): `virtual_freq` = 1
    virtual-view step (whole --virtual-res^2 view; past freeze_epoch its gradient is ACCUMULATED into the next optimiser step) followed
    by `real_freq` = 10 real-view steps of 2048 rays with an optimiser step each, then the reference's loss.item().  Both kinds of
    step share the model, the occupancy grid (refreshed every 16 global steps) and the optimiser."""
    import torch
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer
    cfg = harness.load_config("teddy")
    model = harness.build_model("b", dev, config=cfg).train()
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(dev)
    rend = HotPathRenderer(model, cfg, grid, 200)
    frames = trainstep.make_frames([(25 * rank + 8 * k) % 200 for k in range(8)], 256, 256, dev)
    ts = trainstep.RealViewTrainStep(rend, frames, ray_num=args.rays, glue=args.glue)
    vs = trainstep.VirtualViewTrainStep(rend, res=args.virtual_res, seed=2024 + rank)
    ts.epoch = vs.epoch = 1000                            # mid-training: level 0.75, random shading, freeze_lr off
    opt = FlatAdam(model.get_params_all(cfg["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
    bucket = opt.bucket      # (no early exchange here: the number of backward passes per optimiser step alternates between 2 and 1)
    with torch.no_grad():
        trainstep.warm_up_occupancy(ts)
    ts.global_step = 4096
    occ = float(grid.binaries.float().mean())
    n_virtual, n_real = cfg["train"]["virtual_freq"], cfg["train"]["real_freq"]
    sample_log = []
    graphed = None
    if args.graph:
        # the real-view steps that start from a zeroed gradient (9 of the 10) are replayed from HIP graphs; the first one after a
        # virtual-view step adds its gradient to the virtual view's and stays eager, as does the virtual-view step itself
        assert world == 1 and args.glue == "fused", "--graph replays this build's own single-GPU real-view step"
        graphed = trainstep.GraphedRealViewStep(ts, bucket)
        graphed.prepare()

    def step():
        bucket.zero()
        n = 0
        for _ in range(n_virtual):
            vs.global_step = ts.global_step
            loss = vs() * (1.0 / n_virtual)               # morpheus.py:1401
            loss.backward()                               # epoch > freeze_epoch: no optimiser step of its own (:1403-1409)
            loss = loss.detach()                          # (see below)
            ts.global_step = vs.global_step
            n += vs.last_samples
        for k in range(n_real):
            if k > 0 and graphed is not None:
                loss = graphed()                          # zero + render + losses + backward + gradient gather: one replay
            else:
                if k > 0:
                    bucket.zero()                         # the first real step's update carries the virtual-view gradient
                loss = ts()
                loss.backward()
                # keep the VALUE only: a loss that still holds its (freed) autograd graph keeps the parameters' AccumulateGrad
                # nodes alive on this stream, and a graph captured later in the run (a new capacity bucket) would run its
                # backward's accumulation on them -- outside the capturing stream (a 100-iteration soak crashed on exactly that)
                loss = loss.detach()
                bucket.allreduce_mean()
            opt.step()
            n += ts.last_samples
        sample_log.append(n)
        loss.item()                                       # morpheus.py:1426
        return loss

    rays_per_iter = n_virtual * args.virtual_res ** 2 + n_real * args.rays
    desc = (f"teddy.yaml optimisation loop (morpheus.py:1390-1428): per iteration {n_virtual} virtual-view step ({args.virtual_res} x "
            f"{args.virtual_res} rays, SDS replaced by an injected pred_rgb gradient: the Zero-1-to-3 UNet and its weights are not available "
            f"offline) + {n_real} real-view steps ({args.rays} rays each, Adam step each), occupancy refresh every 16 steps "
            f"({occ * 100:.1f}% of 128^3 cells occupied), loss.item() per iteration; glue: {args.glue}")
    return dict(step=step, rays_per_step=rays_per_iter, bucket=bucket,
                desc=desc + ("; real-view steps 2..%d of an iteration replayed from HIP graphs" % n_real if graphed else ""),
                samples=lambda: (sum(sample_log[-args.steps:]) / max(len(sample_log[-args.steps:]), 1)), occupied=occ, glue=args.glue,
                iters=True, graphed=graphed)


def build_density128(args, rank, world, dev):
    """Forward-only dense field query: export_mesh / update_occ_grid call model.density on grid points
    (morpheus.py:367-408, 905-913).  A 'step' = all 128^3 points in chunks of 2^21, no_grad, colour included."""
    import torch
    from morpheus_amd import harness
    model = harness.build_model("b", dev).eval()
    R = 128
    c = (torch.arange(R, device=dev).float() + 0.5) / R * 2 * model.bound - model.bound
    pts = torch.stack(torch.meshgrid(c, c, c, indexing="ij"), -1).reshape(-1, 3).contiguous()
    t = torch.full((1, 1), 25 / 200, device=dev)

    def step():
        with torch.no_grad():
            tot = 0.0
            for i in range(0, pts.shape[0], 1 << 21):
                out = model.density(pts[i:i + (1 << 21)], t, allow_shape=True)
                tot = tot + out["sdf"][0]
        return tot

    return dict(step=step, rays_per_step=pts.shape[0], bucket=None, samples=lambda: pts.shape[0],
                desc=f"forward-only model.density (warp + both hash grids + sdf/colour nets) on {R}^3 grid points, no_grad")


# ------------------------------------------------------------------------------------------------ roofline
PRODUCTS = {"f32": 1.0, "b3": 6.0}     # matrix-pipe slice products issued per fp32 MAC
PIPE = {"f32": ("fp32 MFMA (v_mfma_f32_32x32x2_f32)", FP32_MFMA_PEAK_TFLOPS),
        "b3": ("bf16 MFMA (v_mfma_f32_32x32x16_bf16)", BF16_MFMA_PEAK_TFLOPS)}
# kernel symbols (rocprofv3 names, profiles/*_pmc_summary.csv) behind each timed C-ABI entry, per arithmetic mode
SYMBOLS = {
    "mh_warp_fwd": {"f32": ["warp_fwd_kernel"], "b3": ["warp_fwd_b3_kernel<8, true>"]},
    # (round 6: at cfg3's size backward-data does not park dPre4 -- <8, false> -- and the two layer-4 weight-gradient launches
    #  regenerate it: wgrad_regen_b3_kernel)
    "mh_warp_bwd_data": {"f32": ["warp_bwd_kernel"], "b3": ["warp_bwd_b3_kernel<8, false>"]},
    "mh_mlp_wgrad[warp]": {"f32": ["wgrad_kernel<4, false>", "wgrad_kernel<2, false>", "wgrad_reduce_kernel"],
                           "b3": ["wgrad_regs_b3_kernel<4>", "wgrad_regen_b3_kernel", "wgrad_regs_b3_kernel<2>", "wgrad_kernel<4, true>",
                                  "wgrad_reduce_kernel"]},
    "mh_field_fwd": {"f32": ["field_fwd_kernel"], "b3": ["field_fwd_b3_kernel"]},
    # (the b3 mode runs the bf16x3 form of the fused kernels, the f32 mode the fp32-MFMA form; ops.FIELD_BWD)
    "mh_field_bwd_fused": {"f32": ["field_fused_color_kernel<false>", "field_fused_sdf_kernel<true, false>"],
                           "b3": ["field_fused_color_kernel<true>", "field_fused_sdf_kernel<true, true>"]},
}
# bytes per sample point.  "algorithmic" = what the operator must move if everything recomputable stayed on the chip (SURVEY 8d:
# inputs in, results out); "parked" = what THIS design moves by construction (activations / pre-activation gradients parked
# in HBM between the forward, backward-data and weight-gradient kernels, csrc/mlp_dev.h)
IO_BYTES = {
    # round 6 (b3 at this size): H0's 24 pad rows and dPre5's 29 | 30 are not written or read, dPre4 (2 x 128 rows) is not parked
    # (parked_f32: the fp32-MFMA kernels keep the round-5 tiles -- every row written, dPre4 parked; their weight gradients skip the pad rows)
    "mh_warp_fwd": dict(algorithmic=12 + 20, parked=4 * (40 + 2 * 640) + 4 * 40 + 12 + 20, parked_f32=4 * (64 + 2 * 640) + 4 * 40 + 12 + 20),
    "mh_warp_bwd_data": dict(algorithmic=12 + 20, parked=4 * (2 * 4 * 128 + 5) + 4 * 40 + 12 + 20, parked_f32=4 * 2 * 672 + 4 * 40 + 12 + 20),
    "mh_mlp_wgrad[warp]": dict(algorithmic=12 + 20, parked=4 * (2 * 40 + 2 * 5 * 128 + 2 * 4 * 128 + 5) + 2 * 8 + 20,
                               parked_f32=4 * (2 * 40 + 2 * 5 * 128 + 2 * 5 * 128 + 5)),
    "mh_field_fwd": dict(algorithmic=12 + 2 * 128 + 8 + 4 + 4 + 12, parked=4 * (96 + 64 * 5) + 4 * 8 + 12 + 2 * 128 + 8 + 20),
    "mh_field_bwd_fused": dict(algorithmic=12 + 20 + 2 * 128 + 8 + 12, parked=4 * (96 + 64 * 5) + 4 * 8 + 64 + 20 + 2 * 128 + 8 + 12),
}


def pmc_step_bytes(symbols, mode):
    """HBM bytes per STEP of the given kernel symbols, from the committed rocprofv3 --pmc passes of `python bench.py` in the
    same arithmetic mode (profiles/r0N_pmc_summary[_mode].csv: FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, KB
    units, separate passes).  None when no matching profile is committed."""
    import csv
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        # the un-suffixed summary is the default mode's (b3); another mode only matches its own passes
        for name in (f"{rnd}_pmc_summary_{mode}.csv",) + ((f"{rnd}_pmc_summary.csv",) if mode == "b3" else ()):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            try:
                rows = {r["k"]: r for r in csv.DictReader(open(path))}
            except Exception:      # noqa: BLE001
                continue
            tot, hit = 0.0, 0
            for sym in symbols:
                r = rows.get(sym)
                if r is None:
                    continue
                per = float(r["hbm_read_MB_per_launch"]) + float(r["hbm_write_MB_per_launch"])
                tot += per * float(r.get("launches_per_step", 1) or 1)
                hit += 1
            if hit:
                return round(tot * 1024 * 1024), name, hit == len(symbols)
    return None, None, False


def _kernel_roofline(name, e, flops_per_call, mode, M, workload, full_size):
    """SURVEY 8(d) numbers of ONE timed C-ABI entry: algorithmic FLOPs (2 per MAC of the reference's dense layers) / HIP-event
    time against the dense peak of the matrix pipe its kernels issue on divided by the slice products they issue per fp32 MAC;
    beside it the HBM view (bytes the design parks by construction / time vs 8 TB/s), SURVEY 8(d)'s algorithmic bytes and the
    PMC-measured traffic of the committed rocprofv3 passes."""
    secs = e["ms_per_step"] * 1e-3
    per_step_flops = flops_per_call * e["calls_per_step"]
    from morpheus_amd import ops as _ops      # the fused field backward is sliced in the b3 mode only (and only unless switched off)
    fused_fp32 = name == "mh_field_bwd_fused" and not (mode == "b3" and _ops.FIELD_BWD == "b3")
    kmode = "f32" if fused_fp32 else mode
    prod = PRODUCTS[kmode]
    pipe, unit_peak = PIPE[kmode]
    alg_tflops = per_step_flops / secs / 1e12
    peak = unit_peak / prod
    io = IO_BYTES[name]
    parked, algb = (io.get("parked_f32", io["parked"]) if kmode == "f32" else io["parked"]) * M, io["algorithmic"] * M
    hbm_gbs = parked / secs / 1e9
    traffic, src, complete = (pmc_step_bytes(SYMBOLS[name][mode], mode)
                              if (full_size and workload == "cfg3") else (None, None, False))
    return dict(kernel=name, launches_per_step=e["calls_per_step"], ms_per_step=e["ms_per_step"], mode=kmode,
                bound="mfma", achieved=round(alg_tflops, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(alg_tflops / peak, 4),
                peak_note=f"{pipe} dense peak {unit_peak} TFLOP/s (MI355X_MICROARCH.md) / {prod:g} slice product(s) issued per fp32 "
                          f"MAC = {peak:.1f} TFLOP/s of ALGORITHMIC fp32 work",
                issued_tflops=round(alg_tflops * prod, 1), flops_per_step=per_step_flops,
                frac_of_sustained=(None if kmode == "f32" else round(alg_tflops * prod / BF16_MFMA_SUSTAINED_TFLOPS, 4)),
                traffic=traffic, algorithmic_bytes=round(algb), parked_bytes=round(parked),
                waste_ratio_traffic_over_algorithmic=None if not traffic else round(traffic / algb, 1),
                traffic_source=(str(src) + ("" if complete else " (some kernels of the group missing in that file)")) if traffic else None,
                hbm=dict(bound="hbm", achieved=round(hbm_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_gbs / HBM_PEAK_GBS, 4),
                         bytes_per_step=round(parked)))


def build_roofline(ktab, mode, M, workload, full_size):
    """`roofline` is the step's LARGEST time item (weight-gradient group included); its top-level bound / achieved /
    peak / frac are the ALGORITHMIC-FLOP view -- FLOPs of the reference's dense layers / HIP-event time / (dense peak of the
    matrix pipe the kernels issue on / slice products per fp32 MAC).  The bytes this design parks, SURVEY 8(d)'s algorithmic
    bytes and the PMC-measured traffic ride beside it (`hbm`, `parked_bytes`, `algorithmic_bytes`, `traffic`,
    `waste_ratio_traffic_over_algorithmic`); `per_kernel` holds the same numbers for every MLP entry of the step."""
    warp_f = 2.0 * (MACS["deform"] + MACS["topo"]) * M
    field_f = 2.0 * (MACS["sdf"] + MACS["color"]) * M
    flops = {"mh_warp_fwd": warp_f, "mh_warp_bwd_data": warp_f, "mh_mlp_wgrad[warp]": warp_f, "mh_field_fwd": field_f,
             "mh_field_bwd_fused": 2.0 * field_f}
    if workload == "cfg3b":
        flops = {k: v for k, v in flops.items() if k.startswith(("mh_warp", "mh_mlp"))}   # field entries mix 1-point and 6-tap calls
    per = {k: _kernel_roofline(k, ktab[k], flops[k], mode, M, workload, full_size)
           for k in flops if k in ktab and ktab[k].get("calls_per_step", 0) > 0}
    if not per:
        return None
    name = max(per, key=lambda k: per[k]["ms_per_step"])
    out = dict(per[name])
    out["largest_item_of_step"] = True
    out["traffic_source"] = ("profiles/" + str(out["traffic_source"]) + " (committed rocprofv3 --pmc passes of this command; NOT measured in this run)") \
        if out["traffic"] else None
    out["traffic_note"] = "HBM bytes per step of this item's kernels, read from traffic_source" if out["traffic"] else \
        "null: no committed PMC profile matches this workload / mode (tools/gpu/r6_round.sh collects it)"
    out["algorithmic_bytes_note"] = ("SURVEY 8(d): inputs in + results out per point (everything else is recomputable on the chip); "
                                     "traffic far above it is the design's activation parking (DESIGN.md section 3)")
    out["hbm"]["note"] = ("bytes this design moves by construction (parked activation / pre-activation-gradient tiles, each row written "
                          "or read once); a streaming read reaches 5.6-6.6 TB/s and a streaming write 6.8 TB/s on this box "
                          "(profiles/r02_micro_hbm_read.txt, r02_micro_hbm_rates.txt)")
    out["frac_of_sustained_note"] = (f"issued slice-product TFLOP/s over the {BF16_MFMA_SUSTAINED_TFLOPS:g} TFLOP/s the bf16 / fp16 matrix pipe sustains "
                                     "chip-wide on live operands under the power budget (tools/micro/mfma_bf16_rate.hip, "
                                     "profiles/r04_micro_mfma_bf16_rate.txt); `frac` stays against the nominal dense peak")
    out["per_kernel"] = {k: {kk: v[kk] for kk in ("ms_per_step", "launches_per_step", "mode", "achieved", "peak", "unit", "frac", "frac_of_sustained",
                                                  "traffic", "algorithmic_bytes", "parked_bytes", "hbm")} for k, v in per.items()}
    return out


GRID_FWD_COMPULSORY, GRID_BWD_COMPULSORY = 12 + 128, 12 + 128 + 12     # x in + features out; grad in + x in + d/dx out (tables stay in L2)


def build_hash_roofline(ktab, M, workload, full_size, mode):
    """HBM roofline of the hash-grid stage (north_star).  `achieved` / `frac` = HBM bytes / time against 8 TB/s, where the bytes are
    the PMC-measured traffic of the committed rocprofv3 passes when this is the profiled workload (traffic_source names the file)
    and the COMPULSORY bytes otherwise (x in, features out: 140 B per point and table; a lower bound of the traffic) -- so `frac` is a
    utilisation and never passes 1.  SURVEY 8(d)'s yardstick (1164 / 2188 algorithmic bytes per point, which count the reference's
    eight gathers per level as memory traffic) rides beside it as `algorithmic_gbs`, a throughput in the reference's units."""
    binned = "mh_grid_encode_fwd_binned" in ktab           # calls of >= 2^20 points: binned first, a brick's rows staged in LDS
    if not binned and "mh_grid_encode_fwd" not in ktab:
        return None
    # table-passes per step: 2 tables x M points (+ the 6 FD taps of the SDF table in cfg3b: grouped launches, their own timer key)
    enc_points = (2 * M + (6 * M if workload == "cfg3b" else 0))
    fwd_key = "mh_grid_encode_fwd_binned" if binned else "mh_grid_encode_fwd"
    symbol = "grid_fwd_brick_kernel" if binned else "grid_fwd_kernel"
    pts_per_launch = (2 * M if binned else enc_points) / ktab[fwd_key]["calls_per_step"]
    secs = ktab[fwd_key]["avg_ms"] * 1e-3
    alg_gbs = GRID_FWD_BYTES * pts_per_launch / secs / 1e9
    compulsory = GRID_FWD_COMPULSORY * pts_per_launch
    traffic, src = None, None
    if full_size and workload != "cfg3b":
        t, src, _ = pmc_step_bytes([symbol], mode)
        traffic = None if t is None else round(t / ktab[fwd_key]["calls_per_step"])
    hbm_bytes = traffic if traffic is not None else compulsory
    gb = hbm_bytes / secs / 1e9
    roof = dict(kernel=symbol, bound="hbm", achieved=round(gb, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gb / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=("profiles/" + src) if traffic is not None else "none: achieved = compulsory bytes / time",
                compulsory_bytes=round(compulsory), algorithmic_bytes=round(GRID_FWD_BYTES * pts_per_launch),
                algorithmic_gbs=round(alg_gbs, 1), launch=fwd_key, ms_per_launch=ktab[fwd_key]["avg_ms"],
                gathers_per_s=round(GRID_GATHERS_PER_POINT * pts_per_launch / secs / 1e9, 1),
                note="frac = HBM bytes (PMC traffic of the committed profile, else the compulsory x-in + features-out bytes) / time / "
                     "8 TB/s.  algorithmic_gbs = SURVEY 8d's 1164 B/point / time: a throughput in the reference's units (its gathers "
                     "counted as memory traffic), not a utilisation -- both 3.2 MB tables are L2 resident and " +
                     ("the brick-binned form stages a brick's 4558 rows in LDS once (gathers_per_s counts LDS reads)" if binned else
                      "the gathers are cache-served (gathers_per_s, in G 8-byte gathers/s, is what bounds the kernel)") +
                     "; DESIGN.md section 3")
    bwd_name = "mh_grid_encode_bwd_binned" if "mh_grid_encode_bwd_binned" in ktab else "mh_grid_encode_bwd"
    if bwd_name in ktab:
        bsecs = ktab[bwd_name]["avg_ms"] * 1e-3
        bpts = enc_points / ktab[bwd_name]["calls_per_step"]
        btraffic = None
        if full_size and workload != "cfg3b" and bwd_name == "mh_grid_encode_bwd_binned":
            t, _, _ = pmc_step_bytes(["grid_bwd_brick_kernel<2, true>"], mode)
            btraffic = None if t is None else round(t / ktab[bwd_name]["calls_per_step"])
        bbytes = btraffic if btraffic is not None else GRID_BWD_COMPULSORY * bpts
        roof.update(bwd_kernel=bwd_name, bwd_achieved=round(bbytes / bsecs / 1e9, 1), bwd_frac=round(bbytes / bsecs / 1e9 / HBM_PEAK_GBS, 4),
                    bwd_traffic=btraffic, bwd_algorithmic_gbs=round(GRID_BWD_BYTES * bpts / bsecs / 1e9, 1), bwd_ms_per_launch=ktab[bwd_name]["avg_ms"],
                    bwd_note="bwd_frac as frac (measured or compulsory HBM bytes); bwd_algorithmic_gbs = the reference formulation's 2188 B "
                             "per point incl. the atomics' read-modify-write / time: the brick kernel accumulates on-chip")
    return roof


# ------------------------------------------------------------------------------------------------ one timed run
def run_one(args):
    """W warm-up steps, K timed steps between barrier + synchronize, MAX over ranks -> the result dict (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from morpheus_amd import dist as mdist
    from morpheus_amd import ops

    stub = bool(os.environ.get("MORPHEUS_BENCH_STUB"))
    if args.mode != "auto":
        ops.set_mlp_mode(args.mode)
    mode = ops.mlp_mode()
    rank, local, world = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if stub:
        dev = torch.device("cpu")
        wl = build_stub(args, rank, world)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        builder = {"train_real": build_train_real, "train_virtual": build_train_virtual, "train_loop": build_train_loop,
                   "density128": build_density128}.get(args.workload, build_render_workload)
        wl = builder(args, rank, world, dev)
    step = wl["step"]
    sync = (lambda: None) if stub else torch.cuda.synchronize
    if world > 1 and not stub and "MORPHEUS_DIST_BACKEND" not in os.environ:
        # a scaling data point is N ranks on N devices over RCCL; anything else (ranks sharing a device, gloo) has to be asked
        # for with MORPHEUS_DIST_BACKEND (bench.py's own launcher sets it when the box has fewer GPUs than ranks)
        if dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py --gpus {world}: backend {dist.get_backend()!r}, expected 'nccl' (RCCL); set MORPHEUS_DIST_BACKEND to override")
        devs = [None] * world
        dist.all_gather_object(devs, (socket.gethostname(), torch.cuda.current_device()))
        if len(set(devs)) != world:
            raise SystemExit(f"bench.py --gpus {world}: ranks share devices {devs}; set MORPHEUS_DIST_BACKEND=gloo for a functional run")

    timers_on = rank == 0 and not stub and not args.no_kernel_timers and not args.graph
    ops.TIMER.reset(enabled=timers_on)          # warm-up steps also fill the timer's event pool
    for _ in range(args.warmup):
        step()
    graph = None
    if args.graph and args.workload not in ("train_real", "train_loop"):     # those bring their own graphed step (trainstep.GraphedRealViewStep)
        assert world == 1 and args.workload in ("cfg3", "cfg2", "cfg3b"), "--graph captures the single-GPU fixed-shape step"
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                   # warm the allocator on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph(keep_graph=True)
        eager_step = step
        with torch.cuda.graph(graph):
            loss_g = eager_step().detach()
        ops.graph_replace_memset_nodes(graph)        # csrc/graph.hip: small memset nodes replay wrongly on ROCm 7.2
        graph.instantiate()

        def step():                                  # noqa: F811 -- replay the captured step
            graph.replay()
            return loss_g
    ops.TIMER.reset(enabled=timers_on)
    captures0 = wl["graphed"].n_captures if wl.get("graphed") is not None else 0
    if not stub:
        torch.cuda.reset_peak_memory_stats(dev)       # `peak_allocated_GB` below is the timed steps' own (not the VRAM pre-touch's)
    mem0 = None if stub else torch.cuda.memory_stats(dev)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    loss_acc = None
    for _ in range(args.steps):
        loss = step()
        if hasattr(loss, "detach"):          # mean loss of the timed steps, accumulated on the device (no sync)
            loss_acc = loss.detach().clone() if loss_acc is None else loss_acc + loss.detach()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    own_ms = elapsed / args.steps * 1e3          # this rank's own clock (the reported time is the MAX over ranks)
    if world > 1:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    timers = ops.TIMER.summary() if rank == 0 else {}
    ops.TIMER.reset(False)
    alloc = None
    if mem0 is not None:
        # the caching allocator inside the timed region (segment allocations, retries, footprint): a ragged workload's memory behaviour
        # is part of what its step costs and is not visible in the kernel table
        mem1 = torch.cuda.memory_stats(dev)
        alloc = {k: int(mem1.get(k, 0) - mem0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
        alloc["reserved_GB"] = round(mem1.get("reserved_bytes.all.current", 0) / 1e9, 2)
        alloc["peak_allocated_GB"] = round(mem1.get("allocated_bytes.all.peak", 0) / 1e9, 2)
        alloc["conf"] = os.environ.get("PYTORCH_HIP_ALLOC_CONF") or os.environ.get("PYTORCH_CUDA_ALLOC_CONF") or os.environ.get("PYTORCH_ALLOC_CONF")

    # who ran where: rank 0 prints it so that a multi-GPU run can be checked for "N ranks, N devices, backend nccl"
    n_dev = 0 if stub else torch.cuda.device_count()
    me = dict(rank=rank, device=str(dev), name=(torch.cuda.get_device_name(dev) if not stub else "cpu"), host_pid=os.getpid(),
              ms_per_step=round(own_ms, 3))       # per rank: a scaling run shows its stragglers
    rccl_version = None
    if not stub:
        try:                                      # which physical device each rank ran on, and the RCCL the exchange went through
            me["uuid"] = str(torch.cuda.get_device_properties(dev).uuid)[-12:]
        except Exception:      # noqa: BLE001
            me["uuid"] = None
        if world > 1 and dist.get_backend() == "nccl":
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:      # noqa: BLE001
                rccl_version = None
    ranks = [me]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks = gathered
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None
    N, S = wl["rays_per_step"], args.samples
    M = float(wl["samples"]())                     # sample points per step per GPU (ragged workloads: mean over the timed steps)
    ms_step = elapsed / args.steps * 1e3
    total_rays = N * world * args.steps
    render_wl = args.workload in ("cfg3", "cfg2", "cfg3b")
    ktab = {}
    for name, (calls, total_ms) in sorted(timers.items(), key=lambda kv: -kv[1][1]):
        avg_ms = total_ms / max(calls, 1)
        ktab[name] = dict(calls_per_step=round(calls / args.steps, 3), avg_ms=round(avg_ms, 4),
                          ms_per_step=round(total_ms / args.steps, 4))
    full = render_wl and N * S == 128 * 128 * 128
    roofline = build_roofline(ktab, mode, M, args.workload, full) if render_wl else None
    if roofline is not None and args.workload != "cfg3b":
        step_flops = 3.0 * (2.0 * (MACS["deform"] + MACS["topo"]) * M * (0 if args.workload == "cfg2" else 1) +
                            2.0 * (MACS["sdf"] + MACS["color"]) * M)
        roofline["whole_step"] = dict(flops=step_flops, tflops=round(step_flops / (ms_step * 1e-3) / 1e12, 2),
                                      frac_of_fp32_mfma_peak=round(step_flops / (ms_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                      note="algorithmic FLOP/s of the whole step (SURVEY 8d: 6 FLOPs per MAC fwd+bwd) over the native "
                                           "fp32 MFMA peak, 157.3 TFLOP/s: the speed of light of an all-fp32-MFMA step, a continuity figure")
    roof_hash = build_hash_roofline(ktab, M, args.workload, full, mode) if render_wl else None
    bucket = wl["bucket"]
    backend = dist.get_backend() if world > 1 else None
    headline = args.workload in ("cfg3", "cfg2", "cfg3b")
    kernel_sum = round(sum(v["ms_per_step"] for v in ktab.values()), 4)
    out = {
        "metric": "rays/sec (fwd+bwd, 128 samples/ray)" if headline else
                  {"train_real": "rays/sec (real-view training step, ragged occupancy samples)",
                   "train_virtual": "rays/sec (virtual-view training step, whole novel view, ragged occupancy samples)",
                   "train_loop": "rays/sec over whole iterations of the optimisation loop (1 virtual + 10 real training steps, SDS stubbed)",
                   "density128": "points/sec (forward-only field query)"}[args.workload],
        "value": round(total_rays / elapsed, 1), "unit": "rays/s" if args.workload != "density128" else "points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ops.MODE_DTYPE[mode], "data": "synthetic",
        "config": {"workload": wl["desc"], "rays_per_gpu": N, "samples_per_ray": S if headline else round(M / max(N, 1), 1),
                   "sample_points_per_step_per_gpu": round(M),
                   "parallelism": f"dp{world}" + ("" if bucket is None else
                                                  f" (rays/frames sharded, {bucket.nbytes / 1e6:.2f} MB of gradients all-reduced "
                                                  f"per step" + (", hash-table range early on a side stream" if world > 1 and
                                                                 not args.no_overlap else "") + ")"),
                   "grad_bucket_MB": None if bucket is None else round(bucket.nbytes / 1e6, 2),
                   "world_size": world, "backend": backend, "devices_visible": n_dev,
                   "rccl_version": rccl_version, "device_uuids": [r.get("uuid") for r in ranks] if world > 1 and not stub else None,
                   "ranks_share_devices": bool(world > 1 and n_dev < world), "ranks": ranks,
                   "mlp_mode": mode,
                   "mlp_arithmetic": {"b3": "fp32 values everywhere; warp nets (forward, backward-data, weight gradients) and the field "
                                            "nets (forward, fused backward): EXACT three-way bf16 split of both operands (all 24 "
                                            "significand bits), six slice products per MAC on the bf16 matrix pipe, fp32 accumulate"
                                            + ("" if ops.FIELD_BWD == "b3" else f"; field backward: MORPHEUS_FIELD_BWD={ops.FIELD_BWD} "
                                               "(native fp32 MFMA for " + ("the colour + sdf pass" if ops.FIELD_BWD == "sdf" else "both passes") + ")"),
                                      "f32": "native fp32 MFMA (v_mfma_f32_32x32x2_f32) in every MLP kernel"}[mode],
                   "weights": "closed-form state b", "loss": float(loss.item()) if hasattr(loss, "item") else float(loss),
                   "loss_mean_of_timed_steps": None if loss_acc is None else float(loss_acc.item()) / args.steps},
        "roofline": roofline, "roofline_hashgrid": roof_hash, "kernels": ktab,
        "kernel_sum_ms_per_step": kernel_sum, "timed_launches_per_step": round(sum(v["calls_per_step"] for v in ktab.values()), 1),
    }
    if wl.get("iters"):
        out["iters_per_s"] = round(args.steps / elapsed, 3)       # BASELINE configs[3]'s unit: end-to-end iterations per second
        out["train_steps_per_s"] = round(args.steps * 11 / elapsed, 2)
    if "occupied" in wl:
        out["config"]["occupied_fraction"] = round(wl["occupied"], 4)
    if "glue" in wl:
        out["config"]["glue"] = wl["glue"]
    if "shadings" in wl:
        out["config"]["shadings_of_timed_steps"] = wl["shadings"]()
    if wl.get("pretouch") is not None:
        out["config"]["vram_pretouch_before_warmup"] = wl["pretouch"]
    out["config"]["allocator_in_timed_region"] = alloc
    if not stub:
        from morpheus_amd import chunking
        out["config"]["parked_memory_bound"] = dict(cap_GB=os.environ.get("MORPHEUS_MAX_PARK_GB") or round(chunking.park_cap_bytes(dev) / 1e9, 1),
                                                   query_calls=chunking.STATS["calls"], chunked_calls=chunking.STATS["chunked_calls"],
                                                   chunks=chunking.STATS["chunks"], rerun_rows=chunking.STATS["rerun_rows"],
                                                   note="morpheus_amd/chunking.py: over the cap a query runs in row chunks, all but the "
                                                        "last re-made in backward (whole process, warm-up included)")
    out["config"]["kernel_timers"] = bool(timers_on)      # per-C-ABI-call HIP events inside the timed region (host cost per call)
    if wl.get("graphed") is not None:
        g = wl["graphed"]
        out["config"]["hip_graph"] = dict(capacity_buckets=sorted({k[0] for k in g.graphs}), bucket_step=g.bucket_step,
                                          capacity_of_last_step=g.last_capacity, samples_of_last_step=g.last_samples,
                                          overflowed_batches=int(g.check_overflow()) + g.overflows, margin=g.margin,
                                          graphs_captured=g.n_captures, captures_inside_the_timed_region=g.n_captures - captures0,
                                          memset_nodes_replaced_by_fill_kernels=g.memset_nodes_replaced, nodes_per_replayed_step=g.last_graph_nodes,
                                          graphs_evicted=g.n_evicted,
                                          note="sample_points_per_step_per_gpu is the mean COUNTED sample number of the batches (as in the "
                                               "eager run); the kernels ran on the bucket capacity (capacity_of_last_step: padding "
                                               "included); each batch is drawn and counted one step ahead on a side stream to pick its bucket")
    elif args.graph:
        out["config"]["hip_graph"] = dict(note="render + loss + backward + Adam captured once, replayed per step")
    out["cpu_baseline"] = None
    if world > 1:
        dist.destroy_process_group()
    return out


# ------------------------------------------------------------------------------------------------ both arithmetic modes

def _run_child(flags, mode, timeout, env=None):
    """One bench.py child process (fresh allocator, timers and operand caches); returns (full result object or None, error text).
    The child writes its FULL object to a temporary --detail-out file; its stdout carries the compact line only."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="morpheus_bench_", suffix=".json")
    os.close(fd)
    try:
        cmd = [sys.executable, os.path.abspath(__file__)] + list(flags) + ["--detail-out", path]
        run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                             env={**os.environ, "MORPHEUS_MLP": mode, **(env or {})})
        if run.returncode != 0 or os.path.getsize(path) == 0:
            return None, (run.stderr or run.stdout)[-400:]
        return json.load(open(path)), None
    except Exception as e:      # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"[:400]
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass

FAITHFUL = ("b3", "f32")      # modes whose operands keep all 24 significand bits of the reference's fp32 (models/decoders.py:59-64)


def run_modes(args, argv):
    """N = 1, headline workloads, --mode auto: time every arithmetic mode in its OWN process (fresh allocator, timers and
    operand caches; the same isolation the CPU baseline gets) and report them side by side.  The top-level line is the
    faster of the two (both keep all 24 operand bits; `modes_ms_per_step` carries both)."""
    base, skip = [], False
    for a in argv:                   # every child gets its own --detail-out
        if skip:
            skip = False
        elif a == "--detail-out":
            skip = True
        elif not a.startswith("--detail-out="):
            base.append(a)
    results, errors = {}, {}
    for m in args.modes.split(","):
        r, err = _run_child(base + ["--mode", m, "--no-cpu-baseline"], m, 900)
        if r is None:
            errors[m] = err
        else:
            results[m] = r
    faithful = [m for m in FAITHFUL if m in results]
    if not faithful:
        raise SystemExit("bench.py: no fp32-faithful mode produced a result: " + json.dumps(errors))
    best = min(faithful, key=lambda m: results[m]["ms_per_step"])
    out = dict(results[best])
    out["headline_mode"] = best
    out["headline_rule"] = ("value / ms_per_step / dtype / roofline / kernels are those of the faster fp32-faithful mode (b3: exact "
                            "3 x bf16 operand split -- a deviation from SURVEY section 7's 'no split-precision tricks', DESIGN.md section 3; "
                            "f32: native fp32 MFMA, the caveat-free number, in `modes`)")
    out["modes"] = {m: {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline", "kernels",
                                          "kernel_sum_ms_per_step")} | {"loss": r["config"]["loss"],
                                                                        "fp32_faithful": m in FAITHFUL}
                    for m, r in results.items()}
    if errors:
        out["mode_errors"] = errors
    # the same step of the headline mode replayed from ONE captured HIP graph (launch gaps of the eager step removed), reported
    # beside the eager measurement, which stays the headline
    r, err = _run_child(base + ["--mode", best, "--graph", "--no-cpu-baseline", "--no-kernel-timers"], best, 600)
    if r is not None:
        out["hip_graph_replay"] = {"mode": best, "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                   "steps": r["steps"], "loss": r["config"]["loss"],
                                   "note": "render + loss + backward + Adam captured once and replayed (memset nodes replaced by fill "
                                           "kernels, csrc/graph.hip); the eager step above is the reported value"}
    else:
        out["hip_graph_replay"] = {"error": err}
    if args.workload == "cfg3" and not args.no_extras:
        out.update(run_extras(best))
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_rays, args.samples)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


def run_extras(mode):
    """The training-step workloads beside the cfg3 headline, each in its own process, per-kernel timers OFF (a real-view step
    issues ~360 launches: the event pairs around its 50 C-ABI calls cost it ~0.25 ms of 7.35, nothing on cfg3's 13), library defaults for
    steps / warm-up:
      train_real     the reference's real-view step (morpheus.py:1147-1236) -- eager with this build's fused caller-side glue,
                     the same replayed from HIP graphs, and eager with the REFERENCE's own glue + loss.item() per step (what
                     INTEGRATION.md's three edits alone give);
      train_virtual  its virtual-view step (:1393-1408) at 72 x 72 and 180 x 180 rays, SDS replaced by an injected pred_rgb
                     gradient (the UNet is not part of the hot path and its weights are not available offline)."""
    def sub(flags, timeout=600, env=None):
        r, err = _run_child(["--gpus", "1", "--mode", mode, "--no-kernel-timers", "--no-cpu-baseline"] + flags, mode, timeout, env)
        if r is None:
            return {"error": err}
        keep = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "iters_per_s", "train_steps_per_s") if k in r}
        c = r["config"]
        keep.update(workload=c["workload"], rays_per_gpu=c["rays_per_gpu"], sample_points_per_step=c["sample_points_per_step_per_gpu"],
                    kernel_timers=c.get("kernel_timers"), loss_mean_of_timed_steps=c.get("loss_mean_of_timed_steps"))
        for k in ("glue", "hip_graph", "shadings_of_timed_steps", "occupied_fraction", "allocator_in_timed_region", "vram_pretouch_before_warmup",
                  "parked_memory_bound"):
            if k in c:
                keep[k] = c[k]
        return keep

    real = ["--workload", "train_real"]
    return {"train_real": {"eager_fused_glue": sub(real), "hip_graph_replay": sub(real + ["--graph"]),
                           "eager_reference_glue": sub(real + ["--glue", "reference"]),
                           "eager_reference_glue_one_scope": sub(real + ["--glue", "reference_scoped"]),
                           "note": "rays/s of the reference's real-view training step, 2048 rays per step; `eager_reference_glue` "
                                   "is the drop-in number (reference caller untouched), `..._one_scope` adds one `with "
                                   "model.operand_scope():` line around the step, the other two need this build's caller"},
            "train_virtual": {"res72": sub(["--workload", "train_virtual", "--virtual-res", "72"]),
                              "res180": sub(["--workload", "train_virtual", "--virtual-res", "180"]),
                              "res180_cap64": sub(["--workload", "train_virtual", "--virtual-res", "180"], env={"MORPHEUS_MAX_PARK_GB": "64"}),
                              "note": "rays/s of the reference's virtual-view training step (render fwd + bwd + Adam under an "
                                      "injected pred_rgb gradient standing for Zero-1-to-3 SDS)"},
            "train_loop": {"res72": sub(["--workload", "train_loop", "--virtual-res", "72"]),
                           "res72_hip_graph": sub(["--workload", "train_loop", "--virtual-res", "72", "--graph"]),
                           "note": "BASELINE configs[3] (teddy.yaml, end-to-end iterations per second) with the UNet replaced by its interface: "
                                   "iters_per_s of (1 virtual + 10 real) training steps; res72_hip_graph: the nine real-view steps of an "
                                   "iteration that start from a zeroed gradient replayed from HIP graphs"}}


# ------------------------------------------------------------------------------------------------ the driver's line
COMPACT_LIMIT = 6000          # bytes; the driver reads the LAST stdout line and keeps an 8 KB tail (round 4's 34 KB line came back unparsed)
SHORT_DTYPE = {"b3": "f32 (exact 3 x bf16 operand split on the bf16 MFMA pipe, fp32 accumulate)",
               "f32": "f32 (native fp32 MFMA)"}


def _num(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(out, detail_path=None):
    """The object bench.py prints as its LAST stdout line: the contract's keys, `roofline` and `cpu_baseline`, and ONE number per
    extra.  Everything else (kernel tables, notes, per-mode lines, allocator statistics) is in the detail file."""
    c = out.get("config", {})
    mode = c.get("mlp_mode")
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "data")}
    line["dtype"] = SHORT_DTYPE.get(mode, out.get("dtype"))
    line["config"] = {"workload": str(c.get("workload", ""))[:160], "rays_per_gpu": c.get("rays_per_gpu"),
                      "samples_per_ray": c.get("samples_per_ray"), "world_size": c.get("world_size"), "backend": c.get("backend"),
                      "devices_visible": c.get("devices_visible"), "mlp_mode": mode,
                      "parallelism": str(c.get("parallelism", "")).split(" (")[0],
                      "grad_bucket_MB": c.get("grad_bucket_MB"), "kernel_timers": c.get("kernel_timers")}
    for k in ("rccl_version", "device_uuids", "ranks_share_devices", "glue", "occupied_fraction"):
        if c.get(k) not in (None, False):
            line["config"][k] = c[k]
    if c.get("ranks") and len(c["ranks"]) > 1:
        line["config"]["rank_ms_per_step"] = [r.get("ms_per_step") for r in c["ranks"]]
    ro = out.get("roofline")
    if ro:
        line["roofline"] = {k: ro.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                   "ms_per_step", "algorithmic_bytes", "parked_bytes", "frac_of_sustained")}
        line["roofline"]["per_kernel"] = {k: {"ms": v.get("ms_per_step"), "frac": v.get("frac")} for k, v in ro.get("per_kernel", {}).items()}
        line["roofline"]["whole_step_tflops"] = _num(ro, "whole_step", "tflops")
    else:
        line["roofline"] = None
    rh = out.get("roofline_hashgrid")
    if rh:
        line["roofline_hashgrid"] = {k: rh.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                            "compulsory_bytes", "algorithmic_gbs", "bwd_achieved", "bwd_frac",
                                                            "bwd_algorithmic_gbs")}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                                "kind": cb.get("kind"), "cpu": cb.get("cpu"), "sample": str(cb.get("sample", ""))[:120]}
    if "speedup_vs_cpu_baseline" in out:
        line["speedup_vs_cpu_baseline"] = out["speedup_vs_cpu_baseline"]
    if "modes" in out:
        line["headline_mode"] = out.get("headline_mode")
        line["modes_ms_per_step"] = {m: r.get("ms_per_step") for m, r in out["modes"].items()}
    if "mode_errors" in out:
        line["mode_errors"] = {m: str(e)[-120:] for m, e in out["mode_errors"].items()}
    if "hip_graph_replay" in out:
        line["hip_graph_replay_ms"] = out["hip_graph_replay"].get("ms_per_step", "error")
    tr = out.get("train_real")
    if tr:
        line["train_real_ms"] = {short: _num(tr, key, "ms_per_step") for short, key in
                                 (("eager", "eager_fused_glue"), ("graph", "hip_graph_replay"), ("reference_glue", "eager_reference_glue"),
                                  ("reference_scoped", "eager_reference_glue_one_scope"))}
    tv = out.get("train_virtual")
    if tv:
        line["train_virtual_ms"] = {"72": _num(tv, "res72", "ms_per_step"), "180": _num(tv, "res180", "ms_per_step"),
                                    "180_reserved_GB": _num(tv, "res180", "allocator_in_timed_region", "reserved_GB"),
                                    "180_peak_allocated_GB": _num(tv, "res180", "allocator_in_timed_region", "peak_allocated_GB"),
                                    "180_cap64": _num(tv, "res180_cap64", "ms_per_step"),
                                    "180_cap64_reserved_GB": _num(tv, "res180_cap64", "allocator_in_timed_region", "reserved_GB")}
    tl = out.get("train_loop")
    if tl:
        line["train_loop_iters_per_s"] = _num(tl, "res72", "iters_per_s")
        line["train_loop_hip_graph_iters_per_s"] = _num(tl, "res72_hip_graph", "iters_per_s")
    for k in ("iters_per_s", "train_steps_per_s", "kernel_sum_ms_per_step"):
        if k in out:
            line[k] = out[k]
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:      # never again: drop the optional parts rather than print an unparseable line
        for k in ("roofline_hashgrid", "train_real_ms", "train_virtual_ms", "modes_ms_per_step", "mode_errors"):
            line.pop(k, None)
        if line.get("roofline"):
            line["roofline"].pop("per_kernel", None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 8192, len(text)
    return text


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))
    stub = bool(os.environ.get("MORPHEUS_BENCH_STUB"))
    headline = args.workload in ("cfg3", "cfg2", "cfg3b")
    if args.gpus == 1 and args.mode == "auto" and headline and not stub and not args.graph:
        out = run_modes(args, argv)
    else:
        out = run_one(args)
        if out is not None and args.gpus == 1 and headline and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_rays, args.samples)
            if out["cpu_baseline"]["value"]:
                out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if out is not None:
        detail = args.detail_out or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail, "w") as f:
                json.dump(out, f)
        except OSError as e:           # a read-only checkout must not cost the line
            print(f"bench.py: detail file not written ({e})", file=sys.stderr)
            detail = None
        print(compact_line(out, detail), flush=True)


if __name__ == "__main__":
    main()
