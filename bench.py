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
ragged samples, albedo_normal, shipped regularisers, point loss, pose optimisation, occupancy refresh every 16 steps),
density128 (forward-only model.density on a 128^3 grid: export_mesh / update_occ_grid's query, morpheus.py:367-408).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel: algorithmic FLOPs per launch / HIP-event launch time vs the fp32 MFMA peak
  cpu_baseline -- the CPU oracle ("port") timed on a bounded sample of the same workload
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
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
L2_PEAK_GBS = 34500.0             # MI355X_MICROARCH.md: aggregate L2 bandwidth
GRID_FWD_BYTES, GRID_BWD_BYTES = 1164, 2188   # per point per encoder
GRID_GATHERS_PER_POINT = 16 * 8   # 8 corners x 16 levels, one 8-byte row each -> one 64-byte L2 sector each (worst case)
WORKLOADS = ["cfg3", "cfg2", "cfg3b", "train_real", "density128"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (32 for train_real: two occupancy refreshes; 60 for "
                                                            "cfg2, whose 5 ms steps need a longer timed region)")
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (15 for cfg2)")
    ap.add_argument("--workload", default="cfg3", choices=WORKLOADS,
                    help="cfg3: deform field, albedo (headline); cfg2: canonical only; cfg3b: deform + albedo_normal "
                         "shading (FD normals); train_real: the reference's real-view training step; density128: "
                         "forward-only field query on a 128^3 grid")
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU per step (default 16384; 2048 for train_real)")
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one all-reduce after backward instead of the early "
                                                              "side-stream exchange of the hash-table gradients")
    ap.add_argument("--graph", action="store_true",
                    help="capture one whole step (render fwd+bwd, all-reduce excluded, Adam) in a HIP graph and replay it; "
                         "disables the per-kernel event timers")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = {"train_real": 32, "cfg2": 60}.get(args.workload, 20)
    if args.warmup is None:
        args.warmup = 15 if args.workload == "cfg2" else 5
    if args.rays is None:
        args.rays = 2048 if args.workload == "train_real" else 128 * 128
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
def cpu_baseline(workload: str, n_rays: int, S: int, reps: int = 2):
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
        return dict(value=None, unit="rays/s", cores=0, kind="port", sample="failed", tried=tried)
    return dict(value=round(best["rays_per_s"], 1), unit="rays/s", cores=best["threads"], kind="port",
                sample=f"{n_rays} rays x {S} samples of the same frame/weights ({wl} render fwd+bwd), min of {reps} after 1 "
                       f"warm-up (oracle/field.py + oracle/hashgrid.c, OpenMP + torch CPU); host has {ncpu} logical CPUs",
                tried=tried)


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
    """The reference's real-view training step (morpheus_amd/trainstep.py restates morpheus.py:1147-1236 around render_rays)."""
    import torch
    from morpheus_amd import harness, trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer
    model = harness.build_model("b", dev).train()
    cfg = model.config                                   # shipped snoopy.yaml values: every regulariser on
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(dev)
    rend = HotPathRenderer(model, cfg, grid, 200)
    frames = trainstep.make_frames([(25 * rank + 8 * k) % 200 for k in range(8)], 256, 256, dev)
    ts = trainstep.RealViewTrainStep(rend, frames, ray_num=args.rays)
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
    return dict(step=step, rays_per_step=args.rays, bucket=bucket, desc=desc,
                samples=lambda: (sum(sample_log[-args.steps:]) / max(len(sample_log[-args.steps:]), 1)), occupied=occ)


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


# ------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))

    import torch
    import torch.distributed as dist
    from morpheus_amd import dist as mdist
    from morpheus_amd import ops

    stub = bool(os.environ.get("MORPHEUS_BENCH_STUB"))
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
        builder = {"train_real": build_train_real, "density128": build_density128}.get(args.workload, build_render_workload)
        wl = builder(args, rank, world, dev)
    step = wl["step"]
    sync = (lambda: None) if stub else torch.cuda.synchronize

    timers_on = rank == 0 and not stub and not args.no_kernel_timers and not args.graph
    ops.TIMER.reset(enabled=timers_on)          # warm-up steps also fill the timer's event pool
    for _ in range(args.warmup):
        step()
    graph = None
    if args.graph:
        assert world == 1 and args.workload in ("cfg3", "cfg2", "cfg3b"), "--graph captures the single-GPU fixed-shape step"
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                   # warm the allocator on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        eager_step = step
        with torch.cuda.graph(graph):
            loss_g = eager_step()

        def step():                                  # noqa: F811 -- replay the captured step
            graph.replay()
            return loss_g
    ops.TIMER.reset(enabled=timers_on)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    timers = ops.TIMER.summary() if rank == 0 else {}
    ops.TIMER.reset(False)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    N, S = wl["rays_per_step"], args.samples
    M = float(wl["samples"]())                     # sample points per step per GPU (ragged workloads: mean over the timed steps)
    ms_step = elapsed / args.steps * 1e3
    total_rays = N * world * args.steps
    render_wl = args.workload in ("cfg3", "cfg2", "cfg3b")
    # algorithmic FLOPs per launch of each timed C-ABI call (2 FLOPs per MAC; bwd-data = wgrad = fwd)
    warp_f = 2.0 * (MACS["deform"] + MACS["topo"]) * M
    field_f = 2.0 * (MACS["sdf"] + MACS["color"]) * M
    flops = {"mh_warp_fwd": warp_f, "mh_warp_bwd_data": warp_f, "mh_mlp_wgrad[warp]": warp_f,
             "mh_field_fwd": field_f, "mh_field_bwd_data": field_f, "mh_mlp_wgrad[field]": field_f} if render_wl and \
        args.workload != "cfg3b" else ({"mh_warp_fwd": warp_f, "mh_warp_bwd_data": warp_f, "mh_mlp_wgrad[warp]": warp_f}
                                       if args.workload == "cfg3b" else {})
    ktab, dominant = {}, None
    for name, (calls, total_ms) in sorted(timers.items(), key=lambda kv: -kv[1][1]):
        avg_ms = total_ms / max(calls, 1)
        ktab[name] = dict(calls_per_step=round(calls / args.steps, 3), avg_ms=round(avg_ms, 4),
                          ms_per_step=round(total_ms / args.steps, 4))
        if name in flops:
            ktab[name]["tflops"] = round(flops[name] / (avg_ms * 1e-3) / 1e12, 2)
            # the roofline entry is the largest SINGLE kernel launch; "mh_mlp_wgrad[...]" is a group of 6-12 per-layer
            # launches plus a reduction (each <= 0.65 ms) and is listed in "kernels" with its own TFLOP/s
            if dominant is None and not name.startswith("mh_mlp_wgrad"):
                dominant = name

    def pmc_traffic(kernel_symbol):
        """HBM bytes per launch measured by the committed rocprofv3 --pmc passes of this same command
        (profiles/r0N_pmc_summary.csv: FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, KB units)."""
        import csv
        for rnd in ("r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.csv")) as f:
                    for row in csv.DictReader(f):
                        if row["k"] == kernel_symbol:
                            return round((float(row["hbm_read_MB_per_launch"]) + float(row["hbm_write_MB_per_launch"])) * 1024 * 1024)
            except Exception:      # noqa: BLE001
                continue
        return None

    mode = ops._warp_mode()                     # "h2" / "b3" / "" (native fp32 MFMA)
    b3 = mode != ""                             # the warp nets run on the 16-bit matrix pipe with sliced operands
    fsym = {"h2": "warp_fwd_h2_kernel<4>", "b3": "warp_fwd_b3_kernel<8>", "": "warp_fwd_kernel"}[mode]
    bsym = {"h2": "warp_bwd_h2_kernel<8>", "b3": "warp_bwd_b3_kernel<8>", "": "warp_bwd_kernel"}[mode]
    symbol = {"mh_warp_fwd": fsym, "mh_warp_bwd_data": bsym, "mh_field_fwd": "field_fwd_kernel", "mh_field_bwd_data": "field_bwd_kernel"}
    full = render_wl and N * S == 128 * 128 * 128
    roofline = None
    if dominant is not None:
        ach = flops[dominant] / (ktab[dominant]["avg_ms"] * 1e-3) / 1e12
        on_b3 = b3 and dominant.startswith("mh_warp")
        # sliced kernels issue `prod` 16-bit slice products per fp32 MAC on the 2.5 PFLOP/s dense bf16 / fp16 matrix pipe: the
        # yardstick for ALGORITHMIC fp32 FLOP/s on that unit is 2500 / prod (the native fp32 MFMA peak, 157.3, is no longer
        # the ceiling).  b3: three bf16 slices, six products; h2: two fp16 slices, three products.
        prod = {"h2": 3.0, "b3": 6.0}.get(mode, 1.0) if on_b3 else 1.0
        peak = BF16_MFMA_PEAK_TFLOPS / prod if on_b3 else FP32_MFMA_PEAK_TFLOPS
        notes = {"b3": "algorithmic fp32 FLOP/s against the dense bf16 MFMA peak (2500 TFLOP/s, MI355X_MICROARCH.md) / 6 "
                       "slice products per MAC: every fp32 operand is cut exactly into three bf16 slices, the six "
                       "significant cross products go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation",
                 "h2": "algorithmic fp32 FLOP/s against the dense fp16 MFMA peak (2500 TFLOP/s, MI355X_MICROARCH.md) / 3 "
                       "slice products per MAC: every fp32 operand is cut into two fp16 slices at a power-of-two scale (22 "
                       "significand bits), the three significant cross products go through v_mfma_f32_32x32x16_f16 with "
                       "fp32 accumulation; fp32-grade by test_warp_sliced_arithmetic_is_fp32_grade"}
        roofline = dict(kernel=dominant, bound="mfma", achieved=round(ach, 2), peak=round(peak, 1),
                        unit="TFLOP/s", frac=round(ach / peak, 4),
                        peak_note=(notes[mode] if on_b3 else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md"),
                        issued_tflops=round(prod * ach, 1) if on_b3 else round(ach, 2),
                        issued_frac_of_unit_peak=round(prod * ach / BF16_MFMA_PEAK_TFLOPS, 4) if on_b3 else round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        vs_fp32_mfma_peak=round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        traffic=pmc_traffic(symbol.get(dominant, "")) if (full and args.workload == "cfg3") else None,
                        traffic_note="HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), not a live "
                                     "measurement; null when the workload differs from the profiled one",
                        flops_per_launch=flops[dominant], avg_launch_ms=ktab[dominant]["avg_ms"])
        if dominant == "mh_warp_fwd" and ktab[dominant].get("calls_per_step", 1.0) == 1.0:
            # the training forward parks every layer's activations for the weight gradients: 1344 fp32 rows + 40 rows of ReLU sign
            # masks per 32-point tile (csrc/mlp_dev.h: WARP_ACT_ROWS), x in, 5 floats out -- 5568 B per point, written once.
            # With the fp16 x 2 slices the kernel sits closer to THAT roof than to the matrix pipe's; report the nearer one as
            # the bound and keep the other view beside it.
            park_bytes = M * (4.0 * (64 + 2 * 640) + 4.0 * 40 + 12.0 + 20.0)
            gbs = park_bytes / (ktab[dominant]["avg_ms"] * 1e-3) / 1e9
            if gbs / HBM_PEAK_GBS > roofline["frac"]:
                mfma_view = {k: roofline[k] for k in ("bound", "achieved", "peak", "unit", "frac", "peak_note", "issued_tflops",
                                                      "issued_frac_of_unit_peak", "vs_fp32_mfma_peak", "flops_per_launch")}
                roofline = dict(kernel=dominant, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=round(gbs / HBM_PEAK_GBS, 4), traffic=roofline["traffic"], traffic_note=roofline["traffic_note"],
                                bytes_per_launch=round(park_bytes), avg_launch_ms=ktab[dominant]["avg_ms"],
                                note="algorithmic bytes = the parked activation tile (5568 B per point, written once; reads are "
                                     "the 12 B of x and the L2-resident weight slices); a pure streaming WRITE reaches 6.8 TB/s on "
                                     "this box (profiles/r02_micro_hbm_rates.txt), and the kernel's clock sits at 1.6 GHz while it "
                                     "parks against 1.9 GHz when it does not (profiles/r02_phase_trace_warp_fwd.txt)",
                                mfma=mfma_view)
        step_flops = 3.0 * (warp_f * (0 if args.workload == "cfg2" else 1) + field_f)
        if args.workload != "cfg3b":
            roofline["whole_step"] = dict(flops=step_flops, tflops=round(step_flops / (ms_step * 1e-3) / 1e12, 2),
                                          frac=round(step_flops / (ms_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                          frac_note="algorithmic FLOP/s of the whole step over the native fp32 MFMA peak (157.3): "
                                                    "a speed-of-light figure for an all-fp32-MFMA step, kept for continuity with "
                                                    "round 1" + ("; the warp nets now run on the 16-bit matrix pipe" if b3 else ""))
    # the warp weight-gradient group is the step's largest item and a streaming kernel: every parked activation and dPre row is
    # read exactly once (2 x 5.4 KB per point: SURVEY 8d's tile geometry, profiles/r02_pmc_summary.csv confirms the bytes)
    roof_wgrad = None
    if render_wl and args.workload != "cfg2" and "mh_mlp_wgrad[warp]" in ktab:
        wg_bytes = M * 4.0 * (2 * 64 + 2 * 5 * 128 + 2 * (5 * 128 + 32))      # H0 per net + hidden activations + dPre rows
        wg_gbs = wg_bytes / (ktab["mh_mlp_wgrad[warp]"]["avg_ms"] * 1e-3) / 1e9
        roof_wgrad = dict(kernel="mh_mlp_wgrad[warp] (12 launches + reduction)", bound="hbm", achieved=round(wg_gbs, 1),
                          peak=HBM_PEAK_GBS, unit="GB/s", frac=round(wg_gbs / HBM_PEAK_GBS, 4), bytes_per_step=round(wg_bytes),
                          note="algorithmic = measured bytes (each operand row read once); this box's torch kernels sustain 4.0 "
                               "(sum) - 5.3 (copy) - 6.8 (fill) TB/s, profiles/r02_micro_hbm_rates.txt")
    roof_hash = None
    if render_wl and ("mh_grid_encode_fwd" in ktab or "mh_grid_encode_fwd2" in ktab):
        # table-passes per step: 2 tables x M points (+ the 6 FD taps of the SDF table in cfg3b).  The two-table launch
        # (mh_grid_encode_fwd2: sdf + colour encoder at the same points) does two table-passes per point, reading x once.
        enc_points = (2 * M + (6 * M if args.workload == "cfg3b" else 0))
        if "mh_grid_encode_fwd2" in ktab:
            fwd_key = "mh_grid_encode_fwd2"
            pts_per_launch = 2 * M                      # table-passes in that launch
            fwd_bytes = (2 * GRID_FWD_BYTES - 12) / 2.0  # algorithmic bytes per table-pass (x is read once for both)
        else:
            fwd_key = "mh_grid_encode_fwd"
            pts_per_launch = enc_points / ktab[fwd_key]["calls_per_step"]
            fwd_bytes = GRID_FWD_BYTES
        secs = ktab[fwd_key]["avg_ms"] * 1e-3
        gb = fwd_bytes * pts_per_launch / secs / 1e9
        l2 = GRID_GATHERS_PER_POINT * 64 * pts_per_launch / secs / 1e9
        traffic = pmc_traffic("grid_fwd_kernel<true>" if fwd_key.endswith("2") else "grid_fwd_kernel") if (full and args.workload != "cfg3b") else None
        roof_hash = dict(kernel="grid_fwd_kernel", bound="hbm", achieved=round(gb, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=round(gb / HBM_PEAK_GBS, 4), traffic=traffic,
                         bytes_per_launch=round(fwd_bytes * pts_per_launch), launch=fwd_key,
                         hbm_measured_gbs=None if traffic is None else round(traffic / secs / 1e9, 1),
                         gathers_per_s=round(GRID_GATHERS_PER_POINT * pts_per_launch / secs / 1e9, 1),
                         l2_sector_gbs_if_every_gather_missed_l1=round(l2, 1), l2_peak_gbs=L2_PEAK_GBS,
                         note="achieved = ALGORITHMIC bytes (SURVEY 8d: 1164 B/point) / time -- the yardstick north_star asks "
                              "for, NOT an HBM utilisation: both 3.2 MB tables are L2/MALL resident, so the gathers are "
                              "cache-served (hbm_measured_gbs = PMC traffic / time).  What bounds the kernel is the gather rate "
                              "(gathers_per_s, in G 8-byte gathers/s): at one 64-byte L2 sector per gather it would need more "
                              "than the 34.5 TB/s aggregate L2 bandwidth, i.e. part of the gathers is served by the per-CU L1 "
                              "(ray-ordered points share cells); see DESIGN.md section 3")
        bwd_name = "mh_grid_encode_bwd_binned" if "mh_grid_encode_bwd_binned" in ktab else "mh_grid_encode_bwd"
        if bwd_name in ktab:
            gbb = GRID_BWD_BYTES * (enc_points / ktab[bwd_name]["calls_per_step"]) / (ktab[bwd_name]["avg_ms"] * 1e-3) / 1e9
            roof_hash["bwd_kernel"] = bwd_name
            roof_hash["bwd_achieved"] = round(gbb, 1)
            roof_hash["bwd_frac"] = round(gbb / HBM_PEAK_GBS, 4)
            roof_hash["bwd_note"] = ("algorithmic bytes of the reference's formulation (2188 B per point incl. the atomics' "
                                     "read-modify-write); the brick kernel accumulates on-chip, so this rate can exceed the HBM "
                                     "peak -- it is a throughput in the reference's units, not an HBM utilisation")
    bucket = wl["bucket"]
    backend = dist.get_backend() if world > 1 else None
    n_dev = 0 if stub else torch.cuda.device_count()
    headline = args.workload in ("cfg3", "cfg2", "cfg3b")
    out = {
        "metric": "rays/sec (fwd+bwd, 128 samples/ray)" if headline else
                  {"train_real": "rays/sec (real-view training step, ragged occupancy samples)",
                   "density128": "points/sec (forward-only field query)"}[args.workload],
        "value": round(total_rays / elapsed, 1), "unit": "rays/s" if args.workload != "density128" else "points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["desc"], "rays_per_gpu": N, "samples_per_ray": S if headline else round(M / max(N, 1), 1),
                   "sample_points_per_step_per_gpu": round(M),
                   "parallelism": f"dp{world}" + ("" if bucket is None else
                                                  f" (rays/frames sharded, {bucket.nbytes / 1e6:.2f} MB of gradients all-reduced "
                                                  f"per step" + (", hash-table range early on a side stream" if world > 1 and
                                                                 not args.no_overlap else "") + ")"),
                   "world_size": world, "backend": backend, "devices_visible": n_dev,
                   "ranks_share_devices": bool(world > 1 and n_dev < world),
                   "mlp_arithmetic": {"b3": "warp nets: fp32 values, exact three-way bf16 split of both operands, six slice "
                                            "products per MAC on the bf16 matrix pipe, fp32 accumulate (fp32-grade: <= 3 * 2^-24 of a "
                                            "product dropped); field forward: fp16 x 2 slices unless MORPHEUS_FIELD_FWD says otherwise, "
                                            "field backward native fp32 MFMA (MORPHEUS_MLP=b3)",
                                      "h2": "warp nets (forward, backward-data, large-batch weight gradients) and the field forward: fp32 "
                                            "values, two fp16 slices per operand at power-of-two scales (per layer for weights, per point "
                                            "for activations / gradients, per tensor for the weight-gradient operands; 22 significand "
                                            "bits), three slice products per MAC on the fp16 matrix pipe, fp32 accumulate -- fp32-grade: "
                                            "measured error against float64 equal to the fp32-MFMA kernels'; 32-row layers' and small "
                                            "batches' weight gradients: bf16 x 3 slices; field backward: native fp32 MFMA (the default, "
                                            "MORPHEUS_MLP=h2)",
                                      "": "native fp32 MFMA (MORPHEUS_MLP=f32)"}[mode],
                   "weights": "closed-form state b", "loss": float(loss.item()) if hasattr(loss, "item") else float(loss)},
        "roofline": roofline, "roofline_hashgrid": roof_hash, "roofline_weight_gradients": roof_wgrad, "kernels": ktab,
    }
    if "occupied" in wl:
        out["config"]["occupied_fraction"] = round(wl["occupied"], 4)
    if world == 1 and not args.no_cpu_baseline and not stub and headline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_rays, S)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
