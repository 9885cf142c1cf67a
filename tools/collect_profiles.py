#!/usr/bin/env python
"""Copy the judged summaries of the latest GPU run from gpurun_out/ (scratch) into profiles/ (tracked).

  python tools/collect_profiles.py [round-tag, default r01]

Inputs (written by tools/gpu/r5_round.sh on the GPU box):
  gpurun_out/bench.log, bench_cfg2.log, bench_cfg3b.log   -> profiles/<tag>_bench_<workload>.json (bench.py's LAST stdout line: the
                                                             compact object the driver parses)
  gpurun_out/bench_detail_<workload>.json                  -> profiles/<tag>_bench_<workload>_detail.json (the full object: kernel
                                                             tables, per-mode lines, notes, allocator statistics)
  gpurun_out/prof/runc/*_kernel_stats.csv                  -> profiles/<tag>_bench_cfg3_kernel_stats.csv
  gpurun_out/prof_cfg3b/runc/*_kernel_stats.csv            -> profiles/<tag>_bench_cfg3b_kernel_stats.csv
  gpurun_out/parity.log                                    -> profiles/<tag>_parity_report.jsonl
  gpurun_out/pmc_{sq,fetch,write}/runc/*_counter_collection.csv -> profiles/<tag>_pmc_summary.csv
PMC units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are KB; FETCH_SIZE
is doubled on gfx950; every counter comes from its own rocprofv3 pass with --kernel-trace only.
"""
import csv
import glob
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def run_start():
    """gpurun_out/ is scratch that survives rounds: only what THIS collection run wrote may be copied under this round's tag (a stale
    file copied under a new name is a mislabelled profile).  The round script writes gpu_tests_full.log first."""
    ref = os.path.join(OUT, "gpu_tests_full.log")
    return os.path.getmtime(ref) - 300.0 if os.path.exists(ref) else 0.0


def fresh(path):
    return bool(path) and os.path.exists(path) and os.path.getmtime(path) >= run_start()


def latest(pattern):
    files = [f for f in glob.glob(os.path.join(OUT, pattern)) if fresh(f)]
    return max(files, key=os.path.getmtime) if files else None


def json_line(path):
    """the LAST line of the log that is a JSON object (bench.py prints the driver's compact line last)"""
    if not fresh(path):
        return None
    found = None
    for line in open(path):
        if line.startswith("{"):
            found = line
    return found


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def pmc_summary(tag, suffix="", steps_in_run=3):
    """suffix: "" for the default (b3) passes in gpurun_out/pmc_{sq,fetch,write}, "_f32" for the per-mode passes in
    gpurun_out/pmc_{sq,fetch,write}_<mode>.  steps_in_run: warm-up + timed steps of the profiled bench command (its
    launches / steps_in_run = launches per step)."""
    per = defaultdict(lambda: defaultdict(list))
    for d in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds", "pmc_lds2"):
        f = latest(f"{d}{suffix}/*/*_counter_collection.csv")
        if not f:
            if d.startswith("pmc_lds"):      # the LDS passes are collected for the default mode only (and a pass with a counter the
                continue                     # box does not know fails alone)
            return False
        for row in csv.DictReader(open(f)):
            per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    mean = lambda v: sum(v) / len(v) if v else 0.0
    cols = ["k", "hbm_read_MB_per_launch", "hbm_write_MB_per_launch", "mfma_busy_frac", "l2_hit_rate", "SQ_WAVE_CYCLES",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "launches_per_step"]
    lds_cols = [c for c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT",
                            "SQ_LDS_ATOMIC_RETURN", "SQ_INSTS_VALU") if any(c in v for v in per.values())]
    cols += lds_cols
    with open(os.path.join(PROF, f"{tag}_pmc_summary{suffix}.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(cols)
        for k in sorted(per):
            c = per[k]
            if not k or k.startswith(("at::", "__amd", "Cijk_")) or "rocprim" in k or "anonymous" in k:
                continue
            hit, miss = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
            # GRBM_GUI_ACTIVE sums the 8 XCDs' clocks, SQ_VALU_MFMA_BUSY_CYCLES the 1024 SIMDs': busy fraction = busy / (gui * 128)
            busy, gui = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), 128.0 * mean(c["GRBM_GUI_ACTIVE"])
            w.writerow([k, round(2.0 * mean(c["FETCH_SIZE"]) / 1024.0, 4), round(mean(c["WRITE_SIZE"]) / 1024.0, 4),
                        round(busy / gui, 4) if gui else 0.0, round(hit / (hit + miss), 4) if hit + miss else 0.0,
                        round(mean(c["SQ_WAVE_CYCLES"]), 4), round(mean(c["SQ_WAIT_ANY"]), 4),
                        round(mean(c["SQ_WAIT_INST_ANY"]), 4), round(mean(c["GRBM_GUI_ACTIVE"]), 4),
                        round(len(c["FETCH_SIZE"]) / float(steps_in_run), 3)] + [round(mean(c[k]), 1) for k in lds_cols])
    return True


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    os.makedirs(PROF, exist_ok=True)
    for log, wl in (("bench.log", "cfg3"), ("bench_cfg2.log", "cfg2"), ("bench_cfg3b.log", "cfg3b"),
                    ("bench_train_real.log", "train_real"), ("bench_train_real_graph.log", "train_real_hip_graph"),
                    ("bench_train_virtual.log", "train_virtual_72"), ("bench_train_virtual_180.log", "train_virtual_180"),
                    ("bench_density128.log", "density128"), ("bench_n2.log", "n2_one_gpu_gloo"),
                    ("bench_cfg3_f32.log", "cfg3_fp32_mfma_kernels"), ("bench_cfg3_b3.log", "cfg3_bf16x3_kernels")):
        line = json_line(os.path.join(OUT, log))
        if line:
            open(os.path.join(PROF, f"{tag}_bench_{wl}.json"), "w").write(line)
            print("bench", wl, len(line), "bytes")
        det = os.path.join(OUT, "bench_detail_" + {"bench.log": "cfg3", "bench_n2.log": "n2", "bench_train_real_graph.log": "train_real_graph",
                                                   "bench_train_virtual.log": "train_virtual"}.get(log, log[len("bench_"):-len(".log")]) + ".json")
        if line and fresh(det):
            shutil.copy(det, os.path.join(PROF, f"{tag}_bench_{wl}_detail.json"))
    for d, wl in (("prof", "cfg3"), ("prof_f32", "cfg3_f32"), ("prof_train_real", "train_real")):
        f = latest(f"{d}/*/*_kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(PROF, f"{tag}_bench_{wl}_kernel_stats.csv"))
            print("kernel stats", wl, os.path.basename(f))
    p = os.path.join(OUT, "parity.log")
    if fresh(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(PROF, f"{tag}_parity_report.jsonl"), "w").writelines(lines)
            print("parity", len(lines), "records")
    for log, name in (("phase_trace_field_bwd.log", "phase_trace_field_bwd.txt"),
                      ("census_fused.log", "launch_census_train_real.txt"), ("census_ref.log", "launch_census_train_real_reference_glue.txt"),
                      ("census_cfg3.log", "launch_census_cfg3.txt"),
                      ("hbm_rates.log", "micro_hbm_rates.txt"), ("mfma_power.log", "micro_mfma_power.txt"),
                      ("mfma_bf16_rate.log", "micro_mfma_bf16_rate.txt"), ("hbm_read.log", "micro_hbm_read.txt"),
                      ("mfma_valu_gap.log", "micro_mfma_valu_gap.txt"), ("phase_trace_b3.log", "phase_trace_warp_fwd_pair.txt"),
                      ("parity_f64.jsonl", "parity_f64.jsonl"),
                      ("bench_grid.log", "micro_hashgrid.txt"), ("gpu_tests.log", "gpu_tests.txt"),
                      ("precision_report.jsonl", "precision_report.jsonl"),
                      ("timeline_train_real_graph.txt", "timeline_train_real_hip_graph.txt"),
                      ("graph_memset_probe.txt", "graph_memset_probe.txt"), ("glue_ab.log", "ab_step_cache.txt"),
                      ("host_profile.log", "host_profile_train_real.txt")):
        src = os.path.join(OUT, log)
        if fresh(src):
            shutil.copy(src, os.path.join(PROF, f"{tag}_{name}"))
            print("copied", name)
    print("pmc summary", pmc_summary(tag), [pmc_summary(tag, "_" + m) for m in ("f32",)])
    parked = hbm_bytes_of("_parked")
    shipped = hbm_bytes_of("")
    if parked and shipped:        # the step's HBM bytes with dPre4 parked (MORPHEUS_REGEN_DPRE4=0) against the shipped form, same box
        with open(os.path.join(PROF, f"{tag}_pmc_hbm_bytes_per_step.txt"), "w") as fo:
            fo.write("# HBM bytes per cfg3 step (b3; rocprofv3 --pmc FETCH_SIZE x 2 (gfx950) / WRITE_SIZE, KB units, own passes; 3 steps per run)\n")
            fo.write("# kernel                                  shipped: read MB  write MB      dPre4 parked (MORPHEUS_REGEN_DPRE4=0): read MB  write MB\n")
            for k in sorted(set(parked) | set(shipped), key=lambda k: -(sum(shipped.get(k, (0, 0))) + sum(parked.get(k, (0, 0))))):
                a, b = shipped.get(k, (0.0, 0.0)), parked.get(k, (0.0, 0.0))
                if max(a + b) < 5.0:
                    continue
                fo.write(f"{k[:40]:40s} {a[0]:16.1f} {a[1]:9.1f} {b[0]:46.1f} {b[1]:9.1f}\n")
            ts, tp = [sum(v[i] for v in shipped.values()) for i in (0, 1)], [sum(v[i] for v in parked.values()) for i in (0, 1)]
            fo.write(f"{'TOTAL (every kernel of the step)':40s} {ts[0]:16.1f} {ts[1]:9.1f} {tp[0]:46.1f} {tp[1]:9.1f}\n")
            fo.write(f"# per step: shipped {sum(ts) / 1024:.2f} GB, dPre4 parked {sum(tp) / 1024:.2f} GB\n")
        print("hbm bytes per step: shipped %.2f GB, parked %.2f GB" % (sum(ts) / 1024, sum(tp) / 1024))


def hbm_bytes_of(suffix, steps_in_run=3):
    """kernel -> (read MB, write MB) per STEP from gpurun_out/pmc_fetch<suffix>, pmc_write<suffix>"""
    out = defaultdict(lambda: [0.0, 0.0])
    for d, col, idx, mul in (("pmc_fetch", "FETCH_SIZE", 0, 2.0), ("pmc_write", "WRITE_SIZE", 1, 1.0)):
        f = latest(f"{d}{suffix}/*/*_counter_collection.csv")
        if not f:
            return None
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == col:
                out[short(row["Kernel_Name"])][idx] += mul * float(row["Counter_Value"]) / 1024.0 / steps_in_run
    return {k: tuple(v) for k, v in out.items()}


if __name__ == "__main__":
    main()
