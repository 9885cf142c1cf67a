#!/usr/bin/env python
"""(train_real variant: a step starts at march_wave_kernel)  Per-phase summary of ONE training step from a rocprofv3 kernel trace: the hot-path kernels in launch order with the
torch "glue" launches between them counted and named.  python tools/step_timeline.py [gpurun_out/prof[_wl]]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
f = max(glob.glob(os.path.join(d, "runc", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("march_wave_kernel")]
seg = rows[first[-2]:first[-1]]
MINE = ("warp_", "field_", "wgrad", "grid_", "bin_", "absmax", "fx_scales", "composite", "sample_", "adam_", "wn_kernel",
        "march_", "generate_")
t0 = int(seg[0]["Start_Timestamp"])
print(f"{len(seg)} kernels, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms  ({os.path.basename(f)})")
cnt, dur, names, gl_n, gl_t = 0, 0.0, {}, 0, 0.0


def flush():
    global cnt, dur, names
    if cnt:
        print(f"      ... {cnt} glue launches, {dur:.0f} us: " + ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])))
    cnt, dur, names = 0, 0.0, {}


for r in seg:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("at::native::", "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if any(m in name[:40] for m in MINE):
        flush()
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.0f} {us:8.0f} us  {name[:60]}")
    else:
        cnt += 1
        dur += us
        gl_n += 1
        gl_t += us
        k = re.sub(r"<.*", "", name)[:24]
        m = re.search(r"(CUDAFunctor_\w+|\w+Functor|NormTwoOps|MeanOps|\w+_kernel_cuda|func_wrapper)", name)
        k += ":" + m.group(1) if m else ""
        names[k] = names.get(k, 0) + 1
flush()
print(f"glue total: {gl_n} launches, {gl_t / 1e3:.3f} ms busy")
