#!/usr/bin/env python
"""One replayed / eager training step from a rocprofv3 kernel trace (a step starts at march_wave_kernel), aggregated by kernel:
launches, busy microseconds, largest single launch.  python tools/step_aggregate.py [gpurun_out/prof_train_real_graph]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_train_real_graph"
f = max(glob.glob(os.path.join(d, "runc", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("march_wave_kernel")]
seg = rows[first[-2]:first[-1]]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in seg:
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("at::native::", "")
    k = re.sub(r"\(.*", "", n)[:60]
    m = re.search(r"(CUDAFunctor_\w+|\w+Functor|NormTwoOps|MeanOps|\w+_kernel_cuda|func_wrapper|CatArray\w+|distribution\w+|multi_tensor\w+)", n)
    if "elementwise" in k or "reduce_kernel" in k or "anonymous" in k or not k.strip():
        k = k[:24] + ":" + (m.group(1) if m else "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[k]
    a[0] += 1
    a[1] += us
    a[2] = max(a[2], us)
tot = sum(v[1] for v in agg.values())
print(f"{len(seg)} kernels, busy {tot:.0f} us, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.0f} us  ({os.path.basename(f)})")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.0f} us {v[0]:4d} x  max {v[2]:6.0f}  {k}")
