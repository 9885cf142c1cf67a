import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from morpheus_amd import chunking
dev = torch.device("cuda", 0)
args = bench.parse_args(["--workload", "train_virtual", "--virtual-res", "180", "--steps", "8", "--no-cpu-baseline", "--no-kernel-timers"])
wl = bench.build_train_virtual(args, 0, 1, dev)
step = wl["step"]
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")
for i in range(14):
    m0 = torch.cuda.memory_stats(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    m1 = torch.cuda.memory_stats(dev)
    print(i, f"host {1e3*(t1-t0):.1f} ms  total {1e3*(t2-t0):.1f} ms", {k: m1.get(k, 0) - m0.get(k, 0) for k in keys},
          f"reserved {m1['reserved_bytes.all.current']/1e9:.1f} GB", dict(chunking.STATS), flush=True)
