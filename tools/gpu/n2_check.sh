#!/bin/bash
# GPU-box script: exercise the N=2 bench path on ONE GPU (gloo for the exchange) and the --graph option
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
MORPHEUS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-kernel-timers > gpurun_out/bench_n2.log 2>&1
tail -1 gpurun_out/bench_n2.log | cut -c1-330
timeout 600 python bench.py --steps 5 --warmup 2 --graph --no-cpu-baseline --no-kernel-timers > gpurun_out/bench_graph.log 2>&1
tail -1 gpurun_out/bench_graph.log | cut -c1-200
