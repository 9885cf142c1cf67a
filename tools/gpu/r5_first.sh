#!/bin/bash
# round 5, first contact: the corrected MFMA/VALU micro, the GPU suite, and the default bench line (compact + detail)
O=gpurun_out/r5a; mkdir -p $O
timeout 200 tools/micro/mfma_valu_gap > $O/micro_mfma_valu_gap.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
( time timeout 1200 python bench.py --detail-out $O/bench_detail.json ) > $O/bench.log 2> $O/bench.err
tail -c 2500 $O/bench.log; tail -5 $O/bench.err
cat $O/micro_mfma_valu_gap.txt
