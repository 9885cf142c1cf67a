#!/bin/bash
# round 6: whole GPU suite + smoke + the default bench (compact line + detail)
O=gpurun_out/r6f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1200 python bench.py --detail-out $O/bench_detail.json ) > $O/bench.log 2> $O/bench.err
tail -c 2300 $O/bench.log; tail -4 $O/bench.err
