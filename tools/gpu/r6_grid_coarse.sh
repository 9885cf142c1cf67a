#!/bin/bash
# GPU-box (round 6): the UPPER BOUND of keeping the coarse levels' corner sums in registers -- variants of the
# brick backward whose level lanes < K never issue their LDS atomics (wrong results on purpose: tools/micro/hashgrid_brk_exp.patch,
# built by tools/build_grid_variants.sh), against the shipped kernel and the no-atomics form, tools/bench_grid.py, one box, 2 repetitions
O=gpurun_out/r6coarse; mkdir -p $O; : > $O/summary.txt
for rep in 1 2; do
for so in morpheus_amd/_build/ab_*.so; do
  n=$(basename $so .so)
  MORPHEUS_HIP_LIB=$PWD/$so timeout 300 python tools/bench_grid.py > $O/${n}_$rep.txt 2>&1
  echo "== $n rep $rep" | tee -a $O/summary.txt
  grep "binned n_levels=16\|^\[" $O/${n}_$rep.txt | tee -a $O/summary.txt
done
done
