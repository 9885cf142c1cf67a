#!/bin/bash
# container-side: build A/B variants of the library that differ in hashgrid.hip only (other objects from the in-tree build).
#   tools/build_grid_variants.sh name1:"-DFLAG=1 ..." name2:"..."   ->  morpheus_amd/_build/ab_<name>.so
# `head` as flags compiles HEAD's hashgrid.hip (git show) instead of the working tree's.
# Flags containing -DBRK_EXP_* (timing experiments that give WRONG results on purpose: NOATOM, SAMEROW, NOFLUSH) compile a temporary
# copy of hashgrid.hip with tools/micro/hashgrid_brk_exp.patch applied: the experiment code never sits in the product source, and
# morpheus_amd/build.py refuses the macros.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
B=morpheus_amd/_build
OBJS=$(ls $B/*.o | grep -v "/hashgrid.o\|ab_")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  src=morpheus_amd/csrc/hashgrid.hip
  if [ "$flags" = head ]; then git show HEAD:$src > morpheus_amd/csrc/_ab_head_hashgrid.hip; src=morpheus_amd/csrc/_ab_head_hashgrid.hip; flags=""; fi
  case "$flags" in *BRK_EXP*)
    cp $src morpheus_amd/csrc/_ab_exp_hashgrid.hip
    sed 's#morpheus_amd/csrc/hashgrid.hip#morpheus_amd/csrc/_ab_exp_hashgrid.hip#g' tools/micro/hashgrid_brk_exp.patch | patch -p1 -s
    src=morpheus_amd/csrc/_ab_exp_hashgrid.hip;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude $flags -c $src -o $B/ab_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/ab_$name.so $OBJS $B/ab_$name.o
  rm -f morpheus_amd/csrc/_ab_head_hashgrid.hip morpheus_amd/csrc/_ab_exp_hashgrid.hip $B/ab_$name.o
  echo built $B/ab_$name.so
done
