#!/bin/bash
# container-side: build A/B variants of the library that differ in hashgrid.hip only (other objects from the in-tree build).
#   tools/build_grid_variants.sh name1:"-DFLAG=1 ..." name2:"..."   ->  morpheus_amd/_build/ab_<name>.so
# `head` as flags compiles HEAD's hashgrid.hip (git show) instead of the working tree's.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
B=morpheus_amd/_build
OBJS=$(ls $B/*.o | grep -v "/hashgrid.o\|ab_")
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  src=morpheus_amd/csrc/hashgrid.hip
  if [ "$flags" = head ]; then git show HEAD:$src > morpheus_amd/csrc/_ab_head_hashgrid.hip; src=morpheus_amd/csrc/_ab_head_hashgrid.hip; flags=""; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude $flags -c $src -o $B/ab_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/ab_$name.so $OBJS $B/ab_$name.o
  rm -f morpheus_amd/csrc/_ab_head_hashgrid.hip $B/ab_$name.o
  echo built $B/ab_$name.so
done
