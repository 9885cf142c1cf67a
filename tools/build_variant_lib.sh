#!/bin/bash
# container-side: compile the WORKING TREE's csrc with extra -D flags into morpheus_amd/_build/libmorpheus_<name>.so for same-box A/Bs
# (loaded through MORPHEUS_HIP_LIB).   tools/build_variant_lib.sh <name> "-DFLAG ..."
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
T=$(mktemp -d)
OBJS=""
for f in morpheus_amd/csrc/*.hip; do
  n=$(basename $f .hip); extra=""; [ $n = losses ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $extra $FLAGS -c $f -o $T/$n.o &
  OBJS="$OBJS $T/$n.o"
done
wait
mkdir -p morpheus_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o morpheus_amd/_build/libmorpheus_$NAME.so $OBJS
rm -rf $T
echo built morpheus_amd/_build/libmorpheus_$NAME.so with $FLAGS
