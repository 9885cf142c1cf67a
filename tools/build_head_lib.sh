#!/bin/bash
# container-side: compile HEAD's csrc (git archive) into morpheus_amd/_build/libmorpheus_head.so for same-box A/Bs (tools/gpu/lib_ab.sh,
# tools/gpu/fbwd_ab.py) against the working tree's library.   tools/build_head_lib.sh [rev=HEAD]
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
T=$(mktemp -d)
git archive $REV morpheus_amd/csrc include | tar -x -C $T
OBJS=""
for f in $T/morpheus_amd/csrc/*.hip; do
  n=$(basename $f .hip); extra=""; [ $n = losses ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $extra -c $f -o $T/$n.o &
  OBJS="$OBJS $T/$n.o"
done
wait
mkdir -p morpheus_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o morpheus_amd/_build/libmorpheus_head.so $OBJS
rm -rf $T
echo built morpheus_amd/_build/libmorpheus_head.so from $REV
