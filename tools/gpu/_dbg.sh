O=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
MORPHEUS_MAX_PARK_GB=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$i -- python $GRAFT_REPO_ROOT/bench.py --workload train_virtual --virtual-res 180 --steps 8 --no-cpu-baseline --no-kernel-timers --detail-out /tmp/dbg.json 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('run $i', d['ms_per_step'])"
f=$(find $O/prof$i -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
done
