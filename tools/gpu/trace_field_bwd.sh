#!/bin/bash
# GPU-box script: phase trace of the fused field backward's sdf kernel, both arithmetic forms, colour + sdf and sdf-only passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export MORPHEUS_HIP_LIB=$PWD/morpheus_amd/_build/libmorpheus_trace.so
for fb in f32 b3; do for wc in 1 0; do
  MORPHEUS_FIELD_BWD=$fb python tools/phase_trace_field_bwd.py $wc 2>&1 | tail -19
done; done
