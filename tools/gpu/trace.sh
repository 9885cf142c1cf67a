#!/bin/bash
# GPU-box script: build the phase-trace variant of the library and print where a warp_fwd_kernel wave spends its time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DMH_PHASE_TRACE -o morpheus_amd/_build/libmorpheus_trace.so morpheus_amd/csrc/*.hip 2>&1 | grep error
echo '--- two workgroups per CU (product configuration)'
python tools/phase_trace.py 2>&1 | grep -v amdgpu
echo '--- one workgroup per CU'
MH_TRACE_DYNLDS=40000 python tools/phase_trace.py 2>&1 | grep -v amdgpu
echo '--- two workgroups per CU, no activation parking (inference form)'
MH_TRACE_NOPARK=1 python tools/phase_trace.py 2>&1 | grep -v amdgpu
echo '--- one workgroup per CU, no activation parking'
MH_TRACE_NOPARK=1 MH_TRACE_DYNLDS=40000 python tools/phase_trace.py 2>&1 | grep -v amdgpu
