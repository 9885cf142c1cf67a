#!/bin/bash
# GPU-box script: rebuild csrc/mlp_h2.hip with each extra flag set in "$@" (quoted, one per arm; "" = the shipped build) and time
# the h2 kernels (tools/gpu/h2_check.py: accuracy lines + forward / backward ms at 2 M points)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for flags in "$@"; do
  echo "=== flags: '$flags'"
  touch morpheus_amd/csrc/mlp_h2.hip
  MH_EXTRA_FLAGS="$flags" python -m morpheus_amd.build > /dev/null 2>&1 || { echo build failed; continue; }
  timeout 250 python tools/gpu/h2_check.py 2>&1 | grep -E "^h2" | tail -3
done
touch morpheus_amd/csrc/mlp_h2.hip; python -m morpheus_amd.build > /dev/null 2>&1
