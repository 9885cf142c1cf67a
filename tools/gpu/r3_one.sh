#!/bin/bash
# GPU-box script: run a pytest selection ($1 = -k expression) and print the tail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "$1" 2>&1 | grep -v "Warning\|warn\|Consider\|run_backward" | tail -60 > gpurun_out/one_tests.log
tail -40 gpurun_out/one_tests.log
