#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python tools/phase_trace_b3.py 2>&1 | tail -14
MH_TRACE_NOPARK=1 python tools/phase_trace_b3.py 2>&1 | tail -12
