#!/bin/bash
# k_intersect time on 4 M incoherent rays per scene (3 launches each): rocprofv3 kernel trace of tools/gpu_trace_rate.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trace
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- python $R/tools/gpu_trace_rate.py > /tmp/prof_trace.log 2>&1
grep -E "^cornell|^atrium" /tmp/prof_trace.log
python $R/tools/rocpd_summary.py stats $(find /tmp/prof_trace -name "*.db" | head -1) | head -6
