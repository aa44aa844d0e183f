#!/bin/bash
# usage: bash tools/prof_ctx.sh CONTEXTS TAG -- rocprofv3 kernel trace of the long-context decode sweep (tools/ctx_sweep.py)
set -e
ctx=$1; tag=${2:-ctx}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python3 tools/ctx_sweep.py $ctx 16 > gpurun_out/prof_${tag}_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_$tag -name "*_results.db" | head -1) > gpurun_out/prof_${tag}_kernels.txt 2>&1 || true
grep "by_context" gpurun_out/prof_${tag}_run.log | cut -c1-300
head -14 gpurun_out/prof_${tag}_kernels.txt
