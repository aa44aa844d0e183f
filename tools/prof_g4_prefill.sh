#!/bin/bash
# rocprofv3 kernel summary of the Gemma prefill probe; the database stays in /tmp on the box, only the summary comes back
set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/g4p && mkdir -p /tmp/g4p
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/g4p -o g4p -- python3 tools/g4_prefill_time.py 256 8 > gpurun_out/g4p_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/g4p -name "*_results.db" | head -1) > gpurun_out/g4p_kernels.txt 2>&1 || true
head -30 gpurun_out/g4p_kernels.txt
