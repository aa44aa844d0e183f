#!/bin/bash
# rocprofv3 kernel trace of the Gemma-4-12B-shaped batch-1 decode step alone -> gpurun_out/prof_g4dec_kernels.txt
set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g4dec && mkdir -p /tmp/prof_g4dec
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g4dec -o g4dec -- python3 tools/g4_decode_only.py ${1:-64} > gpurun_out/prof_g4dec_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_g4dec -name "*_results.db" | head -1) > gpurun_out/prof_g4dec_kernels.txt 2>&1 || true
tail -n 2 gpurun_out/prof_g4dec_run.log
head -30 gpurun_out/prof_g4dec_kernels.txt | cut -c1-160
