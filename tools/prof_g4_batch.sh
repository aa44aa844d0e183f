#!/bin/bash
# usage: bash tools/prof_g4_batch.sh B   -- rocprofv3 kernel trace of the Gemma-4-12B-shaped decode at B concurrent sequences
set -e
b=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g4b && mkdir -p /tmp/prof_g4b
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_g4b -o g4b -- python3 tools/gemma_batch.py $b > gpurun_out/prof_g4b_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_g4b -name "*_results.db" | head -1) > gpurun_out/prof_g4b_kernels.txt 2>&1 || true
head -40 gpurun_out/prof_g4b_kernels.txt
