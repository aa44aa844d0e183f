#!/bin/bash
# usage: bash tools/prof_cmd.sh TAG python3 script.py args...   -- rocprofv3 kernel trace of one command; per-kernel summary to gpurun_out/prof_TAG_kernels.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- "$@" > gpurun_out/prof_${tag}_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_$tag -name "*_results.db" | head -1) > gpurun_out/prof_${tag}_kernels.txt 2>&1 || true
head -${PROF_HEAD:-14} gpurun_out/prof_${tag}_kernels.txt | cut -c1-170
