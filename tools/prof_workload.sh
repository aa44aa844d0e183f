#!/bin/bash
# usage: bash tools/prof_workload.sh WORKLOAD [STEPS]   (qwen3 | gemma4 | mixtral | exl3)
# rocprofv3 kernel trace of ONE workload of bench.py; the database stays in /tmp on the box, the per-kernel summary
# (tools/rocpd_summary.py) and the run's JSON line come back under gpurun_out/.
set -e
w=$1; steps=${2:-32}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$w && mkdir -p /tmp/prof_$w
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python3 bench.py --workload $w --steps $steps --warmup 4 --no-cpu-baseline > gpurun_out/prof_${w}_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_$w -name "*_results.db" | head -1) > gpurun_out/prof_${w}_kernels.txt 2>&1 || true
head -24 gpurun_out/prof_${w}_kernels.txt
