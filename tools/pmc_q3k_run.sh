#!/bin/bash
# HBM traffic and issue counters of the two Q3_K lm_head kernels (tools/pmc_q3k.py): separate rocprofv3 --pmc passes, per-kernel
# averages by tools/pmc_csv_summary.py into gpurun_out/pmc_q3k_*.txt
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_q3k_$i
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_q3k_$i -o p -- python3 tools/pmc_q3k.py > gpurun_out/pmc_q3k_$i.log 2>&1
  python3 tools/pmc_csv_summary.py $(find /tmp/pmc_q3k_$i -name "*counter_collection.csv" | head -1) q3k > gpurun_out/pmc_q3k_$i.txt
done
cat gpurun_out/pmc_q3k_1.txt gpurun_out/pmc_q3k_2.txt gpurun_out/pmc_q3k_3.txt
