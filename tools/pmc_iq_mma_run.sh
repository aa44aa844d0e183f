#!/bin/bash
# SQ issue counters and HBM reads of the matrix-core MMVQ launches of tools/pmc_iq_mma.py: two rocprofv3 --pmc passes, per-kernel
# averages by tools/pmc_csv_summary.py into gpurun_out/pmc_iq_mma_*.txt
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_iqm_$i
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_iqm_$i -o p -- python3 tools/pmc_iq_mma.py > gpurun_out/pmc_iq_mma_$i.log 2>&1
  python3 tools/pmc_csv_summary.py $(find /tmp/pmc_iqm_$i -name "*counter_collection.csv" | head -1) iq_mma_kernel > gpurun_out/pmc_iq_mma_$i.txt
done
cat gpurun_out/pmc_iq_mma_1.txt gpurun_out/pmc_iq_mma_2.txt
