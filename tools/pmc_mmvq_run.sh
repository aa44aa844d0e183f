#!/bin/bash
# HBM traffic of the Gemma-4-12B-shaped MMVQ launches: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# tools/pmc_mmvq.py, per-kernel averages by tools/pmc_csv_summary.py into gpurun_out/pmc_mmvq_{fetch,write}.txt
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 tools/pmc_mmvq.py > gpurun_out/pmc_mmvq_$c.log 2>&1
  python3 tools/pmc_csv_summary.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) iq_ > gpurun_out/pmc_mmvq_$c.txt
done
cat gpurun_out/pmc_mmvq_FETCH_SIZE.txt gpurun_out/pmc_mmvq_WRITE_SIZE.txt
