#!/bin/bash
# HBM traffic of the Qwen3-8B-shaped int4 linear launches at ROWS activation rows: two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE) over tools/pmc_linears.py, per-kernel averages by tools/pmc_csv_summary.py into gpurun_out/pmc_lin<ROWS>_{FETCH,WRITE}_SIZE.txt
set -e
rows=${1:-1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcl_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcl_$c -o p -- python3 tools/pmc_linears.py $rows > gpurun_out/pmc_lin${rows}_$c.log 2>&1
  python3 tools/pmc_csv_summary.py $(find /tmp/pmcl_$c -name "*counter_collection.csv" | head -1) int4_ > gpurun_out/pmc_lin${rows}_$c.txt
done
cat gpurun_out/pmc_lin${rows}_FETCH_SIZE.txt gpurun_out/pmc_lin${rows}_WRITE_SIZE.txt
