#!/bin/bash
# usage: bash tools/pmc_prefill.sh P  -- MFMA-busy counters of the prompt path at a P-token prompt (tools/prefill_long.py), one --pmc pass,
# plus a kernel trace of the same program for the durations; per-kernel averages into gpurun_out/pmc_prefill_P.txt
set -e
P=${1:-2048}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_pf$P /tmp/prof_pf$P
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_pf$P -o p -- python3 tools/prefill_long.py $P > gpurun_out/pmc_prefill_${P}_run.log 2>&1
python3 tools/pmc_csv_summary.py $(find /tmp/pmc_pf$P -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_prefill_$P.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_pf$P -o pf -- python3 tools/prefill_long.py $P > gpurun_out/prof_pf${P}_run.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_pf$P -name "*_results.db" | head -1) > gpurun_out/prof_pf${P}_kernels.txt 2>&1 || true
grep -A6 "prefill_attn\|int4_gemm_tiled" gpurun_out/pmc_prefill_$P.txt | head -60
grep "prefill_attn\|int4_gemm_tiled" gpurun_out/prof_pf${P}_kernels.txt
