set -o pipefail
timeout -k 10 200 python -u -m pytest tests/test_gpu_true_geometry.py -m gpu -x -q --timeout 120 --timeout-method thread -k "qwen3 or Qwen3 or awq or three" > gpurun_out/t1.log 2>&1; echo rc=$? >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
grep -q "rc=0" gpurun_out/t1.log || exit 1
timeout -k 10 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-secondary --no-moe > gpurun_out/b_fold.log 2>&1 || exit 2
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_fold -o fold -- python3 /root/repo/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-secondary --no-moe > /root/repo/gpurun_out/prof_fold.log 2>&1
