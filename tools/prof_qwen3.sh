# usage: bash tools/prof_qwen3.sh TAG [ENV=VAL ...]  -- rocprofv3 kernel trace of a short Qwen3-8B bench into gpurun_out/prof_TAG
tag=$1; shift
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o $tag -- python3 /root/repo/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-secondary --no-moe > /root/repo/gpurun_out/prof_$tag.log 2>&1
