#!/bin/bash
# usage: bash tools/q3_quick.sh TAG [ENV=VAL ...] -- Qwen3-8B AWQ batch sweep (tools/batch_sweep.py) under the given environment
tag=$1; shift
for e in "$@"; do export "$e"; done
timeout -k 10 250 python3 tools/batch_sweep.py 1,2,3,4 > gpurun_out/q3_$tag.log 2>&1 || exit 1
tail -3 gpurun_out/q3_$tag.log
