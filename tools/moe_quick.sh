#!/bin/bash
# usage: bash tools/moe_quick.sh TAG [ENV=VAL ...] -- Mixtral-8x7B-shaped decode tok/s (bench.py --workload mixtral) under the given environment
tag=$1; shift
for e in "$@"; do export "$e"; done
timeout -k 10 250 python3 bench.py --workload mixtral --steps 64 --no-cpu-baseline > gpurun_out/moe_$tag.log 2>&1 || exit 1
python3 -c "
import json;d=json.loads(open('gpurun_out/moe_$tag.log').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'])"
