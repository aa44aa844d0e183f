#!/bin/bash
# usage: bash tools/g4_quick.sh TAG [ENV=VAL ...] -- Gemma-4-12B-shaped decode tok/s (bench.py --workload gemma4, no CPU leg) under the given environment
tag=$1; shift
for e in "$@"; do export "$e"; done
timeout -k 10 200 python3 bench.py --workload gemma4 --steps 64 --no-cpu-baseline > gpurun_out/g4_$tag.log 2>&1 || exit 1
python3 -c "
import json,sys;d=json.loads(open('gpurun_out/g4_$tag.log').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],{k:v['avg_launch_us'] for k,v in d['roofline']['mmvq'].items()})"
