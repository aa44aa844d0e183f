#!/bin/bash
# usage: bash tools/exl3_quick.sh TAG [ENV=VAL ...] -- Qwen3-8B EXL3 4 bpw decode tok/s and per-kind linear times (bench.py --workload exl3)
tag=$1; shift
for e in "$@"; do export "$e"; done
timeout -k 10 250 python3 bench.py --workload exl3 --steps 64 > gpurun_out/exl3_$tag.log 2>&1 || exit 1
python3 -c "
import json;d=json.loads(open('gpurun_out/exl3_$tag.log').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],{k:v['avg_launch_us'] for k,v in d['roofline']['linears'].items()})"
