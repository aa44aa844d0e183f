# usage: bash tools/ab_env.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- one short Qwen3-8B bench per environment setting
set -o pipefail
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-secondary --no-moe > gpurun_out/ab_$i.log 2>&1 || { echo "FAILED: $e"; tail -3 gpurun_out/ab_$i.log; exit 1; }
  python - "$e" gpurun_out/ab_$i.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[1]:50s} tok/s {d['value']:7.1f}  ms {d['ms_per_step']:.4f}  linears avg {r['avg_launch_us']:.3f} us frac {r['frac']:.3f}", flush=True)
PY
done
