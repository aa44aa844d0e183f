"""Prints the headline numbers of a bench.py JSON line (file argument): value, per-kind linear timings, batch sweep."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def walk(d, path=""):
    if not isinstance(d, dict):
        return
    if "value" in d and "unit" in d:
        print(f"{path or 'primary'}: {d['value']} {d['unit']}  ms/step {d.get('ms_per_step')}")
    for k in ("linears", "mmvq"):
        if k in (d.get("roofline") or {}):
            for n, v in d["roofline"][k].items():
                print(f"   {n:10s} {v['avg_launch_us']:7.2f} us  frac {v['frac']:.3f}")
    for k in ("batch_sweep",):
        if k in d:
            b = d[k].get("by_batch", d[k])
            print("   batch:", {kk: vv.get("tok_s") for kk, vv in b.items() if isinstance(vv, dict)})
    if "context_sweep" in d:
        print("   ctx:", {kk: vv.get("tok_s") for kk, vv in d["context_sweep"]["by_context"].items()})
    if "prefill" in d:
        print("   prefill ms", d["prefill"].get("ms"))
    for k, v in d.items():
        if k.startswith("secondary") and isinstance(v, dict):
            walk(v, k)
walk(d)
