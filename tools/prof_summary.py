#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / avg / min / max duration.
Kernel names are shortened to the function name; torch's own kernels are dropped unless --all."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)(<[^>]*>)?\(", name)
    base = m.group(1) + (m.group(2) or "") if m else name[:60]
    return base


def main():
    path = sys.argv[1]
    show_all = "--all" in sys.argv
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            if not show_all and ("at::native" in name or "hiprand" in name):
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
            wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
            rows[(short(name), grid, wg, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))].append(dur)
    print(f"{'kernel':48s} {'grid':>9s} {'wg':>5s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print(f"{key[0][:48]:48s} {key[1]:>9s} {key[2]:>5s} {key[3]:>5s} {key[4]:>6s} {len(d):6d} {sum(d)/len(d)/1e3:9.2f} {min(d)/1e3:9.2f} {max(d)/1e3:9.2f}")


if __name__ == "__main__":
    main()
