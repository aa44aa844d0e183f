#!/usr/bin/env python3
"""Summarise the kernel dispatches of a rocprofv3 results database (rocpd sqlite): per (kernel, grid) calls / avg us / share.
usage: prof_db_summary.py results.db [--all]"""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = defaultdict(list)
for name, s, e, gx, wx in cur.execute(f"select s.kernel_name,k.start,k.end,k.grid_size_x,k.workgroup_size_x from {kd} k join {ks} s on k.kernel_id=s.id"):
    if "--all" not in sys.argv and ("at::native" in name or "hiprand" in name or "rocclr" in name):
        continue
    m = re.search(r"\d+([a-z_0-9]+_kernel)(I[A-Za-z0-9]+E)?", name)
    rows[((m.group(1) + (m.group(2) or "")) if m else name[:48], gx, wx)].append(e - s)
tot = sum(sum(v) for v in rows.values())
print(f"{'kernel':44s} {'grid':>9s} {'wg':>4s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'share':>6s}")
for k, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0][:44]:44s} {k[1]:>9} {k[2]:>4} {len(d):6d} {sum(d)/len(d)/1e3:8.2f} {min(d)/1e3:8.2f} {sum(d)/tot*100:5.1f}%")
