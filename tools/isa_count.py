#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S device listing: python3 tools/isa_count.py file.s substring [topN]."""
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for i, l in enumerate(lines):
    if l.split(";")[0].strip().endswith(":") and pat in l and not l.startswith((".", "\t")):
        end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm"))
        c = Counter()
        for b in lines[i:end + 1]:
            b = b.strip()
            if not b or b[0] in ";." or b.endswith(":"):
                continue
            c[b.split()[0]] += 1
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(l, "total", sum(c.values()), "valu", valu, "ds", sum(v for k, v in c.items() if k.startswith("ds_")),
              "vmem", sum(v for k, v in c.items() if k.startswith("global_") or k.startswith("buffer_")))
        print("  " + ", ".join(f"{k} {v}" for k, v in c.most_common(top)))
for l in lines:
    if pat in l and (".vgpr_count" in l or ".name:" in l):
        pass
