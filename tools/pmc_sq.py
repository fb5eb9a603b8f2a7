#!/usr/bin/env python
"""Per-kernel SQ counter table from rocprofv3 --pmc passes (rocpd sqlite).  usage: tools/pmc_sq.py out.md pass1.db [pass2.db ...]"""
import re, sqlite3, sys
import os
FILTER = os.environ.get("PMC_FILTER", "gemm")
tab = {}
for path in sys.argv[2:]:
    cur = sqlite3.connect(path).cursor()
    for name, ctr, n, val, ns in cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(end-start) from counters_collection group by kernel_name, counter_name"):
        k = re.sub(r"\(.*", "", name)
        tab.setdefault(k, {})[ctr] = val / n
        tab[k]["_us"] = ns / n / 1e3
ctrs = sorted({c for v in tab.values() for c in v if not c.startswith("_")})
lines = ["| kernel | avg us | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["_us"]):
    if FILTER and FILTER not in k: continue
    lines.append(f"| `{k[:70]}` | {v['_us']:.1f} | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in ctrs) + " |")
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
