#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / share, like --stats CSV.
usage: tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n if len(n) < 90 else n[:87] + "..."


lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, mn, mx in rows[:40]:
    lines.append(f"| `{short(n)}` | {c} | {t/1e6:.2f} | {t/c/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/tot:.1f} |")
lines.append(f"\ntotal kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
