#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite output).
usage: tools/pmc_traffic.py fetch.db write.db [out.md]
Values are the tool's kilobyte counters per dispatch, averaged per kernel name.  gfx950 correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts wide coalesced streaming reads at exactly half their bytes ->
doubled here; WRITE_SIZE is calibrated on kernels of this run with a known byte count (cast_kernel, adamw_kernel)."""
import re
import sqlite3
import sys


def load(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value), sum(end-start) from counters_collection where counter_name=? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n if len(n) < 80 else n[:77] + "..."


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
lines = ["| kernel | launches | FETCH_SIZE x2 MB/launch | WRITE_SIZE MB/launch | avg us (PMC run) |", "|---|---|---|---|---|"]
tot = {}
for k in sorted(f, key=lambda k: -(f[k][1] * 2 + w.get(k, (0, 0, 0))[1])):
    n, kb, ns = f[k]
    wn, wkb, _ = w.get(k, (n, 0.0, 0))
    tot[k] = (n, 2 * kb / n / 1024, wkb / max(wn, 1) / 1024)
    lines.append(f"| `{short(k)}` | {n} | {2 * kb / n / 1024:.2f} | {wkb / max(wn, 1) / 1024:.2f} | {ns / n / 1e3:.1f} |")
txt = "\n".join(lines[:42])
print(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt + "\n")
