#!/usr/bin/env python
"""Two-stream timeline of one steady-state pretrain step from a rocprofv3 kernel trace (rocpd sqlite): how long each HIP stream is
busy, how long both / one / neither run, the largest idle gaps of the compute stream, and per kernel family the time spent alone vs
overlapped with the other stream.  usage: tools/timeline.py results.db [out.md] [step_index_from_end]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, stream_id, start, end from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if r[0].startswith("adamw_kernel")]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
i0, i1 = ad[-k - 1], ad[-k]                        # one step: after an AdamW launch up to and including the next
step = rows[i0 + 1:i1 + 1]
t0, t1 = rows[i0][3], rows[i1][3]
wall = (t1 - t0) / 1e6
streams = defaultdict(list)
for n, s, a, b in step:
    streams[s].append((a, b, n))
main = max(streams, key=lambda s: sum(b - a for a, b, _ in streams[s]))


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n if len(n) < 60 else n[:57] + "..."


# sweep: busy state per stream
ev = []
for s, l in streams.items():
    for a, b, n in l:
        ev.append((a, 1, s)); ev.append((b, -1, s))
ev.sort()
act = defaultdict(int)
last = t0
both = only_main = only_side = idle = 0
for t, d, s in ev:
    dt = t - last
    m = act[main] > 0
    o = any(v > 0 for q, v in act.items() if q != main)
    if m and o: both += dt
    elif m: only_main += dt
    elif o: only_side += dt
    else: idle += dt
    last = t
    act[s] += d
idle += t1 - last
out = [f"step wall {wall:.2f} ms, {len(step)} dispatches on {len(streams)} streams (compute stream = id {main})", ""]
out.append("| stream | dispatches | busy ms | first start (ms into step) | last end |")
out.append("|---|---|---|---|---|")
for s, l in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    out.append(f"| {s}{' (compute)' if s == main else ''} | {len(l)} | {sum(b - a for a, b, _ in l) / 1e6:.2f} | {(l[0][0] - t0) / 1e6:.2f} | {(max(b for _, b, _ in l) - t0) / 1e6:.2f} |")
out.append("")
out.append(f"both kinds of stream busy {both / 1e6:.2f} ms, compute stream alone {only_main / 1e6:.2f} ms, other streams alone {only_side / 1e6:.2f} ms, nothing running {idle / 1e6:.2f} ms")
# gaps on the compute stream
ml = sorted(streams[main])
gaps = []
for (a0, b0, n0), (a1, b1, n1) in zip(ml, ml[1:]):
    if a1 > b0: gaps.append((a1 - b0, n0, n1, b0))
gsum = sum(g[0] for g in gaps)
out.append(f"\ncompute stream: {len(gaps)} gaps between consecutive kernels, {gsum / 1e6:.2f} ms in all; median {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.1f} us; the largest:")
out.append("\n| gap us | at ms | after | before |\n|---|---|---|---|")
for g, n0, n1, b0 in sorted(gaps, reverse=True)[:12]:
    out.append(f"| {g / 1e3:.1f} | {(b0 - t0) / 1e6:.2f} | `{short(n0)}` | `{short(n1)}` |")
# per family: alone vs overlapped time
side_iv = sorted((a, b) for s, l in streams.items() if s != main for a, b, _ in l)
main_iv = sorted((a, b) for a, b, _ in ml)


def overlap(a, b, iv):
    import bisect
    tot = 0
    i = bisect.bisect_left(iv, (a, a)) - 1
    i = max(i, 0)
    # intervals on other streams may themselves overlap; clip and merge on the fly
    cur = a
    while i < len(iv) and iv[i][0] < b:
        x, y = max(iv[i][0], cur), min(iv[i][1], b)
        if y > x:
            tot += y - x; cur = y
        i += 1
    return tot


fam = defaultdict(lambda: [0, 0, 0])
for s, l in streams.items():
    for a, b, n in l:
        f = fam[(short(n), "compute" if s == main else "side")]
        f[0] += 1; f[1] += b - a; f[2] += overlap(a, b, side_iv if s == main else main_iv)
out.append("\n| kernel | stream | calls | total ms | of which overlapped with the other stream ms |\n|---|---|---|---|---|")
for (n, s), (c, t, o) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    out.append(f"| `{n}` | {s} | {c} | {t / 1e6:.2f} | {o / 1e6:.2f} |")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write(txt + "\n")
