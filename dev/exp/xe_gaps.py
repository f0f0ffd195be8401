#!/usr/bin/env python3
"""Gaps between the kernels of one XE step from a rocprofv3 rocpd database (kernel trace): python dev/exp/xe_gaps.py results.db
A step = the launches between two adam_step_kernel launches; reports the median step's sum of durations, sum of gaps, and the gap histogram."""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[2]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b + 1]
    dur = sum(e - s for s, e, _ in seg) / 1e3
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
    steps.append((seg[-1][1] - seg[0][0], dur, sum(gaps), len(seg), gaps, seg))
steps.sort(key=lambda t: t[0])
span, dur, gap, n, gaps, seg = steps[len(steps) // 2]
print(f"median step: span {span / 1e3:.1f} us, {n} launches, sum of durations {dur:.1f} us, sum of gaps {gap:.1f} us")
h = [0] * 8
for g in gaps:
    h[min(7, int(g))] += 1
print("gap histogram (us bins 0-1, 1-2, ..., 7+):", h)
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:12]
for g, i in big:
    print(f"  gap {g:7.2f} us after {seg[i][2][:60]} before {seg[i + 1][2][:60]}")
import collections, re
cnt = collections.defaultdict(lambda: [0, 0.0])
for s, e, nme in seg:
    k = re.sub(r"\.kd$", "", nme)[:90]
    cnt[k][0] += 1; cnt[k][1] += (e - s) / 1e3
print("launches of the median step by kernel (count, total us, avg us):")
for k, (c, t) in sorted(cnt.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:4d} {t:8.1f} {t / c:7.2f}  {k}")
