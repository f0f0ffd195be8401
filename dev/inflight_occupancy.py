#!/usr/bin/env python3
"""How full is the chip with several decodes in flight?  From a rocprofv3 --kernel-trace csv of the default bench: the timed part of the trace cut
into slices at every kernel start / end; per slice the kernels running and the CUs their workgroups can occupy at most (sum over kernels of
min(workgroups, 256), capped at 256: a row-block workgroup owns its CU).  Reports the share of wall time by number of concurrent kernels and by CU
demand, and per kernel class the time-weighted share of the chip it had while it ran.
usage: inflight_occupancy.py KERNEL_TRACE.csv [skip_fraction_front]"""
import csv, re, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], max(1, grid // wg), wg))
rows.sort()
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
# the in-flight leg = the longest run of 1-ms bins in which kernels from several launches overlap (>= 1.5 kernels running on average)
nb = int((t_hi - t_lo) // 1_000_000) + 1
busy = [0.0] * nb
for s, e, n, g, w in rows:
    b0, b1 = int((s - t_lo) // 1_000_000), int((e - t_lo) // 1_000_000)
    for b in range(b0, b1 + 1):
        busy[b] += (min(e, t_lo + (b + 1) * 1_000_000) - max(s, t_lo + b * 1_000_000)) / 1e6
best, cur = (0, 0), None
for b in range(nb + 1):
    on = b < nb and busy[b] >= 1.5
    if on and cur is None:
        cur = b
    if not on and cur is not None:
        if b - cur > best[1] - best[0]:
            best = (cur, b)
        cur = None
t0, t1 = t_lo + (best[0] + 1) * 1_000_000, t_lo + (best[1] - 1) * 1_000_000
ev = []
for i, (s, e, n, g, w) in enumerate(rows):
    if e <= t0 or s >= t1:
        continue
    ev.append((max(s, t0), 1, i)); ev.append((min(e, t1), -1, i))
ev.sort()
short = lambda n: re.sub(r"\(.*", "", re.sub(r"<.*", "", n.replace("bofi::", "").replace("void ", "")))[:40]
active, last = set(), t0
by_n, by_cu, cls_time, cls_share = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(float)
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        demand = [min(rows[j][3], 256) * (1.0 if rows[j][4] >= 512 else rows[j][4] / 512.0 if rows[j][4] >= 256 else 0.25) for j in active]
        tot = sum(demand)
        by_n[len(active)] += dt
        by_cu[min(4, int(tot // 64))] += dt
        for j, dm in zip(active, demand):
            c = short(rows[j][2])
            cls_time[c] += dt
            cls_share[c] += dt * (dm / max(tot, 256.0))
    last = t
    active.add(i) if d > 0 else active.discard(i)
wall = last - t0
print(f"in-flight leg: {wall / 1e6:.2f} ms of wall time (of a {(t_hi - t_lo) / 1e6:.0f} ms trace with {len(rows)} kernels)")
print("wall-time share by kernels running at once: " + "  ".join(f"{k}: {100 * v / wall:.1f} %" for k, v in sorted(by_n.items())))
print("wall-time share by CU demand (sum of min(workgroups, 256) x CU share of a workgroup; 256 = the chip): " +
      "  ".join(f"{'%d-%d' % (64 * k, 64 * k + 63) if k < 4 else '>= 256'}: {100 * v / wall:.1f} %" for k, v in sorted(by_cu.items())))
print("per kernel class: time it was running (share of wall), and the mean share of the chip's CUs it could claim meanwhile")
for c, tt in sorted(cls_time.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {c:42s} running {100 * tt / wall:5.1f} % of wall   chip share while running {100 * cls_share[c] / tt:5.1f} %")
