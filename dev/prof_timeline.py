#!/usr/bin/env python3
"""Timeline of ONE decode from a rocprofv3 --kernel-trace csv: per kernel start offset, duration and the gap to its predecessor.
usage: prof_timeline.py KERNEL_TRACE.csv [decode_index] [launches_per_decode]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# a decode starts at an embed/att_embed-sized marker: use bound_init as the anchor
anchors = [i for i, n in enumerate(names) if "bound_init" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(anchors) // 2
a, b = anchors[k], anchors[k + 1]
# walk back to the decode's first kernel: the encoder launches precede bound_init
per = b - a
start = a - (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
seg = rows[a - 40:b - 40] if per > 60 else rows[a:b]
t0 = seg[0][0]
prev_end = seg[0][0]
tot_k = tot_g = 0
short = lambda n: re.sub(r"\(.*", "", n.replace("bofi::", "").replace("void ", ""))[:60]
for s, e, n in seg:
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    tot_k += dur; tot_g += max(gap, 0)
    print(f"{(s - t0) / 1e3:9.2f} us  dur {dur:7.2f}  gap {gap:6.2f}  {short(n)}")
    prev_end = e
print(f"kernels {len(seg)}  sum of durations {tot_k:.1f} us  sum of gaps {tot_g:.1f} us  span {(seg[-1][1] - t0) / 1e3:.1f} us")
