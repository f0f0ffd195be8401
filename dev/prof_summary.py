#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats directory: per-kernel time per decode. usage: prof_summary.py DIR NDECODES"""
import csv, glob, sys
d, n = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    nm = r['Name'].replace('bofi::', '').replace('void ', '')[:64]
    print(f"{nm:64s} calls/dec={int(r['Calls'])/n:7.1f} us/dec={float(r['TotalDurationNs'])/1e3/n:8.1f} avg_us={float(r['AverageNs'])/1e3:7.2f} pct={float(r['Percentage']):5.1f}")
print('total kernel us per decode', round(tot / 1e3 / n, 1))
