#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of ROCm 7.2): per-kernel time per step.
usage: prof_db.py RESULTS.db NSTEPS [ROWS]     (NSTEPS "auto": the number of bound_init_kernel calls = engine launches in the trace; "auto:NAME": the calls of the
kernel whose name contains NAME -- e.g. auto:adam_step_kernel = optimiser steps of an XE trace)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
_key = "bound_init_kernel" if sys.argv[2] == "auto" else sys.argv[2][5:] if sys.argv[2].startswith("auto:") else None
n = float(sum(r[1] for r in rows if _key in r[0])) if _key else float(sys.argv[2])
tot = sum(r[2] for r in rows)
for name, calls, dur, avg in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    nm = re.sub(r"\(.*", "", name.replace("bofi::", "").replace("void ", "").replace("at::native::", ""))[:70]
    print(f"{nm:70s} calls/step={calls / n:7.1f} us/step={dur / n:9.1f} avg_us={avg:8.2f} pct={100 * dur / tot:5.1f}")
print("total kernel us per step", round(tot / n, 1), " launches per step", round(sum(r[1] for r in rows) / n, 1))
