#!/usr/bin/env python3
"""HBM-side traffic per step from two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; both in KiB).
usage: pmc_traffic.py FETCH.db WRITE.db NSTEPS|auto  -- FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B,
MI355X_MICROARCH.md).  Prints a JSON object with bytes per step and the top kernels of each direction."""
import json, re, sqlite3, sys
from collections import defaultdict


def total(path, counter):
    db = sqlite3.connect(path)
    per = defaultdict(float)
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        per[re.sub(r"\(.*", "", name.replace("bofi::", "").replace("void ", ""))[:60]] += val
    return sum(per.values()), sorted(per.items(), key=lambda kv: -kv[1])[:6]


if sys.argv[3] == "auto" or sys.argv[3].startswith("auto:"):    # engine launches in the trace = dispatches of bound_init_kernel ("auto:NAME": of the kernel whose name contains NAME)
    key = "bound_init_kernel" if sys.argv[3] == "auto" else sys.argv[3][5:]
    n = float(sqlite3.connect(sys.argv[1]).execute("select count(distinct dispatch_id) from counters_collection where kernel_name like ?", ("%" + key + "%",)).fetchone()[0])
else:
    n = float(sys.argv[3])
f, ftop = total(sys.argv[1], "FETCH_SIZE")
w, wtop = total(sys.argv[2], "WRITE_SIZE")
print(json.dumps({"hbm_bytes_per_step": round((2.0 * f + w) * 1024.0 / n), "fetch_size_kb_per_step": round(f / n, 1), "write_size_kb_per_step": round(w / n, 1),
                  "steps": n, "top_fetch_kernels_kb_per_step": [[k, round(v / n, 1)] for k, v in ftop],
                  "top_write_kernels_kb_per_step": [[k, round(v / n, 1)] for k, v in wtop]}, indent=1))
