#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 --pmc run (rocpd database).
usage: pmc_summary.py RESULTS.db [ROWS]   -- expects SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY, GRBM_GUI_ACTIVE
mfma_util_vs_duration = busy cycles / (kernel duration * 2.1 GHz * 1024 SIMDs); wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import json, re, sqlite3, sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14
acc = defaultdict(lambda: defaultdict(float))
calls, dur = defaultdict(set), defaultdict(float)
for name, disp, cname, val, d in db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    nm = re.sub(r"\(.*", "", name.replace("bofi::", "").replace("void ", "").replace("at::native::", ""))[:72]
    acc[nm][cname] += val
    if disp not in calls[nm]:
        calls[nm].add(disp)
        dur[nm] += d
out = []
for nm in sorted(dur, key=lambda k: -dur[k])[:rows]:
    n, c = len(calls[nm]), acc[nm]
    busy, wave, wait = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_WAIT_ANY", 0.0)
    out.append({"kernel": nm, "calls": n, "avg_us": round(dur[nm] / n / 1e3, 2), "mfma_busy_cycles_per_call": round(busy / n),
                "mfma_util_vs_duration": round(busy / (dur[nm] * 2.1 * 1024), 4) if dur[nm] else None,
                "wait_any_over_wave_cycles": round(wait / wave, 3) if wave else None})
print(json.dumps(out, indent=1))
