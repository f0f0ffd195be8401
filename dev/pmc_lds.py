#!/usr/bin/env python3
"""Per-kernel LDS bank-conflict share from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (SQ_LDS_ADDR_CONFLICT) run.
usage: pmc_lds.py RESULTS.db [ROWS]"""
import json, re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
acc, calls, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(float)
for name, disp, cname, val, d in db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    nm = re.sub(r"\(.*", "", name.replace("bofi::", "").replace("void ", "").replace("at::native::", ""))[:72]
    acc[nm][cname] += val
    if disp not in calls[nm]:
        calls[nm].add(disp); dur[nm] += d
out = []
for nm in sorted(dur, key=lambda k: -dur[k])[:rows]:
    c = acc[nm]
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    out.append({"kernel": nm, "calls": len(calls[nm]), "avg_us": round(dur[nm] / len(calls[nm]) / 1e3, 2),
                "lds_bank_conflict_over_active": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 3) if act else None,
                "lds_addr_conflict_over_active": round(c.get("SQ_LDS_ADDR_CONFLICT", 0.0) / act, 3) if act else None})
print(json.dumps(out, indent=1))
