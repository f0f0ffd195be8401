# LDS-active cycles against wave cycles / GPU-active cycles per kernel of the default decode (is the LDS pipe the GEMM's limiter?)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ldsb; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $OUT/p -o p -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --inflight 1 --steps 40 --warmup 8 > $OUT/p.log 2>&1
cd $R; python - <<'PY'
import sqlite3, glob, re, json
from collections import defaultdict
db = sqlite3.connect(glob.glob("gpurun_out/ldsb/p/*.db")[0])
acc, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(float)
for name, disp, cname, val, d in db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    nm = re.sub(r"\(.*", "", name.replace("bofi::", "").replace("void ", ""))[:60]
    acc[nm][cname] += val
    if disp not in n[nm]:
        n[nm].add(disp); dur[nm] += d
for nm in sorted(dur, key=lambda k: -dur[k])[:8]:
    c = acc[nm]; k = len(n[nm])
    print(f"{nm:62s} calls {k:5d} avg_us {dur[nm]/k/1e3:7.2f}  LDS_IDX_ACTIVE/call {c['SQ_LDS_IDX_ACTIVE']/k:12.0f}  GUI_ACTIVE/call {c['GRBM_GUI_ACTIVE']/k:10.0f}  SQ_BUSY/call {c['SQ_BUSY_CYCLES']/k:12.0f}  conflict/active {c['SQ_LDS_BANK_CONFLICT']/max(1,c['SQ_LDS_IDX_ACTIVE']):.3f}")
PY
rm -rf $OUT/p
