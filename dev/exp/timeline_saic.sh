R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/tls; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python3 $R/dev/time_saic.py --multi > $OUT/tl.log 2>&1
cd $R; python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tls/tl/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last decode: take the last 1500 kernels, print a window from the middle of the run
names = [r["Kernel_Name"][:70] for r in rows]
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
n = len(rows)
init = [i for i, nm in enumerate(names) if "saic_init" in nm][-1]
idx = [i for i, nm in enumerate(names) if "saic_rows" in nm and i > init]
s = idx[1] if len(idx) > 1 else n // 2
with open("gpurun_out/tls/window.txt", "w") as o:
    for i in range(s - 14, min(s + 62, n)):
        o.write(f"{(t[i][0]-t[s][0])/1e3:9.2f} us dur {(t[i][1]-t[i][0])/1e3:7.2f} gap {(t[i][0]-t[i-1][1])/1e3:6.2f}  {names[i]}\n")
PY
rm -rf $OUT/tl; tail -90 $OUT/window.txt
