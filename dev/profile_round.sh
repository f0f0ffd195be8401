#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root): kernel-trace statistics of the
# default bench (16 batches of 64 per launch: 4 launches in flight, and one launch at a time), a one-launch timeline, HBM-side traffic
# (the default's 16 batches per launch, the driver's 5, and 1) and MFMA counters (separate --pmc passes, counters only), and the XE step's kernel statistics.
# Summaries land in gpurun_out/prof/ (copy into profiles/).  "step" in the summaries = one engine launch.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
TAG=${1:-r06}
[ "${ONLY:-all}" = "xe" ] || rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
run() { name=$1; shift; rocprofv3 "$@" > $OUT/$name.log 2>&1 || { echo "rocprofv3 $name failed"; tail -5 $OUT/$name.log; }; }
if [ "${ONLY:-all}" != "xe" ]; then
# 1. kernel statistics of the default command (16 batches per launch): steps / 16 launches per leg
run ks4 --kernel-trace --stats -d $OUT/ks4 -o ks4 -- $B --steps 320 --warmup 64
run ks1 --kernel-trace --stats -d $OUT/ks1 -o ks1 -- $B --inflight 1 --steps 320 --warmup 64
# 2. timeline of one launch
run tl --kernel-trace --output-format csv -d $OUT/tl -o tl -- $B --coalesce 5 --inflight 1 --hint 4 --steps 40 --warmup 10
run tla --kernel-trace --output-format csv -d $OUT/tla -o tl -- $B --coalesce 5 --inflight 1 --steps 40 --warmup 10
# 3. counters (own passes; --hint 4: the headline's kernel forms, serialised): default batching, and one batch per launch
run fetch --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- $B --coalesce 5 --inflight 1 --hint 4 --steps 40 --warmup 10
run write --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- $B --coalesce 5 --inflight 1 --hint 4 --steps 40 --warmup 10
run fetch16 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch16 -o fetch -- $B --coalesce 16 --inflight 1 --hint 4 --steps 64 --warmup 16
run write16 --pmc WRITE_SIZE --kernel-trace -d $OUT/write16 -o write -- $B --coalesce 16 --inflight 1 --hint 4 --steps 64 --warmup 16
run fetch1 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch1 -o fetch -- $B --coalesce 1 --inflight 1 --steps 20 --warmup 4
run write1 --pmc WRITE_SIZE --kernel-trace -d $OUT/write1 -o write -- $B --coalesce 1 --inflight 1 --steps 20 --warmup 4
# (config 5: batch 256 + 3 refinement rounds, one launch at a time -- VERDICT r5 item 6)
run fetch5 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch5 -o fetch -- $B --batch 256 --refine 3 --coalesce 1 --inflight 1 --hint 4 --steps 12 --warmup 4
run write5 --pmc WRITE_SIZE --kernel-trace -d $OUT/write5 -o write -- $B --batch 256 --refine 3 --coalesce 1 --inflight 1 --hint 4 --steps 12 --warmup 4
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/mfma -o mfma -- $B --coalesce 5 --inflight 1 --hint 4 --steps 40 --warmup 10
fi
# 4. XE step
run xe --kernel-trace --stats -d $OUT/xe -o xe -- python3 $R/bench.py --mode xe --steps 10 --warmup 3 --no-cpu-baseline
run xef --pmc FETCH_SIZE --kernel-trace -d $OUT/xef -o f -- python3 $R/bench.py --mode xe --steps 6 --warmup 2 --no-cpu-baseline
run xew --pmc WRITE_SIZE --kernel-trace -d $OUT/xew -o w -- python3 $R/bench.py --mode xe --steps 6 --warmup 2 --no-cpu-baseline
cd $R
db() { ls $OUT/$1/*.db 2>/dev/null | head -1; }
# "step" of the summaries = one engine launch (counted in the trace: dispatches of bound_init_kernel)
if [ "${ONLY:-all}" != "xe" ]; then
python dev/prof_db.py $(db ks4) auto 24 > $OUT/${TAG}_inflight4_kernel_stats.txt 2>&1
python dev/prof_db.py $(db ks1) auto 24 > $OUT/${TAG}_one_at_a_time_kernel_stats.txt 2>&1
python dev/prof_timeline.py $(ls $OUT/tl/*kernel_trace.csv | head -1) > $OUT/${TAG}_one_launch_timeline.txt 2>&1                   # the headline's (throughput) kernel forms, one launch at a time
python dev/prof_timeline.py $(ls $OUT/tla/*kernel_trace.csv | head -1) > $OUT/${TAG}_one_launch_timeline_alone_forms.txt 2>&1     # the forms a lone decode runs (hint 1)
python dev/pmc_traffic.py $(db fetch) $(db write) auto > $OUT/${TAG}_hbm_traffic_coalesce5.json 2>&1
python dev/pmc_traffic.py $(db fetch16) $(db write16) auto > $OUT/${TAG}_hbm_traffic_coalesce16.json 2>&1
python dev/pmc_traffic.py $(db fetch1) $(db write1) auto > $OUT/${TAG}_hbm_traffic.json 2>&1
python dev/pmc_traffic.py $(db fetch5) $(db write5) auto > $OUT/${TAG}_hbm_traffic_config5.json 2>&1
python dev/pmc_summary.py $(db mfma) 14 > $OUT/${TAG}_mfma_util_pmc.json 2>&1
fi
python dev/prof_db.py $(db xe) auto:uic_criterion_kernel 30 > $OUT/${TAG}_xe_step_kernel_stats.txt 2>&1
python dev/pmc_traffic.py $(db xef) $(db xew) auto:uic_criterion_kernel > $OUT/${TAG}_xe_hbm_traffic.json 2>&1      # 6 + 2 steps + the eager tally pass + the capture warm-up
rm -rf $OUT/ks4 $OUT/ks1 $OUT/tl $OUT/tla $OUT/fetch $OUT/write $OUT/fetch16 $OUT/write16 $OUT/fetch1 $OUT/write1 $OUT/fetch5 $OUT/write5 $OUT/mfma $OUT/xe $OUT/xef $OUT/xew
ls -la $OUT
