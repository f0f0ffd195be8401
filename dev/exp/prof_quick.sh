#!/bin/bash
# quick look: kernel statistics + timeline of one launch at a time (5 batches of 64 per launch) -> gpurun_out/prof/<tag>_*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
TAG=${1:-quick}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
rocprofv3 --kernel-trace --stats -d $OUT/ks1 -o ks1 -- $B --inflight 1 --steps 200 --warmup 20 > $OUT/ks1.log 2>&1 || tail -5 $OUT/ks1.log
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- $B --inflight 1 --steps 40 --warmup 10 > $OUT/tl.log 2>&1 || tail -5 $OUT/tl.log
cd $R
python dev/prof_db.py $(ls $OUT/ks1/*.db | head -1) 48 30 > $OUT/${TAG}_one_at_a_time_kernel_stats.txt 2>&1
python dev/prof_timeline.py $(ls $OUT/tl/*kernel_trace.csv | head -1) > $OUT/${TAG}_one_launch_timeline.txt 2>&1
rm -rf $OUT/ks1 $OUT/tl
cat $OUT/${TAG}_one_at_a_time_kernel_stats.txt
