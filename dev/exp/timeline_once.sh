#!/bin/bash
# one-launch kernel timeline of the default decode (one launch at a time): gpurun_out/timeline.txt.  Extra environment from the caller.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/tl_once
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --inflight 1 --steps 40 --warmup 10 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cd $R
python dev/prof_timeline.py $(ls $OUT/*kernel_trace.csv | head -1) > $R/gpurun_out/timeline${1:+_$1}.txt 2>&1
rm -rf $OUT
