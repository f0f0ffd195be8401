#!/bin/bash
# kernel statistics of the self-critical step (bench.py --mode rl) under rocprofv3: gpurun_out/rl_kernel_stats.txt ("step" = one rl_step)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/rl_prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o rl -- python3 $R/bench.py --mode rl --steps 20 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cd $R
python dev/prof_db.py $(ls $OUT/*.db | head -1) 28 40 > $R/gpurun_out/rl_kernel_stats.txt 2>&1      # 20 timed + 3 warm-up + 1 eager tally + 4 reference-estimator steps (those run 12 forwards each)
rm -rf $OUT
