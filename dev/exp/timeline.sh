R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/tl; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --inflight 1 --steps 40 --warmup 8 > $OUT/tl.log 2>&1
cd $R; python dev/prof_timeline.py $(ls $OUT/tl/*kernel_trace.csv | head -1) > $OUT/timeline.txt 2>&1; rm -rf $OUT/tl; tail -3 $OUT/timeline.txt
