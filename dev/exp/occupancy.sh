# kernel trace of the default bench (four launches in flight) -> dev/inflight_occupancy.py
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/occ; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --steps 200 --warmup 20 > $OUT/tl.log 2>&1
cd $R; python dev/inflight_occupancy.py $(ls $OUT/tl/*kernel_trace.csv | head -1) 0.3 > $OUT/occupancy.txt 2>&1; rm -rf $OUT/tl; cat $OUT/occupancy.txt
