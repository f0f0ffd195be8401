# HBM-side traffic per decode (one at a time) with the default XCD layout choice and with row bands forced to 8 (round-1 layout)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/traffic; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --inflight 1 --steps 20 --warmup 4"
for tag in auto b8; do
  if [ $tag = b8 ]; then export BOFI_GEMM_BANDS=8; else unset BOFI_GEMM_BANDS; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f_$tag -o f -- $B > $OUT/f_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w_$tag -o w -- $B > $OUT/w_$tag.log 2>&1
  (cd $R && python dev/pmc_traffic.py $(ls $OUT/f_$tag/*.db | head -1) $(ls $OUT/w_$tag/*.db | head -1) 26 > $OUT/traffic_$tag.json 2>&1)
  rm -rf $OUT/f_$tag $OUT/w_$tag
done
cd $R; head -c 600 $OUT/traffic_auto.json; echo; head -c 600 $OUT/traffic_b8.json
