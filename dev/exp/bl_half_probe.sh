#!/bin/bash
# gate for the two-workgroup form of the bounding-loop kernel: per-iteration time of bound_loop_kernel when a workgroup streams only HALF of the feed-forward's weights
# (BOFI_BL_DBG=64: timing-only, results invalid) against the whole stream: kernel duration / iterations, one 320-image launch at a time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 64; do
  BOFI_BL_DBG=$v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blh$v -o ks -- python3 $R/bench.py --steps 40 --warmup 10 --coalesce 5 --inflight 1 --hint 4 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host > $R/gpurun_out/blh$v.log 2>&1
  (cd $R && python dev/prof_db.py $(ls gpurun_out/blh$v/*.db | head -1) auto 30 | grep -i "bound_loop\|total kernel"; grep -o '"bound_iterations": [0-9]*' gpurun_out/blh$v.log | head -1)
  rm -rf $R/gpurun_out/blh$v
done
