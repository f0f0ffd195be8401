#!/bin/bash
# bound_loop_kernel with one / two workgroups per group (BOFI_BL_PAIR): kernel duration per 320-image launch, one launch at a time (the headline's kernel forms: --hint 4)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  BOFI_BL_PAIR=$v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blp$v -o ks -- python3 $R/bench.py --steps 40 --warmup 10 --coalesce 5 --inflight 1 --hint 4 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host > $R/gpurun_out/blp$v.log 2>&1
  (cd $R && echo "BOFI_BL_PAIR=$v" && python dev/prof_db.py $(ls gpurun_out/blp$v/*.db | head -1) auto 30 | grep -i "bound_loop\|bl_zero\|total kernel"; grep -o '"bound_iterations": [0-9]*' gpurun_out/blp$v.log | head -1)
  rm -rf $R/gpurun_out/blp$v
done
