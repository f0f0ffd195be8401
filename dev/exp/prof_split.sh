cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export BOFI_RB_ATTN_SPLIT=$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_split$v -o ks -- python3 $R/bench.py --steps 80 --warmup 20 --coalesce 5 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host > $R/gpurun_out/prof_split$v.log 2>&1
  cd $R && python dev/prof_db.py $(ls gpurun_out/prof_split$v/*.db | head -1) auto 26 > gpurun_out/r6_split${v}_kernel_stats.txt 2>&1; cd /tmp
  rm -rf $R/gpurun_out/prof_split$v
done
cat $R/gpurun_out/r6_split0_kernel_stats.txt; echo ======; cat $R/gpurun_out/r6_split1_kernel_stats.txt
