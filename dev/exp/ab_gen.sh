#!/bin/bash
# A/B: headline with the generator as [loop without epilogue] + [generator], vocab_finalize skipped (BOFI_EXP_GEN=1; results invalid, timing only)
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])"; }
for i in 1 2; do
  echo "base      : $($B 2>/dev/null | p)"
  echo "exp_gen   : $(BOFI_EXP_GEN=1 $B 2>/dev/null | p)"
  echo "base 1@t  : $($B --inflight 1 2>/dev/null | p)"
  echo "exp 1@t   : $(BOFI_EXP_GEN=1 $B --inflight 1 2>/dev/null | p)"
done
