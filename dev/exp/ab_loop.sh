#!/bin/bash
# bounding-loop experiments on the headline: forced row-GEMM tiles, idle iterations not enqueued (BOFI_EXP_ITERS=12: this workload's captions end by then)
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])"; }
for v in "" "BOFI_ROWGEMM_NT=4" "BOFI_ROWGEMM_NT=2" "BOFI_ROWGEMM_NT=1" "BOFI_EXP_ITERS=12" "BOFI_EXP_ITERS=13" "BOFI_EXP_SKIP=loop"; do
  echo "[$v] in flight 4: $(env $v $B 2>/dev/null | p)   one at a time: $(env $v $B --inflight 1 2>/dev/null | p)"
done
