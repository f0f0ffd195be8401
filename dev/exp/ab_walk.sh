#!/bin/bash
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])"; }
for v in BOFI_ROWGEMM_WALK=0 BOFI_ROWGEMM_WALK=1 BOFI_ROWGEMM_WALK=4 BOFI_ROWGEMM_WALK=0 BOFI_ROWGEMM_WALK=1; do
  echo "[$v] in flight 4: $(env $v $B 2>/dev/null | p)   one at a time: $(env $v $B --inflight 1 2>/dev/null | p)"
done
