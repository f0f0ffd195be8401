#!/bin/bash
# A/B of the row-block kernels' row threshold: default bench (5 x 64 per launch, 4 in flight) and one batch per launch
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline"
for mr in 1000000 4096 1024; do
  for args in "" "--coalesce 1 --steps 80 --warmup 16" "--inflight 1" "--coalesce 1 --inflight 1 --steps 40 --warmup 8"; do
    echo "BOFI_RB_MIN_ROWS=$mr $args: $(BOFI_RB_MIN_ROWS=$mr $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
