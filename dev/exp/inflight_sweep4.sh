#!/bin/bash
# batches per launch x launches in flight with the row-block kernels (round 3)
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline"
for c in 4 5 8 10; do for i in 2 3 4; do
  st=$((c * i * 8))
  echo "coalesce $c inflight $i: $($B --coalesce $c --inflight $i --steps $st --warmup $((c * i * 2)) 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"])')"
done; done
