#!/bin/bash
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'])"; }
for c in 5 8 10 16; do for f in 2 3 4; do
  echo "coalesce $c inflight $f: $($B --coalesce $c --inflight $f --steps $((c*16)) --warmup $((c*4)) 2>/dev/null | p)"
done; done
