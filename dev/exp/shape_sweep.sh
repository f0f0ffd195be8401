#!/bin/bash
# headline by (batches per launch, launches in flight), interleaved over 2 rounds: fewer, larger launches keep fewer distinct weight sets in each XCD's L2
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in "5 4" "10 2" "10 3" "8 3" "10 4" "6 4" "4 4" "5 3"; do
    set -- $cfg
    python bench.py --coalesce $1 --inflight $2 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host --steps 480 --warmup 80 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep coalesce $1 inflight $2: %.1f img/s  %.4f ms/step  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
"
  done
done
