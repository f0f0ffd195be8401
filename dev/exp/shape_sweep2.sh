#!/bin/bash
# headline by (batches per launch, launches in flight), larger launches
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in "10 4" "12 4" "16 4" "20 4" "16 3" "20 3" "16 2" "20 2"; do
    set -- $cfg
    python bench.py --coalesce $1 --inflight $2 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host --steps 480 --warmup 80 2>gpurun_out/ss2.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep coalesce $1 inflight $2: %.1f img/s  %.4f ms/step  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
" || tail -3 gpurun_out/ss2.err
  done
done
