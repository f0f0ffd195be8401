#!/bin/bash
# the driver's command (--steps 20 --warmup 5: one timed region of 20 batches, median of 5) by batches per launch x launches in flight
mkdir -p gpurun_out
for rep in 1 2; do
  for cfg in "5 4" "4 5" "2 4" "10 2" "20 1" "1 4"; do
    set -- $cfg
    python bench.py --gpus 1 --steps 20 --warmup 5 --coalesce $1 --inflight $2 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host 2>gpurun_out/k20.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep coalesce $1 inflight $2: %.1f img/s  %.4f ms/step  frac %.4f  regions %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('region_ms')))
" || tail -3 gpurun_out/k20.err
  done
done
