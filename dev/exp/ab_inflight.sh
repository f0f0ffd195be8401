#!/bin/bash
# default bench (5 batches per launch) against launches in flight, with the loop kernel and with the five-launch iterations
mkdir -p gpurun_out
for nf in 4 5 6 8; do
  for bl in 1 0; do
    BOFI_BOUND_LOOP=$bl python bench.py --inflight $nf --no-cpu-baseline --no-secondary --no-gemm-roofline --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('inflight $nf BOFI_BOUND_LOOP=$bl: %.1f img/s  %.4f ms/step  frac %.4f  one-at-a-time %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_at_a_time']['launch_ms']))
"
  done
done
