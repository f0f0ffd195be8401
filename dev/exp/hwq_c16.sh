#!/bin/bash
# default bench (16 batches per launch) by hardware queues x launches in flight
for q in 4 8; do for nf in 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --inflight $nf --steps $((nf * 16 * 5)) --warmup $((nf * 16)) --coalesce 16 --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('GPU_MAX_HW_QUEUES=$q inflight $nf: %.1f img/s  frac %.4f' % (d['value'], d['roofline']['frac']))
"
done; done
