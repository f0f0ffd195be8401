#!/bin/bash
# launches in flight x hardware queues with the persistent loop kernel (round 5: a decode's loop phase holds its queue for ~1 ms with 20 workgroups)
for q in 4 6 8; do for nf in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --inflight $nf --no-cpu-baseline --no-secondary --no-gemm-roofline --no-from-host --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('GPU_MAX_HW_QUEUES=$q inflight $nf: %.1f img/s  %.4f ms/step  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
"
done; done
