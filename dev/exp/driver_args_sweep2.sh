#!/bin/bash
# the driver's arguments (--steps 20 --warmup 5) under other batchings of the 20 steps, with the loop kernel
for c in 2 4 5 10; do for nf in 2 4; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --coalesce $c --inflight $nf --no-cpu-baseline --no-secondary --no-gemm-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('coalesce $c inflight $nf: %.1f img/s  %.4f ms/step  frac %.4f  regions %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['region_ms']))"
done; done
