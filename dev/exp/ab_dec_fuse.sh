#!/bin/bash
# interleaved same-box A/B of the headline: decoder layers' attention sublayers as one launch (BOFI_RB_DEC_FUSE=1, rb_dec_attn_kernel) against the three launches (0)
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in 1 0; do
    BOFI_RB_DEC_FUSE=$v python bench.py --no-cpu-baseline --no-secondary --no-gemm-roofline --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep BOFI_RB_DEC_FUSE=$v: %.1f img/s  %.4f ms/step  frac %.4f  one-at-a-time %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_at_a_time']['launch_ms']))
"
  done
done
