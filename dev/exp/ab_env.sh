#!/bin/bash
# interleaved same-box A/B of the headline under environment settings: dev/exp/ab_env.sh "BOFI_X=1" "BOFI_X=0" ... (3 rounds);
# BENCH_ARGS (default "--steps 400 --warmup 40") = the bench arguments, e.g. BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" for the driver's command
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-secondary --no-gemm-roofline ${BENCH_ARGS:---steps 400 --warmup 40} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep $v: %.1f img/s  %.4f ms/step  frac %.4f  one-at-a-time %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['one_at_a_time']['launch_ms']))
"
  done
done
