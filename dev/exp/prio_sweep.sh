#!/bin/bash
# default bench with stream priorities on the launch streams (experiment)
for p in "" "-1,0" "-1,-1,0,0" "-1,0,0,0" "-1"; do
  BOFI_BENCH_STREAM_PRIO="$p" python bench.py --no-cpu-baseline --no-secondary --no-gemm-roofline --steps 400 --warmup 40 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('priorities [$p]: %.1f img/s  %.4f ms/step  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
