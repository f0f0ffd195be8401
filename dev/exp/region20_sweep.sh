# the driver's region (--steps 20) under other partitions of its 20 batches: batches per launch x launches in flight
for cfg in "5 4" "10 2" "20 1" "4 5" "2 4" "1 4"; do set -- $cfg
  python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --steps 20 --warmup 20 --coalesce $1 --inflight $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('coalesce $1 inflight $2:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['region_ms'], flush=True)"
done
