# the driver's arguments (--steps 20 --warmup 5) under different batchings
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --gpus 1 --steps 20 --warmup 5"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_ms_per_step'), d['config'].get('nan_in_output'))"; }
for rep in 1 2; do
$B 2>/dev/null | p "default"
$B --coalesce 10 --inflight 2 2>/dev/null | p "coalesce=10 inflight=2"
$B --coalesce 10 --inflight 4 2>/dev/null | p "coalesce=10 inflight=4"
$B --coalesce 20 --inflight 1 2>/dev/null | p "coalesce=20 inflight=1"
$B --coalesce 4 --inflight 4 2>/dev/null | p "coalesce=4 inflight=4"
done
