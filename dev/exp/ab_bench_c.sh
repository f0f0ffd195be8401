# A/B of an environment knob at a given batching: ab_bench_c.sh KNOB=VALUE COALESCE [repeats]
K=$1; C=$2; N=${3:-3}
for i in $(seq $N); do
  for v in "" "$K"; do
    env $v python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --coalesce $C --steps 64 --warmup 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('${v:-default}', 'C=$C', d['value'], d['ms_per_step'], d['config']['one_at_a_time_ms_per_step'])"
  done
done
