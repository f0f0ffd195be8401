# A/B of several environment knobs (one string, space separated) on the headline bench: ab_bench2.sh "K1=V1 K2=V2" [repeats]
K=$1; N=${2:-3}
for i in $(seq $N); do
  for v in "" "$K"; do
    env $v python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('${v:-default}', d['value'], d['ms_per_step'], d['config']['one_at_a_time_ms_per_step'])"
  done
done
