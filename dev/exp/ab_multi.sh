# A/B of several environment settings on the headline bench, interleaved on one box: ab_multi.sh REPEATS "K=V ..." "K=V ..." ...
# (each argument after the first is one setting: a space-separated list of KEY=VALUE, "" = default).  Prints img/s, ms per batch in flight and one at a time.
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    env $v python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('[${v:-default}]', d['value'], d['ms_per_step'], d['config']['one_at_a_time_ms_per_step'], d['roofline']['frac'], flush=True)"
  done
done
