# throughput against batches per launch x launches in flight, same box (steps chosen as a multiple of every combination)
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --steps 240 --warmup 40"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_ms_per_step'))"; }
for c in 1 2 4 5 8; do for n in 2 3 4 6; do $B --coalesce $c --inflight $n 2>/dev/null | p "coalesce=$c inflight=$n"; done; done
