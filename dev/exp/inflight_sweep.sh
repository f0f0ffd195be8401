# throughput against the number of decodes in flight / batches per launch, same box
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --steps 96 --warmup 16"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for n in 1 2 3 4; do $B --inflight $n 2>/dev/null | p "inflight=$n"; done
for c in 2 4; do for n in 1 2 4; do $B --coalesce $c --inflight $n 2>/dev/null | p "coalesce=$c inflight=$n"; done; done
$B --ids-only 2>/dev/null | p "ids-only inflight=4"
