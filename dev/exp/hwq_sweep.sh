B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --steps 96 --warmup 16"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('decodes_in_flight'))"; }
for q in 4 8 16; do for n in 4 6 8; do GPU_MAX_HW_QUEUES=$q $B --inflight $n 2>/dev/null | p "hwq=$q inflight=$n"; done; done
