# hardware queues x launches in flight with round 4's kernels (coalesce 5): img/s, ms per batch, roofline.frac
for q in 4 8; do for n in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --steps $((5 * n * 6)) --warmup $((5 * n)) --inflight $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('GPU_MAX_HW_QUEUES=$q inflight $n:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['decodes_in_flight'], flush=True)"
done; done
