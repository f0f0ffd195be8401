# launches in flight x batches per launch with round 4's kernels (default bench otherwise): img/s, ms per batch, roofline.frac
for c in 4 5 10; do for n in 2 3 4 5 6 8; do
  python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --steps $((c * n * 6)) --warmup $((c * n)) --coalesce $c --inflight $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('coalesce $c inflight $n:', d['value'], d['ms_per_step'], d['roofline']['frac'], flush=True)"
done; done
