B="python bench.py --mode xe --steps 60 --warmup 8 --no-cpu-baseline"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])"; }
for i in 1 2; do
$B 2>/dev/null | p default
env $1 $B 2>/dev/null | p "$1"
done
