# CU-partitioned streams (bench.py --cu-partitions P with BOFI_GEMM_PERS_GRID = 256 / P) against the default
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --steps 120 --warmup 20"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_ms_per_step'))"; }
$B 2>/dev/null | p "default (4 in flight)"
BOFI_GEMM_PERS_GRID=128 $B --cu-partitions 2 --inflight 2 2>/dev/null | p "2 partitions, 2 in flight"
BOFI_GEMM_PERS_GRID=128 $B --cu-partitions 2 --inflight 4 2>/dev/null | p "2 partitions, 4 in flight"
BOFI_GEMM_PERS_GRID=128 $B --cu-partitions 2 --inflight 6 2>/dev/null | p "2 partitions, 6 in flight"
BOFI_GEMM_PERS_GRID=64 $B --cu-partitions 4 --inflight 4 2>/dev/null | p "4 partitions, 4 in flight"
BOFI_GEMM_PERS_GRID=64 $B --cu-partitions 4 --inflight 8 2>/dev/null | p "4 partitions, 8 in flight"
$B 2>/dev/null | p "default (4 in flight)"
