B="python bench.py --mode xe --steps 40 --warmup 8 --no-cpu-baseline"
p() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
$B 2>/dev/null | p default
BOFI_LN_WS=0 $B 2>/dev/null | p ln_ws=0
BOFI_GEMM_HEUR2=0 $B 2>/dev/null | p heur2=0
BOFI_GEMM_HEUR2=0 BOFI_LN_WS=0 $B 2>/dev/null | p both_off
