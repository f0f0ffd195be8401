#!/bin/bash
# Marginal cost of each kernel class / phase in the headline (4 launches in flight) and one launch at a time: the decode with those launches
# skipped (BOFI_EXP_SKIP in an experiments build of the library, BOFI_LIB_PATH; results invalid, timing only).
#   BOFI_EXPERIMENTS=1 python -m boficap_amd.build --force && cp boficap_amd/libboficap_hip.so build/ab/lib_exp.so && python -m boficap_amd.build --force
export BOFI_LIB_PATH=${BOFI_LIB_PATH:-build/ab/lib_exp.so}
C=${ABL_COALESCE:-5}; K=$((C * 40))      # batches per launch (ABL_COALESCE, default 5 as in rounds 3-4); the figures are ms per 320 images either way
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host --iter-budget off --coalesce $C --steps $K --warmup $((C * 8))"
p() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*5, 4))"; }
for k in none encoder filling loop "encoder,loop" "filling,loop" ffn attn qkv kv gen ; do
  echo "skip $k: in flight 4: $(BOFI_EXP_SKIP=$k $B 2>/dev/null | p) ms per 320 images   one at a time: $(BOFI_EXP_SKIP=$k $B --inflight 1 2>/dev/null | p)"
done
