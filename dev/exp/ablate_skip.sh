#!/bin/bash
# Marginal cost of each kernel class in the headline (4 launches in flight) and one launch at a time: the decode with that class's launches
# skipped (BOFI_EXP_SKIP; results invalid, timing only).
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*5)"; }
for k in none ffn attn qkv kv gen loop "ffn,attn,qkv,kv,gen" ; do
  echo "skip $k: in flight 4: $(BOFI_EXP_SKIP=$k $B 2>/dev/null | p) ms per launch   one at a time: $(BOFI_EXP_SKIP=$k $B --inflight 1 2>/dev/null | p)"
done
