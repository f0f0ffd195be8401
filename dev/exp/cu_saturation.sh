#!/bin/bash
# Is the default bench bound by CUs or by latency chains?  The same run on a part of the CUs (HSA_CU_MASK, process-wide).
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"; }
for m in "" "0:0-223" "0:0-191" "0:0-127"; do
  echo "HSA_CU_MASK=[$m] in flight 4: $(HSA_CU_MASK=$m $B 2>/dev/null | p)   one at a time: $(HSA_CU_MASK=$m $B --inflight 1 2>/dev/null | p)"
done
