#!/bin/bash
# the same code on round 2's FULL preset (captions of 0-1 phrases or all 20 positions, T = 12) and round 3's (every length, T = 8-10)
B="python bench.py --no-secondary --no-cpu-baseline --no-gemm-roofline --no-from-host"
for p in full_r2 full; do
  for args in "" "--inflight 1" "--coalesce 1 --steps 80 --warmup 16"; do
    echo "preset $p $args: $(BOFI_PRESET_FULL=$p $B $args 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], "T", d["config"]["bound_iterations"], "tokens", d["config"]["mean_tokens_per_image"])')"
  done
done
