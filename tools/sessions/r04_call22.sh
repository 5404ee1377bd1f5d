#!/bin/bash
# Round 4, call 22: tile_order = 3 ("grid": per launch an (split, m, n) rectangle of tiles per XCD where whole divisors exist and the panel
# model prefers it) against the default (2): bit-identity tests on hardware, same-box A/B of the train step, and the live PMC traffic.
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "tile_order" ) > gpurun_out/r04_gputest_tile_grid.log 2>&1
tail -1 gpurun_out/r04_gputest_tile_grid.log
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --no-ddim --steps 30 --warmup 5"
for v in auto grid auto grid; do
  CLORA_TILE_ORDER=$v timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB tile_order $v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_tile_grid.txt
done
CLORA_TILE_ORDER=grid timeout 900 python bench.py --no-cpu-baseline --no-full-step --no-ddim --no-rocprof --steps 10 --warmup 3 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']; print('PMC tile_order grid: traffic', r['traffic'], 'algorithmic', r['algorithmic_bytes_per_launch'], 'ratio', round(r['traffic'] / r['algorithmic_bytes_per_launch'], 3))" | tee -a gpurun_out/r04_ab_tile_grid.txt
