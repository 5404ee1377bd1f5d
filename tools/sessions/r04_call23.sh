#!/bin/bash
# Round 4, call 23: live PMC traffic of the bench line under tile_order = 3 ("grid") -- compare with 42.1 MB / launch of the default order
mkdir -p gpurun_out
CLORA_TILE_ORDER=grid timeout 900 python bench.py --no-cpu-baseline --no-full-step --no-ddim --steps 10 --warmup 3 2> gpurun_out/r04_bench_tile_grid.err | grep '^{' > gpurun_out/r04_bench_tile_grid.json
python -c "
import json
d = json.loads(open('gpurun_out/r04_bench_tile_grid.json').read().split('\n')[0]); r = d['roofline']
print('PMC tile_order grid: ms', d['ms_per_step'], 'family ms', r['family_ms_per_step'], 'traffic', r['traffic'], 'algorithmic', r['algorithmic_bytes_per_launch'], 'ratio', round(r['traffic'] / r['algorithmic_bytes_per_launch'], 3))" | tee -a gpurun_out/r04_ab_tile_grid.txt
