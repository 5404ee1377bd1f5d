#!/bin/bash
# Round 4, call 12: tiles for the launches that carry their adapter's down-projection, timed WITH the extra operand rows
# (tools/tune_fused.py -> ":x" entries of the launch table); same-box A/B of the train step and DDIM-50 before / after.
mkdir -p gpurun_out
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_x_before.json
( timeout 1200 python tools/tune_fused.py ) > gpurun_out/r04_tune_fused.log 2>&1
grep -E "signatures|sum over|wrote" gpurun_out/r04_tune_fused.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_x_after.json
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for tbl in before after before after; do
  CLORA_GEMM_TUNING_FILE=$PWD/gpurun_out/r04_table_x_$tbl.json timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB xtable $tbl', d['ms_per_step'], d['value'], d['ddim50']['latency_s'])" | tee -a gpurun_out/r04_ab_xtable.txt
done
