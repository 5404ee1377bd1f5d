#!/bin/bash
# Round 4, call 13: (a) the product processors on a stock CrossAttention module, on the real library; (b) tiles for the FeedForward GEMMs
# timed WITH their fused GEGLU activation (tools/tune_gemm.py --geglu; the default pass times the bare GEMM and picked the 256x256 one-block-
# per-CU tile for them); same-box A/B of the train step and DDIM-50 before / after.
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_full_topology_gpu.py -q -m gpu -k "stock" -s ) > gpurun_out/r04_gputest_stock.log 2>&1
tail -3 gpurun_out/r04_gputest_stock.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_g_before.json
( timeout 900 python tools/tune_gemm.py --geglu ) > gpurun_out/r04_tune_geglu.log 2>&1
grep -E "signatures|best tile|sum over|wrote" gpurun_out/r04_tune_geglu.log | cut -c1-170
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_g_after.json
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for tbl in before after before after; do
  CLORA_GEMM_TUNING_FILE=$PWD/gpurun_out/r04_table_g_$tbl.json timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB geglu-table $tbl', d['ms_per_step'], d['value'], d['ddim50']['latency_s'])" | tee -a gpurun_out/r04_ab_geglu_table.txt
done
