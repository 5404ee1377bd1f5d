#!/bin/bash
# Round 4, call 6: tuning passes.  (a) VAE conv / GEMM signatures (encode + decode of 4 x 512^2 and of 8 x 512^2) merged into the
# launch table; (b) the plain projection signatures re-timed on the tiles whose epilogue changed this round (hoist back on the
# 4-wave BK64 tiles, two-chunk 64x64 form) -- the incumbent defends its entry; (c) VAE bench before / after.
mkdir -p gpurun_out
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_before.json
timeout 300 python tools/vae_bench.py 4 512 2>&1 | grep -E "encode|gemm|groupnorm" | tee gpurun_out/r04_vae_bench_before.txt
( timeout 900 python tools/tune_gemm.py --vae 4 --res 512 --cfgs 1,2,7,8,21,22,31,41,42,71,72,73,74,75,76 ) > gpurun_out/r04_tune_vae4.log 2>&1
tail -3 gpurun_out/r04_tune_vae4.log
( timeout 900 python tools/tune_gemm.py --vae 8 --res 512 --cfgs 1,2,7,8,21,22,31,41,42,71,72,73,74,75,76 ) > gpurun_out/r04_tune_vae8.log 2>&1
tail -2 gpurun_out/r04_tune_vae8.log
timeout 300 python tools/vae_bench.py 4 512 2>&1 | grep -E "encode|gemm|groupnorm" | tee gpurun_out/r04_vae_bench_after.txt
( timeout 1500 python tools/tune_gemm.py --merge --plain-only --cfgs 21,22,23,26,41,42,43,3,33,51,52,54,55 ) > gpurun_out/r04_tune_plain.log 2>&1
tail -3 gpurun_out/r04_tune_plain.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/r04_table_after.json
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for tbl in before after before after; do
  CLORA_GEMM_TUNING_FILE=$PWD/gpurun_out/r04_table_$tbl.json timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB table $tbl', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_table.txt
done
