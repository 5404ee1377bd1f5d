#!/bin/bash
# Round 3, GPU call 6: re-tune the plain (non-conv) GEMM signatures of the train step against the 8-wave tiles now that their epilogue is
# two-phase (the incumbent entry defends itself), same-box A/B of the old vs new table, kernel tests with the outlier / determinism guards.
set -x
mkdir -p gpurun_out
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_before.json
( time timeout 900 python tools/tune_gemm.py --merge --cfgs 51,52,53,54,55,56 --plain-only --infer-batch 0 ) > gpurun_out/r03_tune_wide_two_phase.log 2>&1
tail -4 gpurun_out/r03_tune_wide_two_phase.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_after.json
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for t in before after before after; do
  CLORA_GEMM_TUNING_FILE=$PWD/gpurun_out/gemm_tuning_$t.json timeout 600 $B >> gpurun_out/r03_bench6_$t.json 2>> gpurun_out/r03_bench_ab6.err
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench6_before.json gpurun_out/r03_bench6_after.json
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "without_rowadd or row_segments or tile_configs" ) > gpurun_out/r03_gputest_6.log 2>&1
tail -3 gpurun_out/r03_gputest_6.log
