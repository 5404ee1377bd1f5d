#!/bin/bash
# gpurun call 29: configs[3] shapes on the 256-row wide tiles; configs[3] bench before / after
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
V="--config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --no-roofline --steps 20"
( timeout 300 python bench.py $V ) > gpurun_out/bench_r29_v2_before.log 2>&1
( time timeout 400 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --cfgs 57,58 --plain-only --merge ) > gpurun_out/tune_256_v2.log 2>&1
grep "best tile=5[78]" gpurun_out/tune_256_v2.log | cut -c1-190; tail -1 gpurun_out/tune_256_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( timeout 300 python bench.py $V ) > gpurun_out/bench_r29_v2_after.log 2>&1
for f in gpurun_out/bench_r29_v2_before.log gpurun_out/bench_r29_v2_after.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
