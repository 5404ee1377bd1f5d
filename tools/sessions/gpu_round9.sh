#!/bin/bash
# gpurun call 9: patch-kernel GPU tests (all variants), 128x160 variant vs incumbents, configs[3] shapes, bench lines + traces
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv_patch or attention" -p no:cacheprovider ) > gpurun_out/gputest_patch.log 2>&1
tail -5 gpurun_out/gputest_patch.log
( time timeout 900 python tools/tune_gemm.py --cfgs 76 --patch-only --merge ) > gpurun_out/tune_patch76.log 2>&1
grep "best tile" gpurun_out/tune_patch76.log | cut -c1-200 | head -60; tail -2 gpurun_out/tune_patch76.log
( time timeout 900 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --cfgs 71,72,73,74,75,76 --patch-only --merge ) > gpurun_out/tune_patch_v2.log 2>&1
tail -2 gpurun_out/tune_patch_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
B="--no-cpu-baseline --no-full-step --steps 30"
( timeout 900 python bench.py $B --trace-out gpurun_out/kt_patch2.json ) > gpurun_out/bench_patch2.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_patch2.log | head -1; grep -o '"latency_s": [0-9.]*' gpurun_out/bench_patch2.log; grep -o '"frac": [0-9.]*' gpurun_out/bench_patch2.log | head -1
( timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --steps 20 ) > gpurun_out/bench_patch2_v2.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_patch2_v2.log | head -1
