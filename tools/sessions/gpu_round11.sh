#!/bin/bash
# gpurun call 11: upsampled-conv patch path + stacked adapter downs: GPU tests, tune the upsample signatures, bench, inference profile by shape
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv_patch or lora" -p no:cacheprovider ) > gpurun_out/gputest_k2.log 2>&1
tail -3 gpurun_out/gputest_k2.log
( time timeout 900 python tools/tune_gemm.py --cfgs 71,72,73,74,75,76 --patch-only --merge ) > gpurun_out/tune_patch_ups.log 2>&1
grep "m1k1s1" gpurun_out/tune_patch_ups.log | cut -c1-200; tail -2 gpurun_out/tune_patch_ups.log
( time timeout 600 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --cfgs 71,72,73,74,75,76 --patch-only --merge ) > gpurun_out/tune_patch_ups_v2.log 2>&1
tail -2 gpurun_out/tune_patch_ups_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
B="--no-cpu-baseline --no-full-step --steps 30"
( timeout 900 python bench.py $B --trace-out gpurun_out/kt_r11.json ) > gpurun_out/bench_r11.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_r11.log | head -1; grep -o '"latency_s": [0-9.]*' gpurun_out/bench_r11.log; grep -o '"frac": [0-9.]*' gpurun_out/bench_r11.log | head -1
( timeout 300 python tools/infer_profile.py ) > gpurun_out/infer_profile.log 2>&1
grep -A62 "== one UNet" gpurun_out/infer_profile.log | cut -c1-150
( time timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_full_topology_gpu.py -q -x -s -p no:cacheprovider ) > gpurun_out/gputest_e2e.log 2>&1
grep -E "passed|failed|FULL_TOPOLOGY|DDIM_LATENT" gpurun_out/gputest_e2e.log | cut -c1-260
