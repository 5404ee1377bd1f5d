#!/bin/bash
# gpurun call 8: patch-staged 3x3 conv kernel: GPU tests, sweep against the tuned incumbents (train bs4 + inference bs32), bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv_patch" -p no:cacheprovider ) > gpurun_out/gputest_patch.log 2>&1
tail -5 gpurun_out/gputest_patch.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_before_patch.json
( time timeout 1200 python tools/tune_gemm.py --cfgs 71,72,73,74,75 --patch-only --merge ) > gpurun_out/tune_patch.log 2>&1
grep -v "^/opt\|Warning" gpurun_out/tune_patch.log | tail -70
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
B="--no-cpu-baseline --no-full-step --steps 30"
( timeout 900 python bench.py $B --trace-out gpurun_out/kt_patch.json ) > gpurun_out/bench_patch.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_patch.log | head -1; grep -o '"latency_s": [0-9.]*' gpurun_out/bench_patch.log; grep -o '"frac": [0-9.]*' gpurun_out/bench_patch.log | head -1
