#!/bin/bash
# gpurun call 3: GPU tests, GEMM autotune over the new variant set (train bs4 + inference bs32, then v2 bs8), benches
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gputest.log | tail -20
( time timeout 1500 python tools/tune_gemm.py ) > gpurun_out/tune_a.log 2>&1
tail -3 gpurun_out/tune_a.log
( time timeout 900 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --merge ) > gpurun_out/tune_b.log 2>&1
tail -3 gpurun_out/tune_b.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1
tail -c 1500 gpurun_out/bench.log
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step ) > gpurun_out/bench_v2.log 2>&1
tail -c 600 gpurun_out/bench_v2.log
