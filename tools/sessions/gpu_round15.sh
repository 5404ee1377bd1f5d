#!/bin/bash
# gpurun call 15: epilogue-aware re-tune of the plain GEMM signatures over the full variant set (train bs4 + inference bs32, then configs[3]), bench A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/box_calib.txt
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( timeout 900 python bench.py $B ) > gpurun_out/bench_r15_before.log 2>&1
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_before_r15.json
( time timeout 1500 python tools/tune_gemm.py --plain-only --merge ) > gpurun_out/tune_plain_full.log 2>&1
tail -3 gpurun_out/tune_plain_full.log
( time timeout 900 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --plain-only --merge ) > gpurun_out/tune_plain_full_v2.log 2>&1
tail -2 gpurun_out/tune_plain_full_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( timeout 900 python bench.py $B ) > gpurun_out/bench_r15_after.log 2>&1
for f in gpurun_out/bench_r15_before.log gpurun_out/bench_r15_after.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
