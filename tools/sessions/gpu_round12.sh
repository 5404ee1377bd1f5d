#!/bin/bash
# gpurun call 12: upsampled-conv patch path (fresh build), tune its signatures, A/B of the stacked adapter downs (train + DDIM on one box)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv_patch or lora" -p no:cacheprovider ) > gpurun_out/gputest_k2.log 2>&1
tail -3 gpurun_out/gputest_k2.log
( time timeout 900 python tools/tune_gemm.py --cfgs 71,72,73,74,75,76 --patch-only --merge ) > gpurun_out/tune_patch_ups.log 2>&1
grep "s1e0" gpurun_out/tune_patch_ups.log | cut -c1-200; tail -2 gpurun_out/tune_patch_ups.log
( time timeout 600 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --cfgs 71,72,73,74,75,76 --patch-only --merge ) > gpurun_out/tune_patch_ups_v2.log 2>&1
tail -2 gpurun_out/tune_patch_ups_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( CLORA_MERGE_DOWNS=0 timeout 900 python bench.py $B ) > gpurun_out/bench_r12_nomerge.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r12_merge.log 2>&1
( CLORA_MERGE_DOWNS=0 timeout 900 python bench.py $B ) > gpurun_out/bench_r12_nomerge2.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r12_merge2.log 2>&1
for f in gpurun_out/bench_r12_nomerge.log gpurun_out/bench_r12_merge.log gpurun_out/bench_r12_nomerge2.log gpurun_out/bench_r12_merge2.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
