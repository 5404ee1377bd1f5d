#!/bin/bash
# gpurun call 23: LDS-staged adapter up matrix in the small-tile epilogues: tests + same-box A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or geglu or tile_configs" -p no:cacheprovider ) 2>&1 | tail -1
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r23_prev.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r23_new.log 2>&1
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B --no-ddim ) > gpurun_out/bench_r23_prev2.log 2>&1
( timeout 900 python bench.py $B --no-ddim ) > gpurun_out/bench_r23_new2.log 2>&1
for f in gpurun_out/bench_r23_prev.log gpurun_out/bench_r23_new.log gpurun_out/bench_r23_prev2.log gpurun_out/bench_r23_new2.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
