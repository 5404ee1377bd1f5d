#!/bin/bash
# gpurun call 25: larger row chunks in the batched adapter weight-gradient launches: tests + same-box A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -k "lora or golden or v1 or v2" -p no:cacheprovider ) 2>&1 | tail -1
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline --no-ddim"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r25_prev.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r25_new.log 2>&1
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r25_prev2.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r25_new2.log 2>&1
for f in gpurun_out/bench_r25_prev.log gpurun_out/bench_r25_new.log gpurun_out/bench_r25_prev2.log gpurun_out/bench_r25_new2.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
