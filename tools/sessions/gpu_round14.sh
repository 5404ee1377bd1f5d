#!/bin/bash
# gpurun call 14: adapter epilogue with the up matrix hoisted: GPU tests, same-box A/B (previous build via CLORA_LIB_PATH), trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/box_calib.txt
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or geglu" -p no:cacheprovider ) > gpurun_out/gputest_k4.log 2>&1
tail -3 gpurun_out/gputest_k4.log
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r14_prev.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r14_new.log 2>&1
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r14_prev2.log 2>&1
( timeout 900 python bench.py --no-cpu-baseline --no-full-step --steps 30 --trace-out gpurun_out/kt_r14.json ) > gpurun_out/bench_r14_new2.log 2>&1
for f in gpurun_out/bench_r14_prev.log gpurun_out/bench_r14_new.log gpurun_out/bench_r14_prev2.log gpurun_out/bench_r14_new2.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
