#!/bin/bash
# gpurun call 20: unrolled partial folds (GroupNorm apply / params, split-K finish): tests + same-box A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "norm or gemm_plain or tile_configs or conv" -p no:cacheprovider ) > gpurun_out/gputest_k5.log 2>&1
tail -2 gpurun_out/gputest_k5.log
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r20_prev.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r20_new.log 2>&1
for f in gpurun_out/bench_r20_prev.log gpurun_out/bench_r20_new.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
