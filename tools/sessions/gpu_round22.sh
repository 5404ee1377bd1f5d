#!/bin/bash
# gpurun call 22: backward attention with 8-wave blocks: A/B by env, tests, bench A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( CLORA_ATTN_BWD_WAVES=4 timeout 300 python tools/attn_ab.py gpurun_out/attn_bwd_w4.json ) > gpurun_out/attn_bwd_w4.log 2>&1
( CLORA_ATTN_BWD_WAVES=8 timeout 300 python tools/attn_ab.py gpurun_out/attn_bwd_w8.json ) > gpurun_out/attn_bwd_w8.log 2>&1
paste <(grep -o '"us": [0-9.]*' gpurun_out/attn_bwd_w4.log) <(grep -o '"kernel": "[^"]*", "us": [0-9.]*' gpurun_out/attn_bwd_w8.log)
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -p no:cacheprovider ) 2>&1 | tail -1
( CLORA_ATTN_BWD_WAVES=8 CLORA_ATTN_FWD_WAVES=8 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -p no:cacheprovider ) 2>&1 | tail -1
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( CLORA_ATTN_BWD_WAVES=4 CLORA_ATTN_FWD_WAVES=4 timeout 900 python bench.py $B ) > gpurun_out/bench_r22_w4.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r22_default.log 2>&1
for f in gpurun_out/bench_r22_w4.log gpurun_out/bench_r22_default.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
