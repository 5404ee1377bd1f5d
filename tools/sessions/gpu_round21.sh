#!/bin/bash
# gpurun call 21: forward attention with 4 / 6 / 8 waves per block (K/V stream per flop)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
for nw in 4 6 8; do ( CLORA_ATTN_FWD_WAVES=$nw timeout 300 python tools/attn_ab.py gpurun_out/attn_fwd_w$nw.json ) > gpurun_out/attn_fwd_w$nw.log 2>&1; done
paste <(grep -o '"us": [0-9.]*' gpurun_out/attn_fwd_w4.log) <(grep -o '"us": [0-9.]*' gpurun_out/attn_fwd_w6.log) <(grep -o '"kernel": "[^"]*", "us": [0-9.]*' gpurun_out/attn_fwd_w8.log) | grep fwd
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -p no:cacheprovider ) 2>&1 | tail -1
( CLORA_ATTN_FWD_WAVES=8 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -p no:cacheprovider ) 2>&1 | tail -1
