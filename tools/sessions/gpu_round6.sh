#!/bin/bash
# gpurun call 6: MFMA shape issue rates, attention rework (lazy exponent reference, ones column, accumulator-initialised
# backward): GPU tests + same-box A/B against the previous build, by-shape step profile, bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/probes/_build/mfma_rate_probe > gpurun_out/mfma_rate_probe.txt 2>&1; cat gpurun_out/mfma_rate_probe.txt
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" -p no:cacheprovider ) > gpurun_out/gputest_attn.log 2>&1
tail -3 gpurun_out/gputest_attn.log
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 300 python tools/attn_ab.py gpurun_out/attn_prev.json ) > gpurun_out/attn_prev.log 2>&1
( timeout 300 python tools/attn_ab.py gpurun_out/attn_new.json ) > gpurun_out/attn_new.log 2>&1
paste <(grep -o '"us": [0-9.]*' gpurun_out/attn_prev.log) <(grep -o '"kernel": "[^"]*", "us": [0-9.]*' gpurun_out/attn_new.log)
( timeout 300 python tools/kbench.py gpurun_out/kbench.json ) > gpurun_out/kbench.log 2>&1
( timeout 600 python tools/step_profile.py ) > gpurun_out/step_profile_by_shape.log 2>&1
grep -A75 "by shape" gpurun_out/step_profile_by_shape.log | head -80
B="--no-cpu-baseline --no-full-step --steps 30"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 600 python bench.py $B --no-roofline --no-ddim ) > gpurun_out/bench_prev.log 2>&1
( timeout 900 python bench.py $B --trace-out gpurun_out/kt_attn.json ) > gpurun_out/bench_attn.log 2>&1
for f in gpurun_out/bench_prev.log gpurun_out/bench_attn.log; do grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
( time timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_full_topology_gpu.py -q -x -s -p no:cacheprovider ) > gpurun_out/gputest_e2e.log 2>&1
grep -E "passed|failed|FULL_TOPOLOGY|DDIM_LATENT" gpurun_out/gputest_e2e.log | cut -c1-300
