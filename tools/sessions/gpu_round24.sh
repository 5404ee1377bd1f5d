#!/bin/bash
# gpurun call 24: 256x320 / 256x256 8-wave tiles for the M >= 32768 projections (batch-32 inference): tests, tune, DDIM before/after
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "tile_configs or geglu" -p no:cacheprovider ) 2>&1 | tail -1
B="--no-cpu-baseline --no-full-step --steps 20 --no-roofline"
( timeout 900 python bench.py $B ) > gpurun_out/bench_r24_before.log 2>&1
( time timeout 900 python tools/tune_gemm.py --cfgs 57,58 --plain-only --merge ) > gpurun_out/tune_256.log 2>&1
grep "best tile=5[78]" gpurun_out/tune_256.log | cut -c1-200; tail -1 gpurun_out/tune_256.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( timeout 900 python bench.py $B ) > gpurun_out/bench_r24_after.log 2>&1
for f in gpurun_out/bench_r24_before.log gpurun_out/bench_r24_after.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
