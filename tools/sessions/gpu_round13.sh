#!/bin/bash
# gpurun call 13: 8-wave wide tiles (128x320 / 64x320 / 128x256) for the short-K projections: GPU tests, tune the plain signatures, bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/box_calib.txt
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "tile_configs or geglu or fast_path or gemm" -p no:cacheprovider ) > gpurun_out/gputest_k3.log 2>&1
tail -3 gpurun_out/gputest_k3.log
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline"
( timeout 900 python bench.py $B ) > gpurun_out/bench_r13_before.log 2>&1
( time timeout 1200 python tools/tune_gemm.py --cfgs 51,52,53,54,55,56 --plain-only --merge ) > gpurun_out/tune_wide.log 2>&1
grep "best tile=5" gpurun_out/tune_wide.log | cut -c1-200; tail -2 gpurun_out/tune_wide.log
( time timeout 600 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --cfgs 51,52,53,54,55,56 --plain-only --merge ) > gpurun_out/tune_wide_v2.log 2>&1
tail -2 gpurun_out/tune_wide_v2.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( timeout 900 python bench.py $B ) > gpurun_out/bench_r13_after.log 2>&1
for f in gpurun_out/bench_r13_before.log gpurun_out/bench_r13_after.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
