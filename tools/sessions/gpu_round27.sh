#!/bin/bash
# gpurun call 27: closing refresh at HEAD: all GPU tests, bench line + kernel trace, configs[3] line
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/box_calib_final.txt
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_final.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gputest_final.log | tail -6
( time timeout 900 python bench.py --trace-out gpurun_out/r02_kernel_stats_final.json ) > gpurun_out/bench_final.log 2>&1
grep '^{' gpurun_out/bench_final.log > gpurun_out/r02_bench_final.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_final.log | head -1; grep -o '"latency_s": [0-9.]*' gpurun_out/bench_final.log; grep -o '"frac": [0-9.]*' gpurun_out/bench_final.log | head -1
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --trace-out gpurun_out/r02_kernel_stats_v2_final.json ) > gpurun_out/bench_v2_final.log 2>&1
grep '^{' gpurun_out/bench_v2_final.log > gpurun_out/r02_bench_v2_final.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_v2_final.log | head -1
python __graft_entry__.py --smoke 2>&1 | tail -1
