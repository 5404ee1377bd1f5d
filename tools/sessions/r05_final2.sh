#!/bin/bash
# Round 5, closing check at HEAD (after the opt-in 16-wave attention block joined the attention translation unit): GPU suite, smoke, bench line.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r05_gputest_final3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_gputest_final3.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_gputest_final3.log | tail -5
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee gpurun_out/r05_smoke_final3.txt
( time timeout 1200 python bench.py --trace-out gpurun_out/r05_kernel_stats_final3.json ) > gpurun_out/r05_bench_final3.log 2>&1
grep '^{' gpurun_out/r05_bench_final3.log > gpurun_out/r05_bench_final3.json
head -c 300 gpurun_out/r05_bench_final3.json; echo
