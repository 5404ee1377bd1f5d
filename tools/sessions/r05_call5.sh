#!/bin/bash
# Round 5, call 5: tile 59 offered to the tuner (plain + GEGLU signatures, train batch 4 and inference batch 32); per-layer error budget;
# BASELINE-adjacent config danbooru-sketch.json bench line; the full-topology GPU tests under the new limits + the DPM validation loop.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python tools/tune_gemm.py --merge --plain-only --cfgs 59 ) > gpurun_out/r05_tune_tile59_plain.log 2>&1
grep -c "best tile=59" gpurun_out/r05_tune_tile59_plain.log; grep "best tile=59" gpurun_out/r05_tune_tile59_plain.log | head -20; tail -4 gpurun_out/r05_tune_tile59_plain.log
( time timeout 400 python tools/tune_gemm.py --geglu --cfgs 59 ) > gpurun_out/r05_tune_tile59_geglu.log 2>&1
grep "best tile" gpurun_out/r05_tune_tile59_geglu.log | head -20
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_after_tile59.json
timeout 300 python tools/error_budget_layers.py gpurun_out/r05_error_budget_layers.json 2>&1 | tail -52 | tee gpurun_out/r05_error_budget_layers.txt
( time timeout 600 python bench.py --config danbooru-sketch.json --no-ddim --no-cpu-baseline --no-full-step --no-pmc --trace-out gpurun_out/r05_kernel_stats_sketch.json ) > gpurun_out/r05_bench_sketch.log 2>&1
grep '^{' gpurun_out/r05_bench_sketch.log > gpurun_out/r05_bench_sketch.json; head -c 400 gpurun_out/r05_bench_sketch.json; echo
( time timeout 900 python -m pytest tests/test_full_topology_gpu.py -q -x -s -p no:cacheprovider ) > gpurun_out/r05_gputest_full_topology.log 2>&1
grep -E "passed|failed|NOTE|VALIDATION_DPM|^FAILED|^ERROR" gpurun_out/r05_gputest_full_topology.log | cut -c1-300 | tail -12
