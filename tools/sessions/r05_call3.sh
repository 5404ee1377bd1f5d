#!/bin/bash
# Round 5, call 3: one-wave-per-SIMD tile variants (61 / 62 / 81 / 82) -- parity on hardware + A/B; a fresh per-kernel trace of the step.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "(tile_configs and (61 or 62)) or (conv_patch_kernel and (81 or 82)) or (without_rowadd and (61 or 62))" ) 2>&1 | tail -5 | tee gpurun_out/r05_gputest_onewave.log
timeout 900 python tools/onewave_ab.py 5 2>&1 | tee gpurun_out/r05_onewave_ab.txt
cd /tmp; rm -rf /tmp/stepkt
timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --no-calibration --windows 1 --steps 8 --warmup 2 > $R/gpurun_out/stepkt.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r05_step_trace_by_grid_3.txt 12 140 > /dev/null 2>&1
head -5 gpurun_out/r05_step_trace_by_grid_3.txt
