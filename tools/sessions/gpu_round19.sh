#!/bin/bash
# gpurun call 19: train-step kernel trace aggregated by (kernel, grid)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/stepkt && timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --trace-child --steps 6 --warmup 2 > $R/gpurun_out/step_trace.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r02_step_trace_by_grid.txt 10 140
head -100 gpurun_out/r02_step_trace_by_grid.txt | cut -c1-170
