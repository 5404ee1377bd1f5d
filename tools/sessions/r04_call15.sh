#!/bin/bash
# Round 4, call 15: where the one-launch GroupNorm's time goes -- per-(kernel, grid) trace of the train step with the option on
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
CLORA_GN_RESIDENT=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 8 --warmup 2 > $R/gpurun_out/stepkt.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r04_step_trace_by_grid_gn_resident.txt 12 200 > /dev/null 2>&1
grep -E "gn_|total" gpurun_out/r04_step_trace_by_grid_gn_resident.txt | cut -c1-150
