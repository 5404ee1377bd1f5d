#!/bin/bash
# Round 4, call 11: kernel trace of the DDIM loop (UNet batch 32) by (kernel, grid): where the 40 ms per step go after this round.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/ddimkt
timeout 900 rocprofv3 --kernel-trace -d /tmp/ddimkt -o kt -- python $R/tools/ddim_profile.py 12 > $R/gpurun_out/r04_ddim_kt.log 2>&1
cd $R
tail -2 gpurun_out/r04_ddim_kt.log
python tools/trace_by_grid.py $(find /tmp/ddimkt -name "*.db" | head -1) gpurun_out/r04_ddim_trace_by_grid.txt 14 60 > /dev/null 2>&1
head -45 gpurun_out/r04_ddim_trace_by_grid.txt | cut -c1-150
