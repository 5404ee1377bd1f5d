#!/bin/bash
# Round 5, call 9: kernel trace of the batch-32 DDIM loop (BASELINE config 5) at HEAD: where tiles 59 / 79 sit in the sampler.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/ddimkt
timeout 600 rocprofv3 --kernel-trace -d /tmp/ddimkt -o kt -- python $R/tools/ddim_profile.py 4 > $R/gpurun_out/ddimkt.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/ddimkt -name "*.db" | head -1) gpurun_out/r05_ddim_trace_by_grid.txt 6 60 > /dev/null 2>&1
head -40 gpurun_out/r05_ddim_trace_by_grid.txt | cut -c1-160
tail -3 gpurun_out/ddimkt.log
