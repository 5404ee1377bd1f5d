#!/bin/bash
# gpurun call 10: kernel trace of the DDIM loop (batch 16 x CFG = UNet batch 32), SQ counters of the patch kernel
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/ddimkt && timeout 600 rocprofv3 --kernel-trace -d /tmp/ddimkt -o kt -- python $R/tools/ddim_profile.py 4 > $R/gpurun_out/ddim_profile.log 2>&1
cd $R
python tools/rocprof_summary.py $(find /tmp/ddimkt -name "*.db" | head -1) gpurun_out/r02_ddim_kernel_stats 6 > gpurun_out/ddim_kernel_stats.txt 2>&1
cat gpurun_out/r02_ddim_kernel_stats.md | head -50 | cut -c1-200; tail -2 gpurun_out/ddim_profile.log
