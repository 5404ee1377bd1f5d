#!/bin/bash
# round 6, gpurun call 21: per-(kernel, grid) trace of the batch-32 DDIM loop at HEAD with the compensated trunk off and on
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in off infer; do
  cd /tmp; rm -rf /tmp/ddimkt_$mode
  CLORA_TRUNK_LO=$mode timeout 600 rocprofv3 --kernel-trace -d /tmp/ddimkt_$mode -o kt -- python $R/tools/ddim_profile.py 4 > $R/gpurun_out/ddimkt_$mode.log 2>&1
  cd $R
  python tools/trace_by_grid.py $(find /tmp/ddimkt_$mode -name "*.db" | head -1) gpurun_out/r06_ddim_trace_by_grid_trunk_$mode.txt 6 60 > /dev/null 2>&1
  head -12 gpurun_out/r06_ddim_trace_by_grid_trunk_$mode.txt | cut -c1-170
  tail -2 gpurun_out/ddimkt_$mode.log
done
