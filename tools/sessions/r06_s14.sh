#!/bin/bash
# round 6, gpurun call 14: GroupNorm team kernels -- graph-replay / changing-geometry stress test, the smaller maps (option levels 3 / 4), step A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "groupnorm or deferred or bit_stab" ) > gpurun_out/gputest_s14.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s14.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s14.log | cut -c1-300 | tail -8
timeout 600 python tools/gn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gn_team_bench2.txt
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  for m in 2 3 4; do
    CLORA_GN_TEAM=$m timeout 600 python bench.py $B > gpurun_out/ab14_team${m}_$i.log 2>&1
  done
done
for f in gpurun_out/ab14_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1) $(grep -o '"gn_team_errors": [0-9]*' $f | head -1); done
