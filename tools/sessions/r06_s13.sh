#!/bin/bash
# round 6, gpurun call 13: GroupNorm team kernels (one launch for the 64x64 / 32x32 maps, in-launch exchange of partial sums) -- parity, timings, step A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "groupnorm or deferred or bit_stab or determin" ) > gpurun_out/gputest_s13.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s13.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s13.log | cut -c1-300 | tail -8
timeout 600 python tools/gn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gn_team_bench.txt
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  for m in 0 1 2; do
    CLORA_GN_TEAM=$m timeout 600 python bench.py $B > gpurun_out/ab13_team${m}_$i.log 2>&1
  done
done
for f in gpurun_out/ab13_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -x -k "train_step or golden or graph_replay or properties or zero_init" ) > gpurun_out/gputest_s13b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s13b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s13b.log | cut -c1-300 | tail -6
