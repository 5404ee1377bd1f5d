#!/bin/bash
# round 6, gpurun call 10: patch-staged weight gradients incl. the stride-2 downsamplers, automatic block count -- parity, sweep, step A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "wgrad or conv_fwd_dgrad or padded_channels" ) > gpurun_out/gputest_s10.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s10.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s10.log | cut -c1-300 | tail -8
timeout 600 python tools/wgrad_patch_sweep.py 2>&1 | tee gpurun_out/r06_wgrad_patch_sweep2.txt
timeout 600 python tools/wgrad_bench.py 2>&1 | tail -3
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2 3; do
  CLORA_WGRAD_PATCH=0 timeout 600 python bench.py $B > gpurun_out/ab10_old_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab10_new_$i.log 2>&1
done
for f in gpurun_out/ab10_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -x -k "train_step or golden or graph_replay or properties" ) > gpurun_out/gputest_s10b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s10b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s10b.log | cut -c1-300 | tail -6
