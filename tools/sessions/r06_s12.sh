#!/bin/bash
# round 6, gpurun call 12 (with the 8-channel conv_in variant): strip kernel for the hint encoder's large-map forward / dgrad convolutions -- parity, timings, step A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "conv_strip or wgrad or conv_fwd_dgrad or padded_channels" ) > gpurun_out/gputest_s12.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s12.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s12.log | cut -c1-300 | tail -8
timeout 600 python tools/conv_strip_bench.py 2>&1 | tee gpurun_out/r06_conv_strip_bench.txt
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2 3; do
  CLORA_CONV_STRIP=0 timeout 600 python bench.py $B > gpurun_out/ab12_old_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab12_new_$i.log 2>&1
done
for f in gpurun_out/ab12_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -x -k "train_step or golden or graph_replay or properties or zero_init" ) > gpurun_out/gputest_s12b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s12b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s12b.log | cut -c1-300 | tail -6
