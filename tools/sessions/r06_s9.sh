#!/bin/bash
# round 6, gpurun call 9: patch-staged weight-gradient kernel of the hint encoder's large-map convolutions -- parity on hardware,
# per-layer timings old / new, same-box A/B of the train step
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "wgrad or conv_fwd_dgrad or padded_channels" ) > gpurun_out/gputest_s9.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s9.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s9.log | cut -c1-300 | tail -8
CLORA_WGRAD_PATCH=0 timeout 600 python tools/wgrad_bench.py > gpurun_out/r06_wgrad_bench_old.txt 2>&1
timeout 600 python tools/wgrad_bench.py > gpurun_out/r06_wgrad_bench_new.txt 2>&1
paste -d'\n' gpurun_out/r06_wgrad_bench_old.txt gpurun_out/r06_wgrad_bench_new.txt | cut -c1-160
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2 3; do
  CLORA_WGRAD_PATCH=0 timeout 600 python bench.py $B > gpurun_out/ab9_old_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab9_new_$i.log 2>&1
done
for f in gpurun_out/ab9_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -p no:cacheprovider -x -k "train_step or golden or graph_replay or properties" ) > gpurun_out/gputest_s9b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s9b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s9b.log | cut -c1-300 | tail -6
