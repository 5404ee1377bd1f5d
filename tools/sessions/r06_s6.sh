#!/bin/bash
# round 6, gpurun call 6: whole-chip one-launch GroupNorm (in-kernel barrier per batch element) -- parity on hardware, per-shape timings,
# same-box A/B of the train step
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "groupnorm or deferred or bit_stable" ) > gpurun_out/gputest_s6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s6.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s6.log | cut -c1-300 | tail -14
timeout 600 python tools/gn_bench.py > gpurun_out/r06_gn_bench.txt 2>&1
cat gpurun_out/r06_gn_bench.txt | cut -c1-200
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  CLORA_GN_SYNC=0 timeout 600 python bench.py $B > gpurun_out/ab6_sync0_$i.log 2>&1
  CLORA_GN_SYNC=1 timeout 600 python bench.py $B > gpurun_out/ab6_sync1_$i.log 2>&1
  CLORA_GN_SYNC=2 timeout 600 python bench.py $B > gpurun_out/ab6_sync2_$i.log 2>&1
done
for f in gpurun_out/ab6_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -s -p no:cacheprovider -x ) > gpurun_out/gputest_s6b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s6b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s6b.log | cut -c1-300 | tail -6
