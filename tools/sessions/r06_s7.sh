#!/bin/bash
# round 6, gpurun call 7: split-K finished by the last block of every tile inside the GEMM launch (agent-scope coherent slab accesses, no
# L2-wide fences), timestep embedding in one launch, 32 adapter weight-gradient jobs per launch -- parity on hardware, same-box A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -x ) > gpurun_out/gputest_s7.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s7.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s7.log | cut -c1-300 | tail -8
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2 3; do
  CLORA_SPLITK_TAIL=0 timeout 600 python bench.py $B > gpurun_out/ab7_tail0_$i.log 2>&1
  CLORA_SPLITK_TAIL=1 timeout 600 python bench.py $B > gpurun_out/ab7_tail1_$i.log 2>&1
done
for f in gpurun_out/ab7_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( timeout 900 python bench.py --no-cpu-baseline --no-ddim --no-full-step --steps 30 --trace-out gpurun_out/r06_kernel_stats_s7.json ) > gpurun_out/bench_s7.log 2>&1
tail -1 gpurun_out/bench_s7.log | cut -c1-600
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_kernel_stats_s7.json'))
print('launches_per_step', d['launches_per_step'], 'kernel ms', d['total_kernel_ms_per_step'])
P
( time timeout 1500 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -s -p no:cacheprovider -x ) > gpurun_out/gputest_s7b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s7b.log
grep -E "passed|failed|rc=|Error" gpurun_out/gputest_s7b.log | cut -c1-300 | tail -6
