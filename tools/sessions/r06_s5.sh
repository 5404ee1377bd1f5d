#!/bin/bash
# round 6, gpurun call 5: fused LayerNorm in the 320-column tiles -- kernel parity on hardware, full-size fixtures, same-box A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -k "fused_layernorm or fixture or concat_in_place or deferred" ) > gpurun_out/gputest_s5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s5.log
grep -E "FULL_SIZE|passed|failed|rc=|Error" gpurun_out/gputest_s5.log | cut -c1-700 | tail -14
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2 3; do
  CLORA_FUSE_LN=0 timeout 600 python bench.py $B > gpurun_out/ab5_noln_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab5_ln_$i.log 2>&1
done
for f in gpurun_out/ab5_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1); done
( timeout 900 python bench.py --no-cpu-baseline --no-ddim --no-full-step --steps 30 --trace-out gpurun_out/r06_kernel_stats_s5.json ) > gpurun_out/bench_s5.log 2>&1
tail -1 gpurun_out/bench_s5.log | cut -c1-400
