#!/bin/bash
# round 6, gpurun call 1: parity subset with the tightened limits + same-box A/B of the round-5 host path (env switches) against
# the merged launches (conv pack / unpack multi-job, grouped text K|V)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -x ) > gpurun_out/gputest_s1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s1.log
grep -E "FULL_SIZE|NOTE|passed|failed|rc=|Error" gpurun_out/gputest_s1.log | cut -c1-400 | tail -30
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  CLORA_GROUP_TEXT_KV=0 CLORA_PERSISTENT_CONV_PACKS=0 CLORA_DEFER_UNPACK=0 timeout 600 python bench.py $B > gpurun_out/ab_old_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab_new_$i.log 2>&1
done
for f in gpurun_out/ab_*.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1; done
( timeout 900 python bench.py --no-cpu-baseline --no-ddim --no-full-step --steps 30 --trace-out gpurun_out/r06_kernel_stats_s1.json ) > gpurun_out/bench_s1.log 2>&1
tail -1 gpurun_out/bench_s1.log | cut -c1-1500
