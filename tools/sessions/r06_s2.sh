#!/bin/bash
# round 6, gpurun call 2: full GPU suite with the merged launches (deferred split-K finishes, in-place concat, grouped text K|V,
# multi-job hint-encoder pack / unpack) + same-box A/B against the round-5 host path (every switch off) + kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > gpurun_out/gputest_s2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s2.log
grep -E "passed|failed|rc=|Error|real" gpurun_out/gputest_s2.log | cut -c1-300 | tail -12
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
OLD="CLORA_GROUP_TEXT_KV=0 CLORA_PERSISTENT_CONV_PACKS=0 CLORA_DEFER_UNPACK=0 CLORA_DEFER_FINISH=0 CLORA_CAT_IN_PLACE=0"
for i in 1 2; do
  env $OLD timeout 600 python bench.py $B > gpurun_out/ab2_old_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab2_new_$i.log 2>&1
  CLORA_DEFER_FINISH=0 timeout 600 python bench.py $B > gpurun_out/ab2_nodefer_$i.log 2>&1
  CLORA_CAT_IN_PLACE=0 timeout 600 python bench.py $B > gpurun_out/ab2_nocat_$i.log 2>&1
done
for f in gpurun_out/ab2_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1); done
( timeout 900 python bench.py --no-cpu-baseline --no-ddim --no-full-step --steps 30 --trace-out gpurun_out/r06_kernel_stats_s2.json ) > gpurun_out/bench_s2.log 2>&1
tail -1 gpurun_out/bench_s2.log | cut -c1-600
