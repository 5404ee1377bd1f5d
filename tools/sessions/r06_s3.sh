#!/bin/bash
# round 6, gpurun call 3: deferred-finish policy A/B (fold only at <= 4 rows per thread / everywhere / never), the fp16-regime floors of
# the train-step quantities, the in-graph exchange at one rank, the vendor yardstick
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -k "exchange or fixture or graph" ) > gpurun_out/gputest_s3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s3.log
grep -E "FULL_SIZE|passed|failed|rc=|Error|WARNING" gpurun_out/gputest_s3.log | cut -c1-1500 | tail -16
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  timeout 600 python bench.py $B > gpurun_out/ab3_rows4_$i.log 2>&1
  CLORA_DEFER_FINISH=0 timeout 600 python bench.py $B > gpurun_out/ab3_nodefer_$i.log 2>&1
  CLORA_DEFER_MAX_ROWS=16 timeout 600 python bench.py $B > gpurun_out/ab3_rows16_$i.log 2>&1
  CLORA_DEFER_MAX_ROWS=8 timeout 600 python bench.py $B > gpurun_out/ab3_rows8_$i.log 2>&1
done
for f in gpurun_out/ab3_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1); done
( timeout 900 python tools/vendor_yardstick.py ) > gpurun_out/r06_vendor_yardstick.txt 2>&1
tail -60 gpurun_out/r06_vendor_yardstick.txt
