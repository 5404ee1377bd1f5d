#!/bin/bash
# round 6, gpurun call 18: DDIM-50 leg with the compensated trunk off / on at HEAD (after the unconsumed remainders were dropped), same box
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-full-step --no-pmc --no-roofline --steps 10 --windows 1"
for i in 1 2; do
  CLORA_TRUNK_LO=off timeout 600 python bench.py $B > gpurun_out/ab18_off_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab18_infer_$i.log 2>&1
done
for f in gpurun_out/ab18_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"latency_s": [0-9.]*' $f | head -1); done
