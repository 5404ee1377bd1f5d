#!/bin/bash
# round 6, gpurun call 20: train step with the compensated trunk in the training forward too (CLORA_TRUNK_LO=always) against the default, same box, HEAD
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-full-step --no-pmc --no-roofline --no-ddim --steps 30"
for i in 1 2 3; do
  timeout 600 python bench.py $B > gpurun_out/ab20_infer_$i.log 2>&1
  CLORA_TRUNK_LO=always timeout 600 python bench.py $B > gpurun_out/ab20_always_$i.log 2>&1
done
for f in gpurun_out/ab20_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
