#!/bin/bash
# Round 3, GPU call 10: knob sweeps that were last measured BEFORE the loads-in-flight rewrites of the norm kernels (end of round 2):
# GroupNorm row-chunk blocks in flight, LayerNorm rows per wave.  Same box, bench line only.
set -x
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for v in "512 1" "256 1" "384 1" "768 1" "1024 1" "512 0" "512 1"; do
  set -- $v
  CLORA_GN_BLOCKS=$1 CLORA_LN_ROWS=$2 timeout 300 $B > gpurun_out/tmp_b10.json 2>> gpurun_out/r03_bench_ab10.err
  echo "gn_blocks=$1 ln_rows=$2 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tmp_b10.json)" | tee -a gpurun_out/r03_bench_ab10.txt
done
