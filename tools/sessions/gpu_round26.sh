#!/bin/bash
# gpurun call 26: GroupNorm launch width A/B (256 / 512 / 1024 / 2048 blocks)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline --no-ddim"
for nb in 512 256 1024 2048 512; do ( CLORA_GN_BLOCKS=$nb timeout 900 python bench.py $B ) > gpurun_out/bench_r26_$nb.log 2>&1; echo "GN blocks $nb: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_r26_$nb.log | head -1)"; done
