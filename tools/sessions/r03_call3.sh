#!/bin/bash
# Round 3, GPU call 3: the whole GPU suite at HEAD (CLIP, VAE at SD-1.5 widths, single-rank RCCL through the C ABI, fixture tests with
# final tolerances), and the DDIM-50 A/B of the two defaults that changed this round (tile_order auto, two-phase epilogue).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r03_gputest_3.log 2>&1
tail -3 gpurun_out/r03_gputest_3.log
grep -h "FULL_SIZE\|CLIP_\|VAE_SD15" gpurun_out/r03_gputest_3.log | cut -c1-400
D="python bench.py --no-cpu-baseline --no-roofline --no-full-step --steps 3 --warmup 1"
for v in "m 1" "auto 1" "m 0" "auto 0" "m 1" "auto 1"; do
  set -- $v
  CLORA_TILE_ORDER=$1 CLORA_EPI_TWO_PHASE=$2 timeout 600 $D > gpurun_out/tmp_ddim.json 2>> gpurun_out/r03_ddim_ab.err
  echo "order=$1 two_phase=$2 $(grep -o '"latency_s": [0-9.]*' gpurun_out/tmp_ddim.json) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tmp_ddim.json | head -1)" | tee -a gpurun_out/r03_ddim_ab.txt
done
