#!/bin/bash
# Round 3, GPU call 1: GPU suite (incl. the new full-size fixture tests and the negative-logit attention cases), the error
# budget of the 1e-3 latent tolerance, pending A/Bs of round 2 (tile order, single-pass epilogue staging), a full bench line.
set -x
mkdir -p gpurun_out
python tools/box_calib.py > gpurun_out/r03_box_calib.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r03_gputest_1.log 2>&1
tail -3 gpurun_out/r03_gputest_1.log
grep -h "FULL_SIZE\|DDIM_LATENT" gpurun_out/r03_gputest_1.log
timeout 600 python tools/error_budget.py gpurun_out/r03_error_budget.json > gpurun_out/r03_error_budget.log 2>&1
tail -2 gpurun_out/r03_error_budget.log
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for order in m auto m auto; do
  CLORA_TILE_ORDER=$order timeout 600 $B >> gpurun_out/r03_bench_order_${order}.json 2>> gpurun_out/r03_bench_ab.err
done
for lib in controllora_amd/_build_variant/libclora.so "" controllora_amd/_build_variant/libclora.so ""; do
  CLORA_LIB_PATH=$lib timeout 600 $B >> gpurun_out/r03_bench_epi_$( [ -n "$lib" ] && echo single || echo head ).json 2>> gpurun_out/r03_bench_ab.err
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench_order_*.json gpurun_out/r03_bench_epi_*.json
timeout 900 python bench.py --trace-out gpurun_out/r03_kernel_stats_1.json > gpurun_out/r03_bench_1.json 2> gpurun_out/r03_bench_1.err
cat gpurun_out/r03_bench_1.json | head -c 1500
