#!/bin/bash
# Next measurement session, first call: same-box A/B of the tile -> XCD assignment (CLORA_TILE_ORDER: m = current default,
# auto = fewest distinct operand panels per XCD; model in clora_gemm.hip pick_tile_order, motivation in DESIGN.md section 5
# "Hardware ceilings") per shape and on the bench line, after the GPU suite.
set -x
mkdir -p gpurun_out
python tools/box_calib.py > gpurun_out/r03_box_calib.txt 2>&1
# correctness of everything that changed without a GPU at the end of round 2 (GroupNorm / LayerNorm / wgrad / finish loops, order options)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest.log 2>&1
tail -3 gpurun_out/r03_gputest.log
timeout 600 python tools/tile_order_ab.py --json gpurun_out/r03_tile_order_ab.json > gpurun_out/r03_tile_order_ab.txt 2>&1
for order in m auto m auto; do
  CLORA_TILE_ORDER=$order timeout 600 python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5 \
    >> gpurun_out/r03_bench_order_${order}.json 2>> gpurun_out/r03_bench_order_${order}.err
done
# LayerNorm with several rows in flight per wave (default; CLORA_LN_ROWS=0 = the one-row kernel; clora_norm.hip layernorm_rows_kernel)
for rows in 0 1 0 1; do
  CLORA_LN_ROWS=$rows timeout 600 python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5 \
    >> gpurun_out/r03_bench_ln_rows_${rows}.json 2>> gpurun_out/r03_bench_ln_rows_${rows}.err
done
# GroupNorm / adapter-wgrad loops with loads in flight (HEAD) against the load -> wait -> use loops: build the "before" library
# in the dev container first:  tools/build_prev_lib.sh 62e4530:clora_norm.hip b00c262:clora_lora.hip
if [ -f controllora_amd/_build_prev/libclora.so ]; then
  for lib in controllora_amd/_build_prev/libclora.so "" controllora_amd/_build_prev/libclora.so ""; do
    CLORA_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5 \
      >> gpurun_out/r03_bench_lib_$( [ -n "$lib" ] && echo prev || echo head ).json 2>> gpurun_out/r03_bench_lib.err
  done
fi
# single-pass accumulator staging in the GEMM epilogue (experiment macro): build first with tools/build_variant_lib.sh -DCLORA_EPI_SINGLE_PASS
if [ -f controllora_amd/_build_variant/libclora.so ]; then
  for lib in controllora_amd/_build_variant/libclora.so "" controllora_amd/_build_variant/libclora.so ""; do
    CLORA_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5 \
      >> gpurun_out/r03_bench_epi_$( [ -n "$lib" ] && echo single || echo head ).json 2>> gpurun_out/r03_bench_epi.err
  done
fi
