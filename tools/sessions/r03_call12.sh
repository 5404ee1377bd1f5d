#!/bin/bash
# Round 3, GPU call 12 (last minutes): hoisted epilogue restricted to the one-block-per-CU tiles: strict epilogue test over every tile twice,
# run-to-run bit stability of the full-size forward, A/B against the previous build, the bench line at the final commit.
set -x
mkdir -p gpurun_out
for i in 1 2; do timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "without_rowadd" 2>&1 | tail -2; done | tee gpurun_out/r03_gputest_12.log
timeout 300 python -m pytest tests/test_full_topology_gpu.py -m gpu -q -s -k "full_size_properties" 2>&1 | grep -E "FULL_SIZE|passed|failed" | cut -c1-300 | tee -a gpurun_out/r03_gputest_12.log
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for lib in controllora_amd/_build_variant/libclora.so "" controllora_amd/_build_variant/libclora.so ""; do
  CLORA_LIB_PATH=$lib timeout 300 $B > gpurun_out/tmp_b12.json 2>> gpurun_out/r03_bench_ab12.err
  echo "lib=$( [ -n "$lib" ] && echo hoist_up_to_2_blocks_per_cu || echo hoist_8wave_only ) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tmp_b12.json)" | tee -a gpurun_out/r03_bench_ab12.txt
done
timeout 240 python bench.py --no-cpu-baseline --no-full-step --trace-out gpurun_out/r03_kernel_stats_final3.json > gpurun_out/r03_bench_final3.json 2> gpurun_out/r03_bench_final3.err
head -c 250 gpurun_out/r03_bench_final3.json
