#!/bin/bash
# Round 3, GPU call 7: bytes-in-flight experiments on the streaming kernels (Little's law: the adapter down-projection at M = 16384 has one
# block per CU = 16 KB in flight = the 2.6 TB/s it measures): lora_down launch modes, GroupNorm rows in flight; micro-benchmarks, the
# bench-line A/B on one box, the kernel tests of the new options.
set -x
mkdir -p gpurun_out
timeout 300 python tools/lora_down_ab.py > gpurun_out/r03_lora_down_ab.txt 2>&1; cat gpurun_out/r03_lora_down_ab.txt | cut -c1-200
for u in 0 1; do CLORA_GN_UNROLL=$u timeout 200 python tools/gn_bench.py > gpurun_out/r03_gn_bench_u$u.txt 2>&1; done
paste -d'\n' gpurun_out/r03_gn_bench_u0.txt gpurun_out/r03_gn_bench_u1.txt | cut -c1-200 | head -40
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for v in "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $v
  CLORA_LORA_DOWN_MODE=$1 CLORA_GN_UNROLL=$2 timeout 600 $B > gpurun_out/tmp_b7.json 2>> gpurun_out/r03_bench_ab7.err
  echo "lora_down_mode=$1 gn_unroll=$2 $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tmp_b7.json)" | tee -a gpurun_out/r03_bench_ab7.txt
done
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "lora or groupnorm" ) > gpurun_out/r03_gputest_7.log 2>&1
CLORA_GN_UNROLL=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -k "groupnorm or identity or golden" >> gpurun_out/r03_gputest_7.log 2>&1
grep -E "passed|failed" gpurun_out/r03_gputest_7.log
