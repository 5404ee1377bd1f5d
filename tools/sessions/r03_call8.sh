#!/bin/bash
# Round 3, GPU call 8 (last): lora_down mode 1 as the default (K-split everywhere, 16 waves per row group up to 1024 rows): kernel + end-to-end
# parity, micro-benchmark, same-box A/B, then the complete bench line at HEAD (validates the roofline leg fix; leg timings on stderr).
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_full_topology_gpu.py -m gpu -x -q -s -k "lora or golden or identity or config1_train_step or chain" ) > gpurun_out/r03_gputest_8.log 2>&1
grep -E "passed|failed" gpurun_out/r03_gputest_8.log; grep -h "FULL_SIZE_TRAIN" gpurun_out/r03_gputest_8.log | cut -c1-300
timeout 200 python tools/lora_down_ab.py > gpurun_out/r03_lora_down_ab2.txt 2>&1; grep shape gpurun_out/r03_lora_down_ab2.txt
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for m in 0 1 0 1; do
  CLORA_LORA_DOWN_MODE=$m timeout 600 $B > gpurun_out/tmp_b8.json 2>> gpurun_out/r03_bench_ab8.err
  echo "lora_down_mode=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tmp_b8.json)" | tee -a gpurun_out/r03_bench_ab8.txt
done
( time timeout 900 python bench.py --trace-out gpurun_out/r03_kernel_stats_final2.json ) > gpurun_out/r03_bench_final2.log 2>&1
grep '^{' gpurun_out/r03_bench_final2.log > gpurun_out/r03_bench_final2.json
grep "bench legs" gpurun_out/r03_bench_final2.log; head -c 300 gpurun_out/r03_bench_final2.json; echo; grep -o '"roofline": {"bound": "mfma", "kernel": "[a-z_0-9]*", "achieved": [0-9.]*, "peak": [0-9.]*, "unit": "TFLOP/s", "frac": [0-9.]*' gpurun_out/r03_bench_final2.json
