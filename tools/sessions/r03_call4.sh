#!/bin/bash
# Round 3, GPU call 4: two-phase epilogue on the 64x64 tile + wide-patch conv variants (VAE): kernel tests, GEMM decomposition on / off,
# VAE encode / decode timing, same-box A/B of the bench line, full line.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -s -k "gemm or patch or vae or tile or golden or identity" ) > gpurun_out/r03_gputest_4.log 2>&1
tail -3 gpurun_out/r03_gputest_4.log; grep -h "VAE_SD15" gpurun_out/r03_gputest_4.log
for tp in 1 0; do
  CLORA_EPI_TWO_PHASE=$tp timeout 300 python tools/gemm_decomp.py gpurun_out/r03_gemm_decomp4_tp$tp.json > gpurun_out/r03_gemm_decomp4_tp$tp.txt 2>&1
done
paste -d'\n' gpurun_out/r03_gemm_decomp4_tp1.txt gpurun_out/r03_gemm_decomp4_tp0.txt | grep shape | cut -c1-330
timeout 300 python tools/vae_bench.py 4 512 > gpurun_out/r03_vae_bench4.txt 2>&1; tail -6 gpurun_out/r03_vae_bench4.txt
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for tp in 0 1 0 1; do
  CLORA_EPI_TWO_PHASE=$tp timeout 600 $B >> gpurun_out/r03_bench4_tp$tp.json 2>> gpurun_out/r03_bench_ab4.err
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench4_tp0.json gpurun_out/r03_bench4_tp1.json
timeout 900 python bench.py --trace-out gpurun_out/r03_kernel_stats_4.json > gpurun_out/r03_bench_4.json 2> gpurun_out/r03_bench_4.err
head -c 600 gpurun_out/r03_bench_4.json; grep -o '"ddim50".*"cpu_baseline"' gpurun_out/r03_bench_4.json | cut -c1-500
