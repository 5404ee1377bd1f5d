#!/bin/bash
# Round 3, GPU call 2: new kernel tests (two-phase epilogue, VAE at SD-1.5 widths), where a short-K GEMM spends its time
# (tools/gemm_decomp.py) with the two-phase chunk loop on / off, the stock-torch kernels left in the step with their call sites,
# VAE encode profile, same-box A/B of the bench line, then a full line with the live PMC traffic passes.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -s -k "without_rowadd or vae or clip or tile_configs or gemm" ) > gpurun_out/r03_gputest_2.log 2>&1
tail -3 gpurun_out/r03_gputest_2.log; grep -h "VAE_SD15\|CLIP_" gpurun_out/r03_gputest_2.log
for tp in 1 0; do
  CLORA_EPI_TWO_PHASE=$tp timeout 300 python tools/gemm_decomp.py gpurun_out/r03_gemm_decomp_tp$tp.json > gpurun_out/r03_gemm_decomp_tp$tp.txt 2>&1
done
paste -d'\n' gpurun_out/r03_gemm_decomp_tp1.txt gpurun_out/r03_gemm_decomp_tp0.txt | cut -c1-400
timeout 300 python tools/torch_ops_profile.py > gpurun_out/r03_torch_ops.txt 2>&1; head -45 gpurun_out/r03_torch_ops.txt | cut -c1-260
B="python bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 30 --warmup 5"
for tp in 0 1 0 1; do
  CLORA_EPI_TWO_PHASE=$tp timeout 600 $B >> gpurun_out/r03_bench_tp$tp.json 2>> gpurun_out/r03_bench_ab2.err
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench_tp0.json gpurun_out/r03_bench_tp1.json
timeout 300 python tools/vae_bench.py 4 512 > gpurun_out/r03_vae_bench.txt 2>&1; tail -9 gpurun_out/r03_vae_bench.txt
cd /tmp && rm -rf /tmp/vaekt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vaekt -o kt -- python $GRAFT_REPO_ROOT/tools/vae_bench.py 4 512 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find /tmp/vaekt -name "*.db" | head -1) gpurun_out/r03_vae_kernel_stats 5 > gpurun_out/r03_vae_kernel_stats.txt 2>&1; head -20 gpurun_out/r03_vae_kernel_stats.txt | cut -c1-200
timeout 900 python bench.py --trace-out gpurun_out/r03_kernel_stats_2.json > gpurun_out/r03_bench_2.json 2> gpurun_out/r03_bench_2.err
head -c 3000 gpurun_out/r03_bench_2.json
