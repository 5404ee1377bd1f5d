#!/bin/bash
# round 6, gpurun call 4: launch table -- the signatures new this round (grouped text K|V projections) and the VAE's shapes with the
# round-5 tiles (59 / 79 / 77 / 78) offered; VAE bench before / after; train bench after
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/vae_bench.py 4 512 2>&1 | grep -E "encode|gemm|groupnorm" | tee gpurun_out/r06_vae_bench_before.txt
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_before.json
( time timeout 1200 python tools/tune_gemm.py --vae 4 --cfgs 1,7,8,21,22,41,42,53,56,57,58,59,71,72,73,76,77,78,79 ) > gpurun_out/r06_tune_vae.log 2>&1
tail -5 gpurun_out/r06_tune_vae.log
timeout 300 python tools/vae_bench.py 4 512 2>&1 | grep -E "encode|gemm|groupnorm" | tee gpurun_out/r06_vae_bench_after.txt
( time timeout 1500 python tools/tune_gemm.py --merge --infer-batch 0 --plain-only ) > gpurun_out/r06_tune_plain_train.log 2>&1
tail -5 gpurun_out/r06_tune_plain_train.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  CLORA_GEMM_TUNING_FILE=gpurun_out/gemm_tuning_before.json timeout 600 python bench.py $B > gpurun_out/ab4_before_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab4_after_$i.log 2>&1
done
for f in gpurun_out/ab4_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1); done
