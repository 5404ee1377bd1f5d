#!/bin/bash
# gpurun call 7: MFMA issue rates (fixed probe), attention with LDS-DMA double-buffered tiles (A/B vs previous build, dK/dV at
# 2 vs 3 blocks per CU), where the GEMM wave-cycles go (SQ counters), bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tools/probes/_build/mfma_rate_probe > gpurun_out/mfma_rate_probe.txt 2>&1; cat gpurun_out/mfma_rate_probe.txt
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" -p no:cacheprovider ) > gpurun_out/gputest_attn.log 2>&1
tail -3 gpurun_out/gputest_attn.log
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 300 python tools/attn_ab.py gpurun_out/attn_prev.json ) > gpurun_out/attn_prev.log 2>&1
( timeout 300 python tools/attn_ab.py gpurun_out/attn_new.json ) > gpurun_out/attn_new.log 2>&1
( CLORA_ATTN_DKV_OCC=3 timeout 300 python tools/attn_ab.py gpurun_out/attn_new_occ3.json ) > gpurun_out/attn_new_occ3.log 2>&1
paste <(grep -o '"us": [0-9.]*' gpurun_out/attn_prev.log) <(grep -o '"us": [0-9.]*' gpurun_out/attn_new_occ3.log) <(grep -o '"kernel": "[^"]*", "us": [0-9.]*' gpurun_out/attn_new.log)
B="--no-cpu-baseline --no-full-step --steps 30"
( timeout 900 python bench.py $B --trace-out gpurun_out/kt_attn2.json ) > gpurun_out/bench_attn2.log 2>&1
( CLORA_ATTN_DKV_OCC=3 timeout 600 python bench.py $B --no-roofline --no-ddim ) > gpurun_out/bench_attn2_occ3.log 2>&1
for f in gpurun_out/bench_attn2.log gpurun_out/bench_attn2_occ3.log; do grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"latency_s": [0-9.]*' $f; done
cd /tmp
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
rm -rf /tmp/gp1 /tmp/gp2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d /tmp/gp1 -o p -- python $R/tools/gemm_pmc.py > $R/gpurun_out/gemm_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC -d /tmp/gp2 -o p -- python $R/tools/gemm_pmc.py > $R/gpurun_out/gemm_pmc2.log 2>&1
cd $R
python tools/gemm_pmc.py --summarize $(find /tmp/gp1 -name "*.db" | head -1) gpurun_out/gemm_pmc1.json > gpurun_out/gemm_pmc1.txt 2>&1
python tools/gemm_pmc.py --summarize $(find /tmp/gp2 -name "*.db" | head -1) gpurun_out/gemm_pmc2.json > gpurun_out/gemm_pmc2.txt 2>&1
cat gpurun_out/gemm_pmc1.txt gpurun_out/gemm_pmc2.txt | cut -c1-400
tail -3 gpurun_out/gemm_pmc1.log gpurun_out/gemm_pmc2.log
