#!/bin/bash
# gpurun call 4: kernel tests, A/B of the conv K order, re-tune, bench, DDIM kernel trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider ) > gpurun_out/gputest_k.log 2>&1
tail -4 gpurun_out/gputest_k.log
B="--no-cpu-baseline --no-full-step --no-ddim --steps 30"
( CLORA_KCHUNK=0 timeout 600 python bench.py $B --trace-out gpurun_out/kt_kchunk0.json ) > gpurun_out/bench_kchunk0.log 2>&1
( timeout 600 python bench.py $B --trace-out gpurun_out/kt_kchunk64.json ) > gpurun_out/bench_kchunk64.log 2>&1
for f in gpurun_out/bench_kchunk0.log gpurun_out/bench_kchunk64.log; do grep -o '"ms_per_step": [0-9.]*' $f | head -1; grep -o '"family_ms_per_step": [0-9.]*' $f; done
( time timeout 1500 python tools/tune_gemm.py ) > gpurun_out/tune_a.log 2>&1
tail -2 gpurun_out/tune_a.log
( time timeout 900 python tools/tune_gemm.py --config mpii-pose-v2.json --batch 8 --infer-batch 0 --merge ) > gpurun_out/tune_b.log 2>&1
tail -2 gpurun_out/tune_b.log
cp controllora_amd/gemm_tuning_gfx950.json gpurun_out/gemm_tuning_gfx950.json
( timeout 900 python bench.py --no-cpu-baseline --no-full-step --steps 30 --trace-out gpurun_out/kt_retuned.json ) > gpurun_out/bench_retuned.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_retuned.log | head -1; grep -o '"latency_s": [0-9.]*' gpurun_out/bench_retuned.log
cd /tmp && rm -rf /tmp/ddimkt && timeout 600 rocprofv3 --kernel-trace -d /tmp/ddimkt -o kt -- python $GRAFT_REPO_ROOT/tools/ddim_profile.py 4 > $GRAFT_REPO_ROOT/gpurun_out/ddim_profile.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find /tmp/ddimkt -name "*.db" | head -1) gpurun_out/ddim_kernel_stats 6 > gpurun_out/ddim_kernel_stats.txt 2>&1
head -30 gpurun_out/ddim_kernel_stats.txt
