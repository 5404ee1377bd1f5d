#!/bin/bash
# Round 6 evidence set at HEAD: all GPU tests, smoke, the bench line (calibration, three windows, live PMC traffic, kernel trace),
# BASELINE configs[3], MFMA-busy and per-kernel traffic PMC passes, per-(kernel, grid) trace, the 8192^3 calibration GEMM on tile 59
# with its own MFMA-busy pass.  TAG distinguishes repeated runs (default: final).
set -x
TAG=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/r06_box_calib_$TAG.txt
( time timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=12 ) > gpurun_out/r06_gputest_$TAG.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_gputest_$TAG.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|NOTE " gpurun_out/r06_gputest_$TAG.log | cut -c1-300 | tail -8
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee gpurun_out/r06_smoke_$TAG.txt
( time timeout 1200 python bench.py --mix-lora post --trace-out gpurun_out/r06_kernel_stats_$TAG.json ) > gpurun_out/r06_bench_$TAG.log 2>&1
grep '^{' gpurun_out/r06_bench_$TAG.log > gpurun_out/r06_bench_$TAG.json
head -c 400 gpurun_out/r06_bench_$TAG.json; echo
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --no-pmc --trace-out gpurun_out/r06_kernel_stats_v2_$TAG.json ) > gpurun_out/r06_bench_v2_$TAG.log 2>&1
grep '^{' gpurun_out/r06_bench_v2_$TAG.log > gpurun_out/r06_bench_v2_$TAG.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06_bench_v2_$TAG.log | head -1
B="$R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --no-calibration --windows 1 --steps 3 --warmup 1 --no-graph"
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m /tmp/stepkt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $B > $R/gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $B > $R/gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o m -- python $B > $R/gpurun_out/pmc_m.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --no-calibration --windows 1 --steps 8 --warmup 2 > $R/gpurun_out/stepkt.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) gpurun_out/r06_pmc_traffic_$TAG.json > gpurun_out/r06_pmc_traffic_$TAG.txt 2>&1
python tools/pmc_mfma.py $(find /tmp/pmc_m -name "*.db" | head -1) gpurun_out/r06_pmc_mfma_$TAG.json > gpurun_out/r06_pmc_mfma_$TAG.txt 2>&1
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r06_step_trace_by_grid_$TAG.txt 12 140 > /dev/null 2>&1
head -14 gpurun_out/r06_pmc_traffic_$TAG.txt | cut -c1-200; head -16 gpurun_out/r06_pmc_mfma_$TAG.txt | cut -c1-200
timeout 300 python tools/vae_bench.py 4 512 2>&1 | grep -E "encode|gemm|groupnorm" | tee gpurun_out/r06_vae_bench_$TAG.txt
( time timeout 600 python bench.py --config danbooru-sketch.json --no-ddim --no-cpu-baseline --no-full-step --no-pmc --no-roofline ) > gpurun_out/r06_bench_sketch_$TAG.log 2>&1
grep '^{' gpurun_out/r06_bench_sketch_$TAG.log > gpurun_out/r06_bench_sketch_$TAG.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06_bench_sketch_$TAG.log | head -1
