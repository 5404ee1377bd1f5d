#!/bin/bash
# Round 3, closing evidence set at HEAD: all GPU tests, smoke, the bench line (live PMC traffic, kernel trace), BASELINE configs[3],
# MFMA-busy and per-kernel traffic PMC passes, per-(kernel, grid) trace, the long CPU baseline.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/r03_box_calib_final.txt
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/r03_gputest_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gputest_final.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r03_gputest_final.log | tail -6
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee gpurun_out/r03_smoke_final.txt
( time timeout 1200 python bench.py --trace-out gpurun_out/r03_kernel_stats_final.json ) > gpurun_out/r03_bench_final.log 2>&1
grep '^{' gpurun_out/r03_bench_final.log > gpurun_out/r03_bench_final.json
head -c 400 gpurun_out/r03_bench_final.json; echo
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --no-pmc --trace-out gpurun_out/r03_kernel_stats_v2_final.json ) > gpurun_out/r03_bench_v2_final.log 2>&1
grep '^{' gpurun_out/r03_bench_v2_final.log > gpurun_out/r03_bench_v2_final.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03_bench_v2_final.log | head -1
B="$R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 3 --warmup 1 --no-graph"
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m /tmp/stepkt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $B > $R/gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $B > $R/gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o m -- python $B > $R/gpurun_out/pmc_m.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 8 --warmup 2 > $R/gpurun_out/stepkt.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) gpurun_out/r03_pmc_traffic_final.json > gpurun_out/r03_pmc_traffic_final.txt 2>&1
python tools/pmc_mfma.py $(find /tmp/pmc_m -name "*.db" | head -1) gpurun_out/r03_pmc_mfma_final.json > gpurun_out/r03_pmc_mfma_final.txt 2>&1
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r03_step_trace_by_grid_final.txt 12 90 > /dev/null 2>&1
head -14 gpurun_out/r03_pmc_traffic_final.txt | cut -c1-200; head -16 gpurun_out/r03_pmc_mfma_final.txt | cut -c1-200
timeout 900 python -c "
import json, bench
print('CPU_BASELINE_FULL', json.dumps(bench.cpu_baseline(steps=3, full=True)))" 2>&1 | tail -1 | tee gpurun_out/r03_cpu_baseline_full.txt
