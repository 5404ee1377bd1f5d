#!/bin/bash
# gpurun call 16 (final round-2 evidence set): all GPU tests, bench line + kernel trace, configs[3] line, PMC passes, DDIM trace
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/box_calib.py 2>&1 | grep BOX_CALIB | tee gpurun_out/box_calib_final.txt
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_final.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gputest_final.log | tail -10
( time timeout 900 python bench.py --trace-out gpurun_out/r02_kernel_stats_final.json ) > gpurun_out/bench_final.log 2>&1
grep '^{' gpurun_out/bench_final.log > gpurun_out/r02_bench_final.json
tail -c 1200 gpurun_out/bench_final.log
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --trace-out gpurun_out/r02_kernel_stats_v2_final.json ) > gpurun_out/bench_v2_final.log 2>&1
grep '^{' gpurun_out/bench_v2_final.log > gpurun_out/r02_bench_v2_final.json
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_v2_final.log | head -1
B="$R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 3 --warmup 1 --no-graph"
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m /tmp/ddimkt
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $B > $R/gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $B > $R/gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o m -- python $B > $R/gpurun_out/pmc_m.log 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/ddimkt -o kt -- python $R/tools/ddim_profile.py 4 > $R/gpurun_out/ddim_profile_final.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) gpurun_out/r02_pmc_traffic_final.json > gpurun_out/pmc_traffic_final.txt 2>&1
python tools/pmc_mfma.py $(find /tmp/pmc_m -name "*.db" | head -1) gpurun_out/r02_pmc_mfma_final.json > gpurun_out/pmc_mfma_final.txt 2>&1
python tools/rocprof_summary.py $(find /tmp/ddimkt -name "*.db" | head -1) gpurun_out/r02_ddim_kernel_stats_final 6 > /dev/null 2>&1
head -16 gpurun_out/pmc_traffic_final.txt; head -18 gpurun_out/pmc_mfma_final.txt | cut -c1-200; tail -1 gpurun_out/ddim_profile_final.log
