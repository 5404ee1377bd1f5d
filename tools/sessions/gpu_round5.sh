#!/bin/bash
# gpurun call 5 (round 2 evidence set): GPU tests, the bench line + kernel trace, configs[3] line, PMC passes
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/gputest.log | tail -20
( time timeout 900 python bench.py --trace-out gpurun_out/r02_kernel_stats.json ) > gpurun_out/bench.log 2>&1
grep '^{' gpurun_out/bench.log > gpurun_out/r02_bench.json
tail -c 2500 gpurun_out/bench.log
( time timeout 900 python bench.py --config mpii-pose-v2.json --batch 8 --no-ddim --no-cpu-baseline --no-full-step --trace-out gpurun_out/r02_kernel_stats_v2.json ) > gpurun_out/bench_v2.log 2>&1
grep '^{' gpurun_out/bench_v2.log > gpurun_out/r02_bench_v2.json
tail -c 800 gpurun_out/bench_v2.log
# PMC passes (own runs, --pmc only): fabric traffic and MFMA busy per kernel of the eager step
B="$R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 3 --warmup 1 --no-graph"
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $B > $R/gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $B > $R/gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_m -o m -- python $B > $R/gpurun_out/pmc_m.log 2>&1
cd $R
python tools/pmc_summary.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) gpurun_out/r02_pmc_traffic.json > gpurun_out/pmc_traffic.txt 2>&1
python tools/pmc_mfma.py $(find /tmp/pmc_m -name "*.db" | head -1) gpurun_out/r02_pmc_mfma.json > gpurun_out/pmc_mfma.txt 2>&1
head -20 gpurun_out/pmc_traffic.txt; head -24 gpurun_out/pmc_mfma.txt
