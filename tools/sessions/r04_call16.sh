#!/bin/bash
# Round 4, call 16: one-launch GroupNorm, second version (three barriers instead of nine, affine parameters fetched with x):
# kernel tests, same-box A/B off / on, per-grid trace with it on.
R=$PWD; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" ) > gpurun_out/r04_gputest_gn_resident2.log 2>&1
tail -2 gpurun_out/r04_gputest_gn_resident2.log
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for v in 0 1 0 1; do
  CLORA_GN_RESIDENT=$v timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB gn_resident(v2) $v', d['ms_per_step'], d['value'], d['ddim50']['latency_s'])" | tee -a gpurun_out/r04_ab_gn_resident2.txt
done
cd /tmp; export TMPDIR=/tmp
CLORA_GN_RESIDENT=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/stepkt -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 8 --warmup 2 > $R/gpurun_out/stepkt.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/stepkt -name "*.db" | head -1) gpurun_out/r04_step_trace_by_grid_gn_resident2.txt 12 200 > /dev/null 2>&1
grep -E "gn_.*resident|total" gpurun_out/r04_step_trace_by_grid_gn_resident2.txt | cut -c1-150
