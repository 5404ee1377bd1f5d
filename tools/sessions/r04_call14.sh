#!/bin/bash
# Round 4, call 14: one-launch (register-resident) GroupNorm for the 8x8 / 16x16 / 32x32 maps: kernel tests on the real library, then
# same-box A/B of the train step, configs[3] and DDIM-50 with the option off / on (CLORA_GN_RESIDENT), and a per-grid kernel trace.
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm" ) > gpurun_out/r04_gputest_gn_resident.log 2>&1
tail -3 gpurun_out/r04_gputest_gn_resident.log
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for v in 0 1 0 1; do
  CLORA_GN_RESIDENT=$v timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB gn_resident $v', d['ms_per_step'], d['value'], d['ddim50']['latency_s'])" | tee -a gpurun_out/r04_ab_gn_resident.txt
done
for v in 0 1; do
  CLORA_GN_RESIDENT=$v timeout 600 python $B --config mpii-pose-v2.json --batch 8 --no-ddim 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB gn_resident v2-bs8 $v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_gn_resident.txt
done
( timeout 900 python -m pytest tests/test_full_topology_gpu.py -q -m gpu -k "fixture or properties" ) > gpurun_out/r04_gputest_gn_resident_full.log 2>&1
tail -3 gpurun_out/r04_gputest_gn_resident_full.log
