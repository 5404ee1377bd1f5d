#!/bin/bash
# Round 4, call 19: the control map's share of the concat adapters' down-projections evaluated once per level (v2 / concat_hidden:
# ops.control_down_parts, CLORA_CONTROL_PARTS): configs[3] fixture + v2 topology tests, then same-box A/B on configs[3] (bs 8).
R=$PWD; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -m gpu -k "v2 or sketch or broadcast or repeat or chain or stock" ) > gpurun_out/r04_gputest_control_parts.log 2>&1
tail -3 gpurun_out/r04_gputest_control_parts.log
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --no-ddim --steps 30 --warmup 5 --config mpii-pose-v2.json --batch 8"
for v in 0 1 0 1; do
  CLORA_CONTROL_PARTS=$v timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB control_parts v2-bs8 $v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_control_parts.txt
done
