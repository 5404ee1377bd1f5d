#!/bin/bash
# Round 4, call 7: the v1 control terms in rank space.  (a) kernel parity at the level shapes; (b) end-to-end parity: the full-size
# fixtures (configs[1] bs4, configs[3] v2 bs8, inference at UNet batch 32, VAE 512^2) and the 256^2 oracle tests; (c) same-box A/B
# CLORA_RANK_CONTROL=0 (materialised control terms, round-3 path + this round's fusions) vs 1.
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "rank_space or adapter_down or test_lora" ) > gpurun_out/r04_gputest_rank.log 2>&1
grep -E "RANK_CONTROL|passed|failed" gpurun_out/r04_gputest_rank.log | cut -c1-250
( timeout 1500 python -m pytest tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -k "fixture or matches_oracle or full_size_properties" ) > gpurun_out/r04_gputest_full.log 2>&1
grep -E "FULL_|passed|failed|Error" gpurun_out/r04_gputest_full.log | cut -c1-400
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
: > gpurun_out/r04_ab_rank.txt
for f in 0 1 0 1; do
  CLORA_RANK_CONTROL=$f timeout 600 python $B 2> gpurun_out/r04_ab_rank_$f.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB rank_control=$f', d['ms_per_step'], d['value'], d['loss'])" | tee -a gpurun_out/r04_ab_rank.txt
done
