#!/bin/bash
# Round 4, call 8: the adapter down-projection also on the 4-wave BK64 tiles (8-row operand blocks): parity at the deeper levels'
# shapes, strict epilogue tests, same-box A/B CLORA_FUSE_SMALL=0/1, DDIM A/B, per-(kernel, grid) trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "adapter_down or epilogue or rank_space" ) > gpurun_out/r04_gputest_fused3.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r04_gputest_fused3.log | tail -3
( timeout 900 python -m pytest tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -k "config1_train_step or matches_oracle" ) > gpurun_out/r04_gputest_full2.log 2>&1
grep -E "FULL_|passed|failed|Error" gpurun_out/r04_gputest_full2.log | cut -c1-330
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
: > gpurun_out/r04_ab_small.txt
for f in 0 1 0 1; do
  CLORA_FUSE_SMALL=$f timeout 600 python $B 2> gpurun_out/r04_ab_small_$f.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB fuse_small=$f', d['ms_per_step'], d['value'], d['loss'])" | tee -a gpurun_out/r04_ab_small.txt
done
for f in 0 1; do
  CLORA_FUSE_SMALL=$f timeout 600 python bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 3 --warmup 1 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('DDIM fuse_small=$f', json.dumps(d['ddim50']))" | tee -a gpurun_out/r04_ab_small.txt
done
cd /tmp; rm -rf /tmp/kt2
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt2 -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 8 --warmup 2 > $R/gpurun_out/r04_kt2.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/kt2 -name "*.db" | head -1) gpurun_out/r04_step_trace_by_grid_2.txt 10 90 > /dev/null 2>&1
head -12 gpurun_out/r04_step_trace_by_grid_2.txt | cut -c1-150
