#!/bin/bash
# Round 4, call 3: (a) the no-SLP build of the round-3 reproducer (packed-fp32 hypothesis for the hoisted epilogue); (b) parity of
# the adapter down-projection inside the projection GEMMs at real shapes + the strict epilogue tests; (c) same-box A/B of the
# train step: CLORA_FUSE_DOWN=0 (separate lora_down launches, the round-3 path) vs 1 (default).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
out=gpurun_out/r04_hoist_diag3.txt
CLORA_LIB_PATH=$R/controllora_amd/_build_v_small2_noslp/libclora.so HOIST_DIAG_REPS=6 timeout 300 python tools/hoist_diag.py small2_noslp 43 23 2>&1 | grep HOIST_DIAG > $out
grep TOTAL $out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "adapter_down or epilogue" ) > gpurun_out/r04_gputest_fused.log 2>&1
tail -3 gpurun_out/r04_gputest_fused.log
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for f in 0 1 0 1; do
  CLORA_FUSE_DOWN=$f timeout 600 python $B 2> gpurun_out/r04_ab_fuse_$f.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB fuse_down=$f', d['ms_per_step'], d['value'], d.get('launches_per_step'))" | tee -a gpurun_out/r04_ab_fuse.txt
done
( timeout 900 python bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --steps 20 --trace-out gpurun_out/r04_kernel_stats_fuse1.json ) > gpurun_out/r04_bench_fuse1.log 2>&1
grep '^{' gpurun_out/r04_bench_fuse1.log > gpurun_out/r04_bench_fuse1.json; head -c 600 gpurun_out/r04_bench_fuse1.json; echo
