#!/bin/bash
# Round 4, call 5: the hoisted epilogue with the scalar rank-4 update as the production build (hoist back on every kernel built for
# <= 2 blocks per CU + the 64x64 two-chunk form).  (a) same-box control: the production library vs the same sources with
# -DCLORA_HOIST_PACKED_FMA (expected: clean vs dirty); (b) strict epilogue / fused / tuned-table / bit-stability GPU tests on the
# production build; (c) same-box A/B of the train step: round-3 hoist scope (-DCLORA_HOIST_8WAVE_ONLY -DCLORA_SMALL2_OFF) vs production.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
out=gpurun_out/r04_hoist_diag5.txt
: > $out
HOIST_DIAG_REPS=8 timeout 400 python tools/hoist_diag.py production 43 23 42 22 21 2>&1 | grep HOIST_DIAG >> $out
CLORA_LIB_PATH=$R/controllora_amd/_build_v_packedfma/libclora.so HOIST_DIAG_REPS=4 timeout 400 python tools/hoist_diag.py packedfma 43 23 42 2>&1 | grep HOIST_DIAG >> $out
grep TOTAL $out
( timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "epilogue or adapter_down or tuned_table or bit_stable or tile_configs" ) > gpurun_out/r04_gputest_hoist.log 2>&1
tail -3 gpurun_out/r04_gputest_hoist.log
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
: > gpurun_out/r04_ab_hoist.txt
for v in r03scope production r03scope production; do
  if [ $v = production ]; then lib=$R/controllora_amd/_build/libclora.so; else lib=$R/controllora_amd/_build_v_$v/libclora.so; fi
  CLORA_LIB_PATH=$lib timeout 600 python $B 2> gpurun_out/r04_ab_hoist_$v.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB hoist $v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_hoist.txt
done
