#!/bin/bash
# Round 4, call 4: (a) the scalar-FMA build of the round-3 reproducer: same hoisted epilogue, the rank-4 update forced to v_fma_f32
# (packed-fp32 hypothesis); (b) fused adapter-down tests; (c) same-box A/B train step CLORA_FUSE_DOWN=0/1 with the fusion
# restricted to launches that fill the chip; (d) per-(kernel, grid) trace of the fused step; (e) DDIM-50 A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CLORA_LIB_PATH=$R/controllora_amd/_build_v_small2_sfma/libclora.so HOIST_DIAG_REPS=8 timeout 300 python tools/hoist_diag.py small2_sfma 43 23 2>&1 | grep HOIST_DIAG > gpurun_out/r04_hoist_diag4.txt
CLORA_LIB_PATH=$R/controllora_amd/_build_v_small2/libclora.so HOIST_DIAG_REPS=3 timeout 300 python tools/hoist_diag.py small2_again 43 2>&1 | grep HOIST_DIAG_TOTAL >> gpurun_out/r04_hoist_diag4.txt
grep TOTAL gpurun_out/r04_hoist_diag4.txt
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "adapter_down" ) > gpurun_out/r04_gputest_fused2.log 2>&1
tail -2 gpurun_out/r04_gputest_fused2.log
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
: > gpurun_out/r04_ab_fuse2.txt
for f in 0 1 0 1; do
  CLORA_FUSE_DOWN=$f timeout 600 python $B 2> gpurun_out/r04_ab_fuse2_$f.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB fuse_down=$f', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_fuse2.txt
done
cd /tmp; rm -rf /tmp/kt1
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt1 -o kt -- python $R/bench.py --no-cpu-baseline --no-ddim --no-roofline --no-full-step --steps 8 --warmup 2 > $R/gpurun_out/r04_kt1.log 2>&1
cd $R
python tools/trace_by_grid.py $(find /tmp/kt1 -name "*.db" | head -1) gpurun_out/r04_step_trace_by_grid_fuse1.txt 10 70 > /dev/null 2>&1
head -30 gpurun_out/r04_step_trace_by_grid_fuse1.txt | cut -c1-150
for f in 0 1; do
  CLORA_FUSE_DOWN=$f timeout 600 python bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 3 --warmup 1 2> gpurun_out/r04_ddim_ab_$f.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('DDIM fuse_down=$f', json.dumps(d['ddim50']))" | tee -a gpurun_out/r04_ab_fuse2.txt
done
