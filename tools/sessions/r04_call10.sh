#!/bin/bash
# Round 4, call 10: attention backward with a software-pipelined dK/dV tile loop (-DCLORA_ATTN_PIPE) vs the default build:
# kernel-level A/B at the SD-1.5 site shapes, parity of the variant, train-step A/B.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
V=$R/controllora_amd/_build_v_attn_pipe/libclora.so
timeout 300 python tools/attn_ab.py gpurun_out/r04_attn_ab_default.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attn_ab_default.txt
CLORA_LIB_PATH=$V timeout 300 python tools/attn_ab.py gpurun_out/r04_attn_ab_pipe.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attn_ab_pipe.txt
( CLORA_LIB_PATH=$V timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention or bit_stable" ) > gpurun_out/r04_gputest_attn_pipe.log 2>&1
tail -2 gpurun_out/r04_gputest_attn_pipe.log
B="bench.py --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for v in default pipe default pipe; do
  if [ $v = default ]; then lib=$R/controllora_amd/_build/libclora.so; else lib=$V; fi
  CLORA_LIB_PATH=$lib timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB attn $v', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r04_ab_attn_pipe.txt
done
