#!/bin/bash
# Round 4, call 1: root-cause experiments for the hoisted adapter epilogue.  Five builds of the library, same
# diagnostic (tools/hoist_diag.py: M = 16384, N = K = 320 on the 128x64 BK64 tiles 42 / 22 with the hoist forced on):
#   default     HEAD (hoist only on the 8-wave tiles)                      -> expected clean (control)
#   hoist_all   -DCLORA_HOIST_ALL (the round-2 scope)                        -> expected dirty (reproduction)
#   hoist_fz    + -mllvm -amdgpu-waitcnt-forcezero                           -> clean => a missing / short wait
#   hoist_1blk  + 90 KB LDS request (ONE block per CU, same instructions)   -> clean => inter-block effect
#   hoist_nop   + drain + s_nop after the LDS reads of every chunk          -> clean => intra-wave hazard near the LDS reads
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
out=gpurun_out/r04_hoist_diag.txt
: > $out
for v in default hoist_all hoist_fz hoist_1blk hoist_nop; do
  if [ $v = default ]; then lib=$R/controllora_amd/_build/libclora.so; else lib=$R/controllora_amd/_build_v_$v/libclora.so; fi
  CLORA_LIB_PATH=$lib timeout 300 python tools/hoist_diag.py $v 42 22 2>&1 | grep HOIST_DIAG >> $out
done
grep TOTAL $out
