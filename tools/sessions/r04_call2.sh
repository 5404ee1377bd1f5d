#!/bin/bash
# Round 4, call 2: the hoisted-epilogue root-cause experiments on the configuration that reproduced reliably in round 3 (the 64x64
# BK64 tile with the two-chunks-per-thread hoisted epilogue, -DCLORA_SMALL2_ON; tiles 43 / 23), plus the 2-segment case on tile 42.
#   small2       reproduction            small2_fz   + waitcnt forcezero        small2_nop  + drain / s_nop after the LDS reads
#   small2_1blk  90 KB LDS request (1 block per CU)   small2_2blk  60 KB (2 blocks per CU); HEAD's 48 KB ring gives 3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
out=gpurun_out/r04_hoist_diag2.txt
: > $out
for v in small2 small2_fz small2_1blk small2_2blk small2_nop; do
  CLORA_LIB_PATH=$R/controllora_amd/_build_v_$v/libclora.so HOIST_DIAG_REPS=6 timeout 300 python tools/hoist_diag.py $v 43 23 2>&1 | grep HOIST_DIAG >> $out
done
CLORA_LIB_PATH=$R/controllora_amd/_build_v_hoist_all/libclora.so HOIST_DIAG_REPS=6 timeout 300 python tools/hoist_diag.py hoist_all 42 2>&1 | grep HOIST_DIAG >> $out
grep TOTAL $out
