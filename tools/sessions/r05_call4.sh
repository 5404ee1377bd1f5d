#!/bin/bash
# Round 5, call 4: ping-pong patch-conv tiles 84 / 86 / 89: parity on hardware, then A/B against 71 / 76 / 79.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "conv_patch_kernel and (84 or 86 or 89)" ) 2>&1 | tail -5 | tee gpurun_out/r05_gputest_pingpong.log
timeout 900 python tools/patch_ab.py 5 2>&1 | tee gpurun_out/r05_patch_pingpong_ab.txt
