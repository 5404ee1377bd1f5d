#!/bin/bash
# Round 5, call 2: first hardware run of tile_cfg 59 (gemm_8p_kernel): parity cases, race screen, A/B against the incumbent tiles.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "59 or eight_phase" ) 2>&1 | tail -5 | tee gpurun_out/r05_gputest_tile59.log
timeout 900 python tools/gemm8p_ab.py 5 2>&1 | tee gpurun_out/r05_gemm8p_ab.txt
