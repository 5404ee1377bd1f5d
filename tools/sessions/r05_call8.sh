#!/bin/bash
# Round 5, call 8: forward attention with 16-wave blocks (512 queries): parity + A/B at the level-0 shapes.
set -x
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "block_widths" ) 2>&1 | tail -4 | tee gpurun_out/r05_gputest_attn_widths.log
timeout 600 python tools/attn_fwd_width_ab.py 5 2>&1 | tee gpurun_out/r05_attn_fwd_width_ab.txt
