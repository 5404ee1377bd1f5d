#!/bin/bash
# Round 4, call 24: the library as rebuilt with the tile_order = 3 decode (default order unchanged): kernel suite + configs[1] fixture + smoke
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_full_topology_gpu.py -q -m gpu -k "not ddim and not infer and not vae and not v2 and not sketch and not stock and not chain and not full_sd15" ) > gpurun_out/r04_gputest_head_kernels.log 2>&1
tail -2 gpurun_out/r04_gputest_head_kernels.log
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
