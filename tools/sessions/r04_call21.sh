#!/bin/bash
# Round 4, call 21: the alternate settings of the round's two new library knobs are green too: epi_hoist = 0 (adapter epilogue fetches
# its operands per chunk on every tile) and gn_resident = 0 (two-launch GroupNorm everywhere) -- kernel suite + the configs[1] fixture.
mkdir -p gpurun_out
( CLORA_EPI_HOIST=0 CLORA_GN_RESIDENT=0 timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_full_topology_gpu.py -q -m gpu -k "not ddim and not infer and not vae and not v2 and not sketch and not stock and not chain" ) > gpurun_out/r04_gputest_knobs_off.log 2>&1
tail -3 gpurun_out/r04_gputest_knobs_off.log
