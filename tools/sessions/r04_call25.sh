#!/bin/bash
# Round 4, call 25: rebuilt library, the remaining full-size fixtures (configs[3] bs 8, UNet batch 32, VAE 512^2)
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_full_topology_gpu.py -q -m gpu -k "config3 or batch32 or vae_512" ) > gpurun_out/r04_gputest_head_fixtures.log 2>&1
tail -2 gpurun_out/r04_gputest_head_fixtures.log
