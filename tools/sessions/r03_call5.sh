#!/bin/bash
# Round 3, GPU call 5: diagnostic of the 64x64-tile two-phase epilogue failure (transposed-U sub-case), two-stream probe, VAE conv A/B.
set -x
mkdir -p gpurun_out
timeout 300 python tools/epi_diag.py > gpurun_out/r03_epi_diag.txt 2>&1; cat gpurun_out/r03_epi_diag.txt | cut -c1-420
timeout 300 python tools/two_stream_probe.py gpurun_out/r03_two_stream.json > gpurun_out/r03_two_stream.txt 2>&1; cat gpurun_out/r03_two_stream.txt | cut -c1-300
timeout 300 python tools/vae_conv_ab.py > gpurun_out/r03_vae_conv_ab.txt 2>&1; cat gpurun_out/r03_vae_conv_ab.txt | cut -c1-400
