#!/bin/bash
# Build libclora.so with the DMA-traffic timing probes of clora_gemm.hip (-DCLORA_DMA_PROBE: tile_cfg 91..96) into
# controllora_amd/_build_probe/ (never the product library).  Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/.."
mkdir -p controllora_amd/_build_probe
B=controllora_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCLORA_DMA_PROBE -c controllora_amd/csrc/clora_gemm.hip -o controllora_amd/_build_probe/clora_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/clora_attn.o $B/clora_ew.o controllora_amd/_build_probe/clora_gemm.o $B/clora_lora.o $B/clora_norm.o -o controllora_amd/_build_probe/libclora.so
echo built controllora_amd/_build_probe/libclora.so
