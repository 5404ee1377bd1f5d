#!/bin/bash
# round 6, gpurun call 15: the error budget one level down: which roundings inside the worst block carry its error,
# and what a compensated / fp32 residual trunk would buy on the whole UNet (fp32 oracle with the product's roundings)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/error_budget_sublayers.py gpurun_out/r06_error_budget_sublayers.json 512 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_error_budget_sublayers.txt
timeout 900 python tools/error_budget_sublayers.py gpurun_out/r06_error_budget_trunk.json 512 2 unet 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_error_budget_trunk.txt
