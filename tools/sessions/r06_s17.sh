#!/bin/bash
# round 6, gpurun call 17: the per-block error budget (tools/error_budget_layers.py) at HEAD, compensated trunk off and on
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
CLORA_TRUNK_LO=off timeout 600 python tools/error_budget_layers.py gpurun_out/r06_error_budget_layers_trunk_off.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_error_budget_layers_trunk_off.txt
timeout 600 python tools/error_budget_layers.py gpurun_out/r06_error_budget_layers.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_error_budget_layers.txt
head -3 gpurun_out/r06_error_budget_layers_trunk_off.txt; head -3 gpurun_out/r06_error_budget_layers.txt; tail -2 gpurun_out/r06_error_budget_layers.txt | cut -c1-300
