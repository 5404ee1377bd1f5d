#!/bin/bash
# one gpurun call: GPU test suite + headline bench (outputs under gpurun_out/)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
grep -E "FULL_TOPOLOGY|DDIM_LATENT|RANK256|CHAIN|passed|failed|rc=" gpurun_out/gputest.log | tail -40
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
