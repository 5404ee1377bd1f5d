#!/bin/bash
# round 6, gpurun call 8: GroupNorm parameter-gradient fold inside the apply launch (parity, same-box A/B against the previous norm
# kernels through CLORA_LIB_PATH), 16 vs 32 adapter weight-gradient jobs per launch, the whole GPU suite with per-test durations
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "groupnorm or elementwise or bit_stable" ) > gpurun_out/gputest_s8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s8.log
grep -E "passed|failed|rc=" gpurun_out/gputest_s8.log | tail -3
B="--no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-roofline --steps 30"
PREV=$GRAFT_REPO_ROOT/controllora_amd/_build_prev/libclora.so
for i in 1 2 3; do
  CLORA_LIB_PATH=$PREV timeout 600 python bench.py $B > gpurun_out/ab8_prevnorm_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab8_head_$i.log 2>&1
  CLORA_WGRAD_JOBS=16 timeout 600 python bench.py $B > gpurun_out/ab8_wgrad16_$i.log 2>&1
done
for f in gpurun_out/ab8_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=60 ) > gpurun_out/gputest_s8_all.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s8_all.log
grep -E "passed|failed|rc=|^real" gpurun_out/gputest_s8_all.log | tail -4
grep -E "s call|s setup" gpurun_out/gputest_s8_all.log | head -60
