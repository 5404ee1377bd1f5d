#!/bin/bash
# round 6, gpurun call 16: compensated residual trunk (clora_epilogue_t.residual_lo / c_lo) -- kernel parity, the fixture errors with the
# trunk off / on at inference / always, cost on the DDIM leg and (A/B of the two libraries) on the train step
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "compensated_trunk or gemm_epilogue or fused_layernorm or deferred or bit_stab" ) > gpurun_out/gputest_s16.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_s16.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/gputest_s16.log | cut -c1-300 | tail -8
for mode in off infer always; do
  ( CLORA_TRUNK_LO=$mode timeout 1500 python -m pytest tests/test_full_topology_gpu.py -q -s -p no:cacheprovider -k "config1_train_step or unet_batch32 or ddim50_512" ) > gpurun_out/gputest_s16_$mode.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/gputest_s16_$mode.log
  echo "== CLORA_TRUNK_LO=$mode"; grep -E "FULL_SIZE|passed|failed|rc=" gpurun_out/gputest_s16_$mode.log | cut -c1-400 | tail -8
done
B="--no-cpu-baseline --no-full-step --no-pmc --no-roofline --steps 30"
for i in 1 2; do
  CLORA_LIB_PATH=controllora_amd/_build_prev/libclora.so CLORA_TRUNK_LO=off timeout 600 python bench.py $B > gpurun_out/ab16_prevlib_$i.log 2>&1
  CLORA_TRUNK_LO=off timeout 600 python bench.py $B > gpurun_out/ab16_off_$i.log 2>&1
  timeout 600 python bench.py $B > gpurun_out/ab16_infer_$i.log 2>&1
  CLORA_TRUNK_LO=always timeout 600 python bench.py $B > gpurun_out/ab16_always_$i.log 2>&1
done
for f in gpurun_out/ab16_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"gemm8192_cfg1_us": [0-9.]*' $f | head -1) $(grep -o '"latency_s": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done
