#!/bin/bash
# Round 4, call 17: GroupNorm apply kernels with the first batch of rows / the affine parameters / the statistics requested BEFORE the
# partials are folded (one memory round trip instead of three dependent ones).  Kernel tests, then same-box A/B against a library
# built from the previous clora_norm.hip (CLORA_LIB_PATH), train step + DDIM-50 + VAE.
R=$PWD; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "groupnorm or bit_stable" ) > gpurun_out/r04_gputest_gn_prefetch.log 2>&1
tail -2 gpurun_out/r04_gputest_gn_prefetch.log
B="bench.py --no-cpu-baseline --no-full-step --no-pmc --no-rocprof --steps 30 --warmup 5"
for v in before after before after; do
  if [ $v = before ]; then export CLORA_LIB_PATH=$R/controllora_amd/_build_v_gn_before/libclora.so; else unset CLORA_LIB_PATH; fi
  timeout 600 python $B 2> /dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('AB gn_prefetch $v', d['ms_per_step'], d['value'], d['ddim50']['latency_s'])" | tee -a gpurun_out/r04_ab_gn_prefetch.txt
done
for v in before after; do
  if [ $v = before ]; then export CLORA_LIB_PATH=$R/controllora_amd/_build_v_gn_before/libclora.so; else unset CLORA_LIB_PATH; fi
  echo "VAE $v" >> gpurun_out/r04_ab_gn_prefetch.txt
  timeout 300 python tools/vae_bench.py 2>/dev/null | head -3 | tee -a gpurun_out/r04_ab_gn_prefetch.txt
done
