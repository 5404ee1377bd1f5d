#!/bin/bash
# Round 4, call 20: the bench line with the exchange step on the C ABI's own RCCL communicator (bench.py --comm clora), one rank:
# clora_comm_init / clora_allreduce_flat_f32 / clora_comm_destroy on hardware, allreduce_ms from that path.
mkdir -p gpurun_out
timeout 600 python bench.py --comm clora --no-cpu-baseline --no-ddim --no-full-step --no-pmc --no-rocprof --steps 20 --warmup 5 2> gpurun_out/r04_bench_comm_clora.err | grep '^{' > gpurun_out/r04_bench_comm_clora.json
python -c "
import json; d = json.loads(open('gpurun_out/r04_bench_comm_clora.json').read().split('\n')[0])
print(d['ms_per_step'], d['value'], {k: d['config'][k] for k in ('comm', 'rccl_ranks', 'allreduce_ms', 'allreduce_bytes')})"
tail -3 gpurun_out/r04_bench_comm_clora.err
