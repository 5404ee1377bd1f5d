#!/bin/bash
# Round 5, call 6: danbooru-sketch.json (rank-256 concat control adapters) on the wide-rank GEMM path: parity tests, bench line A/B.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_full_topology_gpu.py tests/test_e2e_gpu.py -q -x -s -p no:cacheprovider -k "sketch or rank256" ) > gpurun_out/r05_gputest_sketch_wide.log 2>&1
grep -E "passed|failed|FULL_TOPOLOGY|^FAILED|^ERROR|rank" gpurun_out/r05_gputest_sketch_wide.log | cut -c1-400 | tail -8
Q="--config danbooru-sketch.json --no-ddim --no-cpu-baseline --no-full-step --no-pmc"
( timeout 600 python bench.py $Q --trace-out gpurun_out/r05_kernel_stats_sketch_wide.json ) > gpurun_out/r05_bench_sketch_wide.log 2>&1
grep '^{' gpurun_out/r05_bench_sketch_wide.log > gpurun_out/r05_bench_sketch_wide.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_sketch_wide.json"))
print("sketch wide:", d["ms_per_step"], d["value"], d["timed_windows"], d["loss"], (d["roofline"] or {}).get("kernel_trace", {}).get("launches_per_step"))
PY
( timeout 600 python bench.py --no-ddim --no-cpu-baseline --no-full-step --no-roofline ) 2>/dev/null | grep '^{' > gpurun_out/r05_bench_headline_6.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_headline_6.json"))
print("headline:", d["ms_per_step"], d["value"], d["timed_windows"], d["calibration"])
PY
