#!/bin/bash
# gpurun call 18: cross-attention dK/dV splits folded from slabs instead of atomics: tests, per-grid trace, same-box bench A/B
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" -p no:cacheprovider ) > gpurun_out/gputest_attn2.log 2>&1
tail -2 gpurun_out/gputest_attn2.log
B="--no-cpu-baseline --no-full-step --steps 30 --no-roofline --no-ddim"
( CLORA_LIB_PATH=$PWD/controllora_amd/_build_prev/libclora.so timeout 900 python bench.py $B ) > gpurun_out/bench_r18_prev.log 2>&1
( timeout 900 python bench.py $B ) > gpurun_out/bench_r18_new.log 2>&1
for f in gpurun_out/bench_r18_prev.log gpurun_out/bench_r18_new.log; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
cd /tmp && rm -rf /tmp/attnkt && timeout 300 rocprofv3 --kernel-trace -d /tmp/attnkt -o kt -- python $R/tools/attn_ab.py > $R/gpurun_out/attn_trace.log 2>&1
cd $R
python - <<'PY' > gpurun_out/attn_trace_by_grid_after.txt 2>&1
import sqlite3, glob, re, collections
db = glob.glob('/tmp/attnkt/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gx = [c for c in cols if "grid" in c]
agg = collections.OrderedDict()
for row in cur.execute(f"select {name}, {', '.join(gx)}, (end - start) from kernels"):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", row[0])[:60]
    if "attn" not in n and "fill" not in n.lower():
        continue
    a = agg.setdefault((n, tuple(row[1:-1])), [0, 0.0])
    a[0] += 1; a[1] += row[-1]
for (n, g), (c, t) in agg.items():
    print(f"{t / c / 1e3:9.1f} us x{c:4d}  grid={g}  {n}")
PY
cat gpurun_out/attn_trace_by_grid_after.txt | cut -c1-160 | head -30
