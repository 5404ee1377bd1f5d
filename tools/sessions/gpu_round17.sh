#!/bin/bash
# gpurun call 17: per-launch durations of the attention kernels by shape (cross-attention backward looks latency-bound)
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rm -rf /tmp/attnkt && timeout 300 rocprofv3 --kernel-trace -d /tmp/attnkt -o kt -- python $R/tools/attn_ab.py > $R/gpurun_out/attn_trace.log 2>&1
cd $R
python - <<'PY' > gpurun_out/attn_trace_by_grid.txt 2>&1
import sqlite3, glob, re, collections
db = glob.glob('/tmp/attnkt/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gx = [c for c in cols if "grid" in c]
q = f"select {name}, {', '.join(gx)}, (end - start) from kernels"
agg = collections.OrderedDict()
for row in cur.execute(q):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", row[0])[:60]
    if "attn" not in n and "fill" not in n.lower():
        continue
    key = (n, tuple(row[1:-1]))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += row[-1]
for (n, g), (c, t) in agg.items():
    print(f"{t / c / 1e3:9.1f} us x{c:4d}  grid={g}  {n}")
PY
cat gpurun_out/attn_trace_by_grid.txt | cut -c1-200 | head -60
