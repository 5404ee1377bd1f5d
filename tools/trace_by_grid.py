"""Aggregate a rocprofv3 --kernel-trace database by (kernel, grid): per-launch average / total -- the per-shape view the plain
kernel table hides (a kernel template serves very different problem sizes).
usage: python tools/trace_by_grid.py results.db out.txt [steps] [top]"""
import collections
import re
import sqlite3
import sys


def main(db, outp, steps=1, top=90):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gx = [c for c in cols if "grid" in c]
    agg = collections.OrderedDict()
    for row in cur.execute(f"select {name}, {', '.join(gx)}, (end - start) from kernels"):
        n = re.sub(r"\(anonymous namespace\)::|void ", "", row[0])[:70]
        a = agg.setdefault((n, tuple(row[1:-1])), [0, 0.0])
        a[0] += 1
        a[1] += row[-1]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(outp, "w") as f:
        tot = sum(t for _, (c, t) in rows) / 1e6 / steps
        f.write(f"total kernel time {tot:.3f} ms per step over {steps} steps\n")
        for (n, g), (c, t) in rows[:top]:
            f.write(f"{t / 1e6 / steps:8.3f} ms/step {c / steps:7.1f}x {t / c / 1e3:9.1f} us  grid={g}  {n}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1, int(sys.argv[4]) if len(sys.argv) > 4 else 90)
