"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel table (markdown + json).
usage: python tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats [steps]"""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]


def load(db):
    """per-kernel rows (name, calls, total / avg / min / max time) of a rocprofv3 --kernel-trace rocpd database"""
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    dur = "(end - start)" if "end" in cols and "start" in cols else "duration"
    rows = cur.execute(f"select {name_col}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels "
                       f"group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    return [dict(kernel=short(r[0]), calls=r[1], total_ms=round(r[2] / 1e6, 3), avg_us=round(r[3] / 1e3, 2),
                 min_us=round(r[4] / 1e3, 2), max_us=round(r[5] / 1e3, 2), pct=round(100.0 * r[2] / total, 2)) for r in rows]


def main(db, out, steps=None):
    table = load(db)
    total = sum(t["total_ms"] for t in table) * 1e6
    with open(out + ".json", "w") as f:
        json.dump(dict(total_kernel_ms=round(total / 1e6, 3), steps=steps, kernels=table), f, indent=1)
    with open(out + ".md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace summary ({db})\n\n")
        f.write(f"total kernel time {total / 1e6:.2f} ms over the whole run"
                + (f" ({steps} train steps incl. warm-up -> {total / 1e6 / steps:.2f} ms/step of kernel time)" if steps else "") + "\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for t in table[:60]:
            f.write(f"| `{t['kernel']}` | {t['calls']} | {t['total_ms']} | {t['avg_us']} | {t['min_us']} | {t['max_us']} | {t['pct']} |\n")
    for t in table[:28]:
        print(f"{t['pct']:6.2f}% {t['total_ms']:9.3f} ms {t['calls']:6d} x {t['avg_us']:9.2f} us  {t['kernel'][:100]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
