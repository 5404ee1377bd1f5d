"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite .db): how much of the
replayed hipGraph step is launch gap rather than kernel time.  usage: python tools/rocprof_gaps.py results.db [tail_fraction]"""
import sqlite3
import sys


def main(db, tail=0.4):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end from kernels order by start").fetchall()
    rows = rows[int(len(rows) * (1 - tail)):]                    # steady state: the last part of the run (graph replays)
    busy = sum(e - s for s, e in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
    overlap = sum(max(0, rows[i][1] - rows[i + 1][0]) for i in range(len(rows) - 1))
    gs = sorted(gaps)
    print(f"{len(rows)} kernels, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms ({100 * busy / span:.1f}%), "
          f"gaps {sum(gaps) / 1e6:.2f} ms, overlap {overlap / 1e6:.2f} ms")
    print(f"gap per launch: median {gs[len(gs) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gs[int(len(gs) * 0.9)] / 1e3:.2f} us, "
          f"max {gs[-1] / 1e3:.1f} us;  gaps > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)
