"""Per-kernel sums of every counter of one rocprofv3 --pmc pass (rocpd sqlite) -> json + a short table.
usage: python tools/pmc_dump.py results.db out.json [name-filter]"""
import json
import re
import sqlite3
import sys


def main(db, outp, filt=None):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    agg = {}
    seen = {}
    for kname, cname, val, dur, did in cur.execute(f"select kernel_name, {name_col}, value, duration, dispatch_id from counters_collection"):
        kname = re.sub(r"\(anonymous namespace\)::", "", kname)[:110]
        if filt and filt not in kname:
            continue
        a = agg.setdefault(kname, {"launches": 0, "dur_ns": 0.0})
        a[cname] = a.get(cname, 0.0) + val
        if (kname, did) not in seen:
            seen[(kname, did)] = 1
            a["launches"] += 1
            a["dur_ns"] += dur
    json.dump(agg, open(outp, "w"), indent=1)
    for k, a in agg.items():
        n = max(1, a["launches"])
        print(k[:90], f"x{a['launches']} avg {a['dur_ns'] / n / 1e3:.1f}us")
        print("   " + "  ".join(f"{c}={v / n:.3g}" for c, v in a.items() if c not in ("launches", "dur_ns")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
