"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE).
usage: python tools/pmc_mfma.py results.db out.json
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CU_CYCLES)   (fraction of the busy CUs' MFMA issue capacity)
and, independent of which CUs were busy,  MFMA busy cycles / (kernel duration * 2.4 GHz * 256 CUs * 4 SIMDs)."""
import json
import re
import sqlite3
import sys


def main(db, outp):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", cols)
    name_col = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    agg = {}
    for kname, cname, val, dur in cur.execute(f"select kernel_name, {name_col}, value, duration from counters_collection"):
        kname = re.sub(r"\(anonymous namespace\)::", "", kname)[:100]
        a = agg.setdefault(kname, {"launches": 0, "dur_ns": 0.0})
        a[cname] = a.get(cname, 0.0) + val
        if cname == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["launches"] += 1
            a["dur_ns"] += dur
    res = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        mf, cu = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("SQ_BUSY_CU_CYCLES", 0.0)
        if mf <= 0:
            continue
        res[k] = {"launches": a["launches"], "mfma_busy_cycles_per_launch": mf / max(a["launches"], 1),
                  "mfma_util_of_busy_cus": (mf / (4.0 * cu)) if cu else None,
                  "mfma_util_of_chip_time": mf / (a["dur_ns"] * 2.4 * 256 * 4) if a["dur_ns"] else None,
                  "raw": {c: v for c, v in a.items() if c not in ("launches", "dur_ns")}}
    json.dump(res, open(outp, "w"), indent=1)
    for k, v in list(res.items())[:12]:
        print(f"{v['launches']:5d}x  util(busy CUs) {v['mfma_util_of_busy_cus']}  util(chip time) {v['mfma_util_of_chip_time']:.3f}  {k[:70]}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
