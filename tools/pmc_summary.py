"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite).
usage: python tools/pmc_summary.py fetch_results.db write_results.db out.json
Units: the counters are in KiB.  Calibration (MI355X_MICROARCH.md section HBM: FETCH_SIZE under-counts wide
streaming reads by 2x on gfx950) is checked in-place on a kernel with a known read volume (the first
GroupNorm statistics pass of the hint encoder reads exactly B*H*W*C*2 bytes once)."""
import json
import re
import sqlite3
import sys


def per_kernel(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, grid, val, dur in cur.execute("select kernel_name, grid_size, value, duration from counters_collection"):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        out.setdefault(name, []).append((grid, val * 1024.0, dur))
    return out


def main(fdb, wdb, outp):
    F, W = per_kernel(fdb), per_kernel(wdb)
    # calibration: largest gn_fwd_partial launch = hint encoder, 4 x 512 x 512 x 32 fp16 = 67.1 MB read once
    cal = max(v for k, vs in F.items() if "gn_fwd_partial" in k for _, v, _ in vs)
    known = 4 * 512 * 512 * 32 * 2
    factor = known / cal
    res = {"calibration": {"kernel": "gn_fwd_partial_kernel (hint encoder level 0)", "known_read_bytes": known,
                           "FETCH_SIZE_bytes": cal, "read_correction_factor": round(factor, 3)}, "kernels": {}}
    for k in sorted(F, key=lambda k: -sum(v for _, v, _ in F[k])):
        fv = [v for _, v, _ in F[k]]
        wv = [v for _, v, _ in W.get(k, [])]
        res["kernels"][k[:100]] = {"launches": len(fv), "fetch_bytes_per_launch_raw": sum(fv) / len(fv),
                                   "fetch_bytes_per_launch_corrected": sum(fv) / len(fv) * factor,
                                   "write_bytes_per_launch_raw": (sum(wv) / len(wv)) if wv else None}
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps(res["calibration"]))
    for k, v in list(res["kernels"].items())[:14]:
        print(f"{v['launches']:5d}x fetch {v['fetch_bytes_per_launch_corrected']/1e6:9.2f} MB (raw {v['fetch_bytes_per_launch_raw']/1e6:8.2f})  "
              f"write {0 if v['write_bytes_per_launch_raw'] is None else v['write_bytes_per_launch_raw']/1e6:8.2f} MB  {k[:70]}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
