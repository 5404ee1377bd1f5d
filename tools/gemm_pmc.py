"""Where a GEMM launch's wave-cycles go: a few (shape, tile_cfg) pairs of the train step run eagerly under
`rocprofv3 --pmc <SQ counters>`; `--summarize results.db out.json` then prints, per (kernel, grid), the counters per launch
and the derived split  parked (SQ_WAIT_ANY) / issue-stalled (SQ_WAIT_INST_ANY, of which LDS) / issuing (SQ_ACTIVE_INST_ANY),
MFMA-busy and LDS-array-busy cycles against the busy CUs' cycles.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d /tmp/x -o p -- python tools/gemm_pmc.py
    python tools/gemm_pmc.py --summarize /tmp/x/.../p_results.db gpurun_out/gemm_pmc.json"""
import json
import math
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (M, N, K, conv side H, Cin, tile_cfg, split_k): tuned choices of the train step + the alternatives worth comparing
CASES = [
    (16384, 320, 2880, 64, 320, 26, 1), (16384, 320, 2880, 64, 320, 21, 1), (16384, 320, 2880, 64, 320, 8, 1),
    (4096, 640, 5760, 32, 640, 41, 3), (4096, 640, 5760, 32, 640, 21, 2),
    (1024, 1280, 11520, 16, 1280, 21, 6),
    (16384, 320, 320, 0, 0, 33, 1), (16384, 320, 320, 0, 0, 43, 1), (1024, 1280, 1280, 0, 0, 43, 1), (4096, 640, 640, 0, 0, 43, 1),
    (16384, 2560, 320, 0, 0, 31, 1), (16384, 320, 1280, 0, 0, 26, 1),
    (8192, 8192, 8192, 0, 0, 21, 1), (8192, 8192, 8192, 0, 0, 8, 1), (8192, 8192, 8192, 0, 0, 1, 1),
]


def run():
    import torch
    from controllora_amd import kernels as K
    dev = torch.device("cuda", 0)
    for (M, N, Kd, H, Cin, cfg, sk) in CASES:
        if H:
            cd, _, _ = K.conv_fwd_desc(H, H, Cin)
            A = torch.randn(M, Cin, device=dev).half()
        else:
            cd = None
            A = torch.randn(M, Kd, device=dev).half()
        Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        for _ in range(3):
            K.gemm(A, Bw, M, N, Kd, conv=cd, out=out, split_k=sk, tile_cfg=cfg, _tuned=False)
        torch.cuda.synchronize()
        print("ran", M, N, Kd, cfg, sk, flush=True)


def summarize(db, outp):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    agg, seen = {}, set()
    for kname, grid, cname, val, dur, did in cur.execute(
            f"select kernel_name, grid_size, {name_col}, value, duration, dispatch_id from counters_collection"):
        if "gemm_dma" not in kname and "splitk" not in kname:
            continue
        key = re.sub(r"\(anonymous namespace\)::|void |\(GemmArgs.*", "", kname)[:60] + f" grid={grid}"
        a = agg.setdefault(key, {"launches": 0, "dur_ns": 0.0})
        a[cname] = a.get(cname, 0.0) + val
        if (key, did) not in seen:
            seen.add((key, did))
            a["launches"] += 1
            a["dur_ns"] += dur
    out = {}
    for k, a in agg.items():
        n = max(1, a["launches"])
        r = {c: v / n for c, v in a.items() if c not in ("launches", "dur_ns")}
        r["launches"], r["avg_us"] = a["launches"], a["dur_ns"] / n / 1e3
        wc = r.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if c in r:
                    r["frac_" + c[3:].lower()] = round(r[c] / wc, 4)
        cu = r.get("SQ_BUSY_CU_CYCLES")
        if cu:
            if "SQ_VALU_MFMA_BUSY_CYCLES" in r:
                r["mfma_busy_of_busy_cu_simds"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * cu), 4)
            if "SQ_LDS_IDX_ACTIVE" in r:
                r["lds_array_busy_of_busy_cu"] = round(r["SQ_LDS_IDX_ACTIVE"] / cu, 4)
            if "SQ_LDS_BANK_CONFLICT" in r:
                r["lds_conflict_of_busy_cu"] = round(r["SQ_LDS_BANK_CONFLICT"] / cu, 4)
        out[k] = r
        print(k, f"x{r['launches']} {r['avg_us']:.1f}us", {kk: vv for kk, vv in r.items() if kk.startswith(("frac_", "mfma_", "lds_"))})
    json.dump(out, open(outp, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run()
