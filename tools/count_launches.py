"""Dry-run launch census of one train step (CPU, no GPU, no kernel executes): every C-ABI entry point is replaced by a counter, the
host code (autograd functions, launch planning in Python) runs for real on CPU tensors of the real shapes.

    python tools/count_launches.py [--config fill50k.json] [--batch 4] [--res 512]

Counts C-ABI CALLS (a split-K GEMM call = GEMM + finish kernel, a two-pass GroupNorm call = 2 kernels: those are decided inside the
library; the rocprof kernel trace of bench.py is the authority for kernel launches).  Used to check a host-side merge (grouped
projections, multi-job launches, removed copies) before spending GPU minutes."""
from __future__ import annotations

import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="fill50k.json")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--detail", action="store_true")
    a = ap.parse_args()
    from controllora_amd import capi
    import bench
    from controllora_amd import kernels as K
    from controllora_amd.train import ControlLoRATrainer

    lib = capi.Lib(capi.LIB_PATH, require_device=False)
    counts, phase = collections.Counter(), ["setup"]
    real_call = lib.call

    seq = []

    def fake_call(name, *args):
        if name in ("clora_set_option", "clora_comm_library"):
            return real_call(name, *args)
        counts[(phase[0], name)] += 1
        if a.detail and phase[0] == "step":
            if name == "clora_gemm_f16_ex":
                A_, lda, B_, C_, ldc, M, N, K_, conv, epi, split, cfg = args[:12]
                e = epi._obj if hasattr(epi, "_obj") else None
                cv = conv._obj if conv is not None and hasattr(conv, "_obj") else None
                seq.append(("gemm", M, N, K_, "conv" if cv is not None else "", split, cfg, A_, C_,
                            "res" if (e is not None and e.residual) else "", "rowadd" if (e is not None and e.rowadd) else "",
                            "lora" if (e is not None and e.lora_t) else "", "geglu%d" % e.geglu if (e is not None and e.geglu) else ""))
            elif name == "clora_groupnorm_fwd_f16":
                seq.append(("gn_fwd", args[5], args[6], args[7], args[0], args[1]))
            elif name == "clora_groupnorm_bwd_f16":
                seq.append(("gn_bwd", args[9], args[10], args[11], args[1], args[3]))
            elif name == "clora_layernorm_fwd_f16":
                seq.append(("ln_fwd", args[4], args[5], args[0], args[1]))
            elif name == "clora_layernorm_bwd_f16":
                seq.append(("ln_bwd", args[5], args[6], args[1], args[3]))
            else:
                seq.append((name.replace("clora_", ""),))

    lib.call = fake_call
    capi._LIB = lib
    torch.manual_seed(0)
    unet, clora = bench.build_models("cpu", config=a.config)
    inp = bench.synthetic_batch(a.batch, a.res, "cpu", 42)
    if inp is None:
        raise SystemExit("bench.synthetic_batch not found")
    tr = ControlLoRATrainer(unet, clora)
    noisy = inp["latents"]
    for ph in ("warm", "step"):
        phase[0] = ph
        cfb = sum(counts.values())
        tr.forward_backward(noisy, inp["timesteps"], inp["ehs"], inp["guide"], inp["noise"])
        n_fb = sum(counts.values()) - cfb
        tr.optimizer_step()
    tot = sum(v for (ph, _), v in counts.items() if ph == "step")
    print(f"config {a.config} batch {a.batch} res {a.res}: {tot} C-ABI calls per step ({n_fb} forward/backward)")
    by = collections.Counter()
    for (ph, name), v in counts.items():
        if ph == "step":
            by[name] += v
    for name, v in by.most_common():
        print(f"  {v:5d}  {name}")
    if a.detail:
        for rec in seq:
            print("  ".join(str(x) for x in rec))


if __name__ == "__main__":
    main()
