"""Error budget of "denoised latents within 1e-3 rel fp16" on BASELINE's inference geometry:
the committed fp32-oracle fixture (tests/golden/full_ddim_512_50.safetensors: 50 DDIM steps, CFG 9.0, 512x512, UNet batch 4)
against (a) the product, (b) the oracle in the reference's own fp16 arithmetic, (c) the oracle with fp16 branches and an
fp32 residual trunk -- (b) and (c) run with stock torch ops on the GPU (oracle/precision_regimes.py; test infrastructure).

    python tools/error_budget.py [out.json]          (GPU box; ~2 min)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import precision_regimes as PR
    from oracle.make_fullsize_golden import DDIM_KEEP, ddim_inputs
    from tests import full_cases as F
    dev = "cuda"
    out = {"what": "rel-L2 vs the fp32 CPU oracle fixture, 50-step DDIM + CFG 9.0, 512x512, 2 images (UNet batch 4), fill50k adapters"}
    t0 = time.time()
    out["product"] = F.ddim_vs_fixture(dev, graph=True)
    out["product"]["seconds"] = time.time() - t0
    fx, meta = F.load_fixture("full_ddim_512_50.safetensors")
    o_unet, o_clora, _, _ = F.build_pair(meta["config"], dev)
    guide, cond, uncond, lat0 = ddim_inputs(int(meta["res"]), int(meta["images"]), int(meta["input_seed"]))
    for regime in ("fp16", "fp16_trunk32"):
        t0 = time.time()
        u, c = PR.build_regime(o_unet, o_clora, regime, dev)
        x, traj, eps1 = PR.ddim_loop(u, c, guide, cond, uncond, lat0, int(meta["steps"]), float(meta["guidance_scale"]), keep=DDIM_KEEP)
        r = {"latents": F.rel(x, fx["latents"]), "eps_step01": F.rel(eps1, fx["eps_step01"])}
        for i, v in traj.items():
            r[f"latents_step{i:02d}"] = F.rel(v, fx[f"latents_step{i:02d}"])
        r["seconds"] = time.time() - t0
        out[f"oracle_{regime}"] = r
        del u, c
        torch.cuda.empty_cache()
    print("ERROR_BUDGET", json.dumps(out))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
