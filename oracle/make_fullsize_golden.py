"""ORACLE tooling (test infrastructure only; never imported by the product path).

Fixtures at BASELINE.json's OWN sizes (VERDICT r02 "Next round" item 1), written once in the build container
because the CPU oracle needs minutes per step at 512x512:

  tests/golden/full_train_512_bs4.safetensors
      one reference train step (reference train_text_to_image_control_lora.py:751-796) of configs/fill50k.json on
      the SD-1.5 topology at 512x512, batch 4 (BASELINE configs[1]): UNet prediction, loss, the four control maps
      and the flat gradient of all 6,047,040 trainable parameters.  The big tensors are stored as a fixed strided
      sample (every STRIDE-th element of the flattened tensor, fp32) plus their full fp64 norms and per-parameter
      norms: a rel-L2 over 1/STRIDE of the elements is the same statistic, per-tensor norms keep localised errors
      visible, and the file stays a few MB.
  tests/golden/full_ddim_512_50.safetensors
      the inference call pattern (reference apps/gradio_canny2image.py:83-89; BASELINE config 5 geometry with 2
      images instead of 16): hint-encode ONE guide, 50 DDIM steps (eta 0) with classifier-free guidance 9.0 at
      512x512, UNet batch 4 (uncond first); final latents and the trajectory at steps 1, 2, 5, 10, 20, 30, 40, 50.

Weights and inputs are the seeded ones of tests/full_cases.py (regenerated on the GPU box from the same seeds; their
checksums are stored so a mismatch is loud).  The oracle's frozen weights and inputs are fp16-rounded values in fp32
(SURVEY.md section 8c "Tolerance reading").

    python -m oracle.make_fullsize_golden [train] [ddim] [--threads N]
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
STRIDE_GRAD = 8           # 6,047,040 / 8 = 755,880 sampled gradient elements
STRIDE_CTRL = 16
TRAIN_FILE = "full_train_512_bs4.safetensors"
DDIM_FILE = "full_ddim_512_50.safetensors"
DDIM_KEEP = (1, 2, 5, 10, 20, 30, 40, 50)
DDIM_SEED, DDIM_IMAGES, DDIM_STEPS, DDIM_SCALE = 5, 2, 50, 9.0


def checksum(t: torch.Tensor) -> torch.Tensor:
    t = t.detach().double().reshape(-1)
    return torch.stack([t.sum(), t.abs().sum()])


def sample(t: torch.Tensor, stride: int) -> torch.Tensor:
    return t.detach().reshape(-1)[::stride].float().contiguous()


def norm64(t: torch.Tensor) -> torch.Tensor:
    return t.detach().double().norm().reshape(1)


def ddim_inputs(res=512, nb=DDIM_IMAGES, seed=DDIM_SEED, ctx_len=77, ctx_dim=768):
    """Same draw order as tests/full_cases.ddim_parity."""
    g = torch.Generator().manual_seed(seed)
    L = res // 8
    guide = ((torch.rand(1, 3, res, res, generator=g) > 0.9).float() * 2 - 1)
    cond = torch.randn(nb, ctx_len, ctx_dim, generator=g).half().float()
    uncond = torch.randn(nb, ctx_len, ctx_dim, generator=g).half().float()
    lat0 = torch.randn(nb, 4, L, L, generator=g).half().float()
    return guide, cond, uncond, lat0


def make_train():
    from safetensors.torch import save_file
    from oracle import cases
    from oracle.controllora_ref import ControlLoRARef, map_processors_to_unet, randomize_adapters_
    from tests import full_cases as F
    o_unet = F.oracle_unet_sd15()
    torch.manual_seed(1)
    o_clora = ControlLoRARef.from_config(os.path.join(ROOT, "configs", "fill50k.json"))
    randomize_adapters_(o_clora, seed=1, std=0.02)
    o_unet.set_attn_processor(map_processors_to_unet(o_unet, o_clora))
    inp = F.inputs(512, 4, seed=7)                      # the inputs of full_cases.full_size_properties
    t0 = time.time()
    gold = cases.oracle_train_step(o_unet, o_clora, o_clora, inp)
    dt = time.time() - t0
    out = {"pred": gold["pred"].float().contiguous(), "loss": gold["loss"].float(),
           "grads_sample": sample(gold["grads"], STRIDE_GRAD), "grads_norm": norm64(gold["grads"]),
           "weights_checksum_unet": cases.weight_checksum(o_unet), "weights_checksum_clora": cases.weight_checksum(o_clora),
           "oracle_seconds": torch.tensor([dt])}
    names, norms, off = [], [], 0
    for n, p in o_clora.named_parameters():
        k = p.numel()
        norms.append(gold["grads"][off:off + k].double().norm())
        names.append(n)
        off += k
    out["grads_param_norms"] = torch.stack(norms)
    for i in range(4):
        c = gold[f"control_{i}"]
        out[f"control_{i}_sample"] = sample(c, STRIDE_CTRL)
        out[f"control_{i}_norm"] = norm64(c)
    for k in ("guide", "latents", "noise", "ehs"):
        out[f"in_{k}_checksum"] = checksum(inp[k])
    out["in_timesteps"] = inp["timesteps"].clone()
    save_file(out, os.path.join(GOLD, TRAIN_FILE),
              metadata={"stride_grad": str(STRIDE_GRAD), "stride_ctrl": str(STRIDE_CTRL), "config": "fill50k.json",
                        "res": "512", "batch": "4", "input_seed": "7", "param_names": "\n".join(names)})
    print(f"train step 512^2 bs4: oracle {dt:.1f} s, loss {float(gold['loss']):.6f}, |grads| {float(out['grads_norm']):.4e}",
          flush=True)


@torch.no_grad()
def make_ddim():
    from safetensors.torch import save_file
    from oracle import cases, unet_ref
    from oracle.controllora_ref import ControlLoRARef, map_processors_to_unet, randomize_adapters_
    from tests import full_cases as F
    o_unet = F.oracle_unet_sd15()
    torch.manual_seed(1)
    o_clora = ControlLoRARef.from_config(os.path.join(ROOT, "configs", "fill50k.json"))
    randomize_adapters_(o_clora, seed=1, std=0.02)
    o_unet.set_attn_processor(map_processors_to_unet(o_unet, o_clora))
    guide, cond, uncond, lat0 = ddim_inputs()
    sch = unet_ref.DDPMSchedule()
    o_clora(guide)
    ehs = torch.cat([uncond, cond], 0)
    x = lat0.clone()
    out = {}
    t0 = time.time()
    for i, t in enumerate(sch.ddim_timesteps(DDIM_STEPS), 1):
        eps = o_unet(torch.cat([x, x], 0), t, ehs).sample
        eu, ec = eps.chunk(2)
        x = sch.ddim_step(eu + DDIM_SCALE * (ec - eu), t, x, DDIM_STEPS)
        if i in DDIM_KEEP:
            out[f"latents_step{i:02d}"] = x.clone().contiguous()
        if i == 1:
            out["eps_step01"] = eps.clone().contiguous()          # single-forward error, before any sampler feedback
        print(f"ddim step {i}/{DDIM_STEPS} t={int(t)} |x|={float(x.norm()):.4f} ({time.time() - t0:.0f} s)", flush=True)
    out["latents"] = x.contiguous()
    out["weights_checksum_unet"] = cases.weight_checksum(o_unet)
    out["weights_checksum_clora"] = cases.weight_checksum(o_clora)
    for k, v in (("guide", guide), ("cond", cond), ("uncond", uncond), ("lat0", lat0)):
        out[f"in_{k}_checksum"] = checksum(v)
    out["oracle_seconds"] = torch.tensor([time.time() - t0])
    save_file(out, os.path.join(GOLD, DDIM_FILE),
              metadata={"config": "fill50k.json", "res": "512", "images": str(DDIM_IMAGES), "steps": str(DDIM_STEPS),
                        "guidance_scale": str(DDIM_SCALE), "input_seed": str(DDIM_SEED)})
    print(f"ddim 50 steps 512^2 x{DDIM_IMAGES}: {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--threads" in args:
        torch.set_num_threads(int(args[args.index("--threads") + 1]))
    what = [a for a in args if a in ("train", "ddim")] or ["train", "ddim"]
    if "train" in what:
        make_train()
    if "ddim" in what:
        make_ddim()
