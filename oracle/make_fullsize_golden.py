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

  tests/golden/full_train_512_bs8_v2.safetensors   (round 4)
      the same train-step record for BASELINE configs[3] as it is quoted: configs/mpii-pose-v2.json (v2 "decomposed" processors,
      reference models.py:292-431), SD-1.5 topology, 512x512, batch 8.
  tests/golden/full_infer_512_b32.safetensors      (round 4)
      BASELINE config 5 at its OWN batch: 16 images => UNet batch 32 (reference apps/gradio_canny2image.py:83-89), one guide image
      broadcast over the batch, 50-step DDIM schedule, CFG 9.0: the first UNet evaluation (all 32 samples) and the latents after
      steps 1 and 5.  The oracle evaluates the batch in chunks of 4 samples (the UNet is sample-independent; 17 GB of fp32
      attention scores otherwise).
  tests/golden/full_vae_512.safetensors            (round 4)
      oracle/vae_ref.py (SD-1.5 VAE topology, real widths) at 512x512, batch 1: moments, sampled latents, decoded image
      (reference train...:753-754, apps/gradio_canny2image.py:88-92).

The hint encoder + adapters run through the REFERENCE's own `ControlLoRA` (reference models.py, imported in place under
oracle/diffusers_shim, never copied) whenever /root/reference is present -- i.e. always when the fixtures are (re)generated in the
build container; metadata "clora_impl" records which implementation produced the file.  Its weights are the restatement's seeded
ones (strict state-dict load: identical key sets), so the GPU box regenerates them from seeds without the reference.

Big tensors carry TWO strided samples with coprime strides (8 and 13 for the gradient, 16 and 31 for the control maps): an error
confined to positions congruent mod one stride cannot hide from the other.

Weights and inputs are the seeded ones of tests/full_cases.py (regenerated on the GPU box from the same seeds; their
checksums are stored so a mismatch is loud).  The oracle's frozen weights and inputs are fp16-rounded values in fp32
(SURVEY.md section 8c "Tolerance reading").

    python -m oracle.make_fullsize_golden [train] [train_v2] [ddim] [infer32] [vae] [--threads N]
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
STRIDE_GRAD2 = 13         # second, coprime sample
STRIDE_CTRL2 = 31
TRAIN_V2_FILE = "full_train_512_bs8_v2.safetensors"
INFER32_FILE = "full_infer_512_b32.safetensors"
VAE_FILE = "full_vae_512.safetensors"
INFER32_SEED, INFER32_IMAGES, INFER32_KEEP = 6, 16, (1, 5)
VAE_SEED, VAE_STRIDE, VAE_STRIDE2 = 3, 4, 5
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


def build_oracle_pair(config_name):
    """(oracle UNet, the module whose forward/parameters define the fixture, impl name).  With /root/reference present the module is
    the REFERENCE's `ControlLoRA` carrying the restatement's seeded weights; the processors it hands to the UNet are the reference's."""
    from oracle.controllora_ref import ControlLoRARef, map_processors_to_unet, randomize_adapters_
    from tests import full_cases as F
    o_unet = F.oracle_unet_sd15()
    torch.manual_seed(1)
    path = os.path.join(ROOT, "configs", config_name)
    o_clora = ControlLoRARef.from_config(path)
    randomize_adapters_(o_clora, seed=1, std=0.02)
    impl = "restatement"
    if os.path.isdir("/root/reference"):
        from oracle.diffusers_shim import import_reference_models
        ref_models = import_reference_models()
        r_clora = ref_models.ControlLoRA.from_config(path)
        r_clora.load_state_dict(o_clora.state_dict(), strict=True)
        assert [n for n, _ in r_clora.named_parameters()] == [n for n, _ in o_clora.named_parameters()]
        o_clora, impl = r_clora, "reference"
    o_unet.set_attn_processor(map_processors_to_unet(o_unet, o_clora))
    return o_unet, o_clora, impl


def train_record(o_unet, o_clora, inp):
    """one reference train step -> the tensors a train fixture stores (shared by make_train and the reference-in-place test)"""
    from oracle import cases
    t0 = time.time()
    gold = cases.oracle_train_step(o_unet, o_clora, o_clora, inp)
    dt = time.time() - t0
    out = {"pred": gold["pred"].float().contiguous(), "loss": gold["loss"].float(),
           "grads_sample": sample(gold["grads"], STRIDE_GRAD), "grads_sample2": sample(gold["grads"], STRIDE_GRAD2),
           "grads_norm": norm64(gold["grads"]),
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
        out[f"control_{i}_sample2"] = sample(c, STRIDE_CTRL2)
        out[f"control_{i}_norm"] = norm64(c)
    for k in ("guide", "latents", "noise", "ehs"):
        out[f"in_{k}_checksum"] = checksum(inp[k])
    out["in_timesteps"] = inp["timesteps"].clone()
    return out, names, gold


def make_train(config="fill50k.json", batch=4, fname=TRAIN_FILE, input_seed=7):
    from safetensors.torch import save_file
    from tests import full_cases as F
    o_unet, o_clora, impl = build_oracle_pair(config)
    inp = F.inputs(512, batch, seed=input_seed)         # fill50k / bs 4 / seed 7: the inputs of full_cases.full_size_properties
    out, names, gold = train_record(o_unet, o_clora, inp)
    save_file(out, os.path.join(GOLD, fname),
              metadata={"stride_grad": str(STRIDE_GRAD), "stride_ctrl": str(STRIDE_CTRL), "stride_grad2": str(STRIDE_GRAD2),
                        "stride_ctrl2": str(STRIDE_CTRL2), "config": config, "res": "512", "batch": str(batch),
                        "input_seed": str(input_seed), "clora_impl": impl, "param_names": "\n".join(names)})
    print(f"train step {config} 512^2 bs{batch} ({impl} ControlLoRA): oracle {float(out['oracle_seconds']):.1f} s, "
          f"loss {float(gold['loss']):.6f}, |grads| {float(out['grads_norm']):.4e}", flush=True)


def infer32_inputs(res=512, nb=INFER32_IMAGES, seed=INFER32_SEED):
    return ddim_inputs(res, nb, seed)


@torch.no_grad()
def make_infer32(chunk=4):
    from safetensors.torch import save_file
    from oracle import cases, unet_ref
    o_unet, o_clora, impl = build_oracle_pair("fill50k.json")
    guide, cond, uncond, lat0 = infer32_inputs()
    sch = unet_ref.DDPMSchedule()
    o_clora(guide)                                       # ONE guide: control batch 1, broadcast over the UNet batch (quirk C6)
    ehs = torch.cat([uncond, cond], 0)
    x = lat0.clone()
    out = {}
    t0 = time.time()
    for i, t in enumerate(sch.ddim_timesteps(DDIM_STEPS), 1):
        xin = torch.cat([x, x], 0)
        eps = torch.cat([o_unet(xin[j:j + chunk], t, ehs[j:j + chunk]).sample for j in range(0, xin.shape[0], chunk)], 0)
        eu, ec = eps.chunk(2)
        x = sch.ddim_step(eu + DDIM_SCALE * (ec - eu), t, x, DDIM_STEPS)
        if i == 1:
            out["eps_step01"] = eps.clone().contiguous()
        if i in INFER32_KEEP:
            out[f"latents_step{i:02d}"] = x.clone().contiguous()
        print(f"infer32 step {i} t={int(t)} |x|={float(x.norm()):.4f} ({time.time() - t0:.0f} s)", flush=True)
        if i == max(INFER32_KEEP):
            break
    out["weights_checksum_unet"] = cases.weight_checksum(o_unet)
    out["weights_checksum_clora"] = cases.weight_checksum(o_clora)
    for k, v in (("guide", guide), ("cond", cond), ("uncond", uncond), ("lat0", lat0)):
        out[f"in_{k}_checksum"] = checksum(v)
    out["oracle_seconds"] = torch.tensor([time.time() - t0])
    save_file(out, os.path.join(GOLD, INFER32_FILE),
              metadata={"config": "fill50k.json", "res": "512", "images": str(INFER32_IMAGES), "steps": str(DDIM_STEPS),
                        "keep": ",".join(map(str, INFER32_KEEP)), "guidance_scale": str(DDIM_SCALE), "input_seed": str(INFER32_SEED),
                        "clora_impl": impl})
    print(f"infer32: {time.time() - t0:.0f} s", flush=True)


def vae_oracle(seed=VAE_SEED):
    """SD-1.5 VAE topology at real widths with the seeded weights of tests/vae_cases.check_vae (same draw order)."""
    from controllora_amd import vae as V
    from oracle import vae_ref as R
    torch.manual_seed(seed)
    o = R.AutoencoderKL(**V.SD15_VAE)
    with torch.no_grad():
        for n, p in o.named_parameters():
            if p.ndim == 1:
                p.copy_((0.2 * torch.randn_like(p) + (1.0 if "norm" in n and n.endswith("weight") else 0.0)))
            p.copy_(p.half().float())
    return o


def vae_inputs(res=512, batch=1):
    """drawn right after vae_oracle()'s weights from the same global generator, like tests/vae_cases.check_vae"""
    x = (torch.rand(batch, 3, res, res) * 2 - 1).half().float()
    eps = torch.randn(batch, 4, res // 8, res // 8)
    return x, eps


@torch.no_grad()
def make_vae(res=512, batch=1):
    from safetensors.torch import save_file
    from oracle import cases
    o = vae_oracle()
    x, eps = vae_inputs(res, batch)
    t0 = time.time()
    mean, logvar = o.moments(x)
    z = o.encode_sample(x, eps)
    img = o.decode(z.half().float())
    dt = time.time() - t0
    out = {"mean": mean.contiguous(), "logvar": logvar.contiguous(), "z": z.contiguous(),
           "img_sample": sample(img, VAE_STRIDE), "img_sample2": sample(img, VAE_STRIDE2), "img_norm": norm64(img),
           "weights_checksum": cases.weight_checksum(o), "in_x_checksum": checksum(x), "in_eps_checksum": checksum(eps),
           "oracle_seconds": torch.tensor([dt])}
    save_file(out, os.path.join(GOLD, VAE_FILE), metadata={"res": str(res), "batch": str(batch), "seed": str(VAE_SEED),
                                                           "stride": str(VAE_STRIDE), "stride2": str(VAE_STRIDE2)})
    print(f"vae {res}^2 bs{batch}: oracle {dt:.1f} s, |mean| {float(mean.norm()):.4f} |img| {float(img.norm()):.4f}", flush=True)


@torch.no_grad()
def make_ddim():
    from safetensors.torch import save_file
    from oracle import cases, unet_ref
    o_unet, o_clora, impl = build_oracle_pair("fill50k.json")
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
                        "guidance_scale": str(DDIM_SCALE), "input_seed": str(DDIM_SEED), "clora_impl": impl})
    print(f"ddim 50 steps 512^2 x{DDIM_IMAGES}: {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--threads" in args:
        torch.set_num_threads(int(args[args.index("--threads") + 1]))
    known = ("train", "train_v2", "ddim", "infer32", "vae")
    what = [a for a in args if a in known] or list(known)
    if "vae" in what:
        make_vae()
    if "train" in what:
        make_train()
    if "train_v2" in what:
        make_train("mpii-pose-v2.json", 8, TRAIN_V2_FILE, input_seed=9)
    if "infer32" in what:
        make_infer32()
    if "ddim" in what:
        make_ddim()
