"""ORACLE tooling: generate ``tests/golden/*.safetensors`` by running the REFERENCE's own
``models.py`` (imported in place from /root/reference under ``oracle/diffusers_shim``; nothing is
copied) on top of the restated UNet (``oracle/unet_ref.py``).

Run in the build container only (``python -m oracle.make_golden``); the GPU box has no
/root/reference, it only reads the committed fixtures.

Each fixture holds, for one seeded small case (``oracle/cases.py``): the 4 control maps, the UNet
prediction, the MSE loss and the flat gradient of every ControlLoRA parameter after one
reference train-step forward/backward (reference ``train_text_to_image_control_lora.py:771-790``),
plus weight checksums.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

from . import cases, unet_ref
from .controllora_ref import map_processors_to_unet
from .diffusers_shim import import_reference_models

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def run_case(unet, clora, inp):
    """One reference-style train-step forward/backward in fp32 (train...:757-790)."""
    sched = unet_ref.DDPMSchedule()
    for p in unet.parameters():
        p.requires_grad_(False)
    for p in clora.parameters():
        p.requires_grad_(True)
        p.grad = None
    ctrl = clora(inp["guide"]).control_states
    noisy = sched.add_noise(inp["latents"], inp["noise"], inp["timesteps"])
    pred = unet(noisy, inp["timesteps"], inp["ehs"]).sample
    loss = F.mse_loss(pred.float(), inp["noise"].float(), reduction="mean")
    loss.backward()
    out = {f"control_{i}": c.detach().contiguous() for i, c in enumerate(ctrl)}
    out.update(pred=pred.detach().contiguous(), loss=loss.detach().reshape(1), grads=cases.flat_grads(clora))
    return out


def build(ref_models, case: str):
    unet = unet_ref.UNet2DConditionModel(**cases.SMALL_UNET)
    cases.seeded_weights_(unet, seed=11)
    if case == "lora":      # plain LoRACrossAttnProcessor on all 32 sites (dreambooth-style use of P1)
        procs = {}
        holder = torch.nn.ModuleList()
        for name in unet.attn_processors.keys():
            cad = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
            bid = int(name.split(".")[1]) if not name.startswith("mid") else 3
            hid = (list(reversed(unet.config.block_out_channels))[bid] if name.startswith("up_blocks")
                   else unet.config.block_out_channels[bid])
            p = ref_models.LoRACrossAttnProcessor(hid, cad, rank=4)
            procs[name] = p
            holder.append(p)
        clora = holder
        cases.seeded_weights_(clora, seed=23)
        unet.set_attn_processor(procs)
        clora_fwd = lambda x: type("o", (), {"control_states": ()})()
        return unet, clora, clora_fwd
    clora = ref_models.ControlLoRA(**cases.CASES[case])
    cases.seeded_weights_(clora, seed=23)
    unet.set_attn_processor(map_processors_to_unet(unet, clora))
    return unet, clora, clora


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("make_golden needs /root/reference (build container only)")
    ref_models = import_reference_models()
    os.makedirs(OUT, exist_ok=True)
    inp = cases.seeded_inputs()
    for case in list(cases.CASES) + ["lora"]:
        unet, clora, fwd = build(ref_models, case)

        class _Wrap:
            def __init__(self, mod, f):
                self.mod, self.f = mod, f

            def parameters(self):
                return self.mod.parameters()

            def named_parameters(self):
                return self.mod.named_parameters()

            def __call__(self, x):
                return self.f(x)

        res = run_case(unet, _Wrap(clora, fwd), inp)
        res["unet_checksum"] = cases.weight_checksum(unet)
        res["clora_checksum"] = cases.weight_checksum(clora)
        path = os.path.join(OUT, f"case_{case}.safetensors")
        save_file({k: v.contiguous() for k, v in res.items()}, path)
        print(f"{case:8s} loss={float(res['loss']):.6f} |pred|={float(res['pred'].norm()):.4f} "
              f"|grads|={float(res['grads'].norm()):.5f} n={res['grads'].numel()} -> {path}")


if __name__ == "__main__":
    main()
