"""ORACLE tooling (test infrastructure only): seeded small cases shared by
``oracle/make_golden.py``, the tests and ``__graft_entry__.smoke()``.

A "case" = reduced-size UNet + ControlLoRA config (same topology as SD-1.5 / the reference
configs, fewer channels) + seeded weights + seeded inputs.  Weights are drawn parameter by
parameter from an explicit ``torch.Generator`` so they regenerate identically on the GPU box;
``weight_checksum`` is stored in every golden file to make a mismatch loud.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

SMALL_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64, 128, 128), layers_per_block=2,
                  attention_head_dim=4, cross_attention_dim=64, norm_num_groups=8, norm_eps=1e-5)

_SMALL_CROSS = ([None, 64] * 5, [None, 64] * 5, [None, 64] * 5, [None, 64])

SMALL_CLORA_V1 = dict(in_channels=3, block_out_channels=(8, 16, 16, 32), norm_num_groups=8,
                      lora_block_in_channels=(32, 32, 32, 32), lora_block_out_channels=(32, 64, 128, 128),
                      lora_cross_attention_dims=_SMALL_CROSS, lora_rank=4)

SMALL_CLORA_V2 = dict(SMALL_CLORA_V1, lora_control_version=2, lora_concat_hidden=True, lora_pre_conv_skipped=True,
                      lora_key_states_skipped=True, lora_value_states_skipped=True, lora_control_self_add=False,
                      lora_control_channels=(32, 32, 32))

SMALL_CLORA_SKETCH = dict(SMALL_CLORA_V1, lora_concat_hidden=True, lora_pre_conv_skipped=True, lora_control_rank=16,
                          lora_control_self_add=False, lora_control_channels=(32, 32, 32))

SMALL_CLORA_POSTADD = dict(SMALL_CLORA_V1, lora_post_add=True)

# post_add together with concat_hidden (reference models.py:208-218 with :236-238; no shipped config, round 6)
SMALL_CLORA_POSTADD_CONCAT = dict(SMALL_CLORA_V1, lora_post_add=True, lora_concat_hidden=True, lora_pre_conv_skipped=True,
                                  lora_control_self_add=False, lora_control_channels=(32, 32, 32))

CASES = {"v1": SMALL_CLORA_V1, "v2": SMALL_CLORA_V2, "sketch": SMALL_CLORA_SKETCH, "postadd": SMALL_CLORA_POSTADD,
         "postadd_concat": SMALL_CLORA_POSTADD_CONCAT}

LATENT, RES, BATCH, CTX_LEN = 16, 128, 2, 7


def seeded_weights_(module: torch.nn.Module, seed: int, up_std: float = 0.05) -> None:
    """Deterministic fan-in init for every parameter (adapter ``up`` gets N(0, up_std) so that
    adapter bugs are visible; GroupNorm/LayerNorm affine gets (1 + small, small))."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.ndim >= 2:
                std = up_std if name.endswith(".up.weight") else 1.0 / math.sqrt(p[0].numel())
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


def weight_checksum(module: torch.nn.Module) -> torch.Tensor:
    s = torch.zeros(2, dtype=torch.float64)
    for _, p in module.named_parameters():
        s[0] += p.detach().double().sum()
        s[1] += p.detach().double().abs().sum()
    return s


def seeded_inputs(seed: int = 7, batch: int = BATCH, ctx_dim: int = 64) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return dict(
        guide=torch.rand(batch, 3, RES, RES, generator=g) * 2 - 1,
        latents=torch.randn(batch, 4, LATENT, LATENT, generator=g),
        noise=torch.randn(batch, 4, LATENT, LATENT, generator=g),
        timesteps=torch.randint(0, 1000, (batch,), generator=g),
        ehs=torch.randn(batch, CTX_LEN, ctx_dim, generator=g),
    )


def flat_grads(module: torch.nn.Module) -> torch.Tensor:
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float()
                      for _, p in module.named_parameters()])


def build_oracle_case(case: str):
    """Oracle-side twin of ``oracle/make_golden.build`` (uses controllora_ref, not the reference)."""
    from . import unet_ref
    from .controllora_ref import ControlLoRARef, LoRAProcRef, map_processors_to_unet
    unet = unet_ref.UNet2DConditionModel(**SMALL_UNET)
    seeded_weights_(unet, seed=11)
    if case == "lora":
        procs, holder = {}, torch.nn.ModuleList()
        boc = unet.config.block_out_channels
        for name in unet.attn_processors.keys():
            cad = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
            bid = int(name.split(".")[1]) if not name.startswith("mid") else 3
            hid = list(reversed(boc))[bid] if name.startswith("up_blocks") else boc[bid]
            p = LoRAProcRef(hid, cad, rank=4)
            procs[name] = p
            holder.append(p)
        seeded_weights_(holder, seed=23)
        unet.set_attn_processor(procs)
        return unet, holder, None
    clora = ControlLoRARef(**CASES[case])
    seeded_weights_(clora, seed=23)
    unet.set_attn_processor(map_processors_to_unet(unet, clora))
    return unet, clora, clora


def oracle_train_step(unet, params_module, clora_fwd, inp):
    """fp32 CPU restatement of the reference step body (train...:757-790): returns dict like the goldens."""
    import torch.nn.functional as F
    from .unet_ref import DDPMSchedule
    for p in unet.parameters():
        p.requires_grad_(False)
    for p in params_module.parameters():
        p.requires_grad_(True)
        p.grad = None
    ctrl = clora_fwd(inp["guide"]).control_states if clora_fwd is not None else ()
    noisy = DDPMSchedule().add_noise(inp["latents"], inp["noise"], inp["timesteps"])
    pred = unet(noisy, inp["timesteps"], inp["ehs"]).sample
    loss = F.mse_loss(pred.float(), inp["noise"].float(), reduction="mean")
    loss.backward()
    out = {f"control_{i}": c.detach() for i, c in enumerate(ctrl)}
    out.update(pred=pred.detach(), loss=loss.detach().reshape(1), grads=flat_grads(params_module))
    return out
