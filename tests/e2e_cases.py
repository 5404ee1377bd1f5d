"""End-to-end parity of the product path (controllora_amd: HIP kernels through the C ABI) against the
CPU oracle and the committed golden vectors made by the reference's own models.py.  Shared by the CPU
(emulated kernels) and GPU test modules."""
import os

import torch
from safetensors.torch import load_file

from controllora_amd import models as M
from controllora_amd import unet as U
from controllora_amd.train import ControlLoRATrainer
from oracle import cases, unet_ref

f16 = torch.float16


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def build_product_case(case, dev):
    """Product UNet + ControlLoRA carrying exactly the oracle's seeded weights (fp16-rounded for the UNet)."""
    o_unet, o_params, _ = cases.build_oracle_case(case)
    unet = U.UNet2DConditionModel(**cases.SMALL_UNET)
    U.load_from_oracle_(unet, o_unet)
    unet.to(dev)
    if case == "lora":
        procs, holder = {}, torch.nn.ModuleList()
        boc = unet.config.block_out_channels
        for name in unet.attn_processors.keys():
            cad = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
            bid = int(name.split(".")[1]) if not name.startswith("mid") else 3
            hid = list(reversed(boc))[bid] if name.startswith("up_blocks") else boc[bid]
            p = M.LoRACrossAttnProcessor(hid, cad, rank=4)
            procs[name] = p
            holder.append(p)
        holder.load_state_dict(o_params.state_dict())
        holder.to(dev)
        unet.set_attn_processor(procs)
        return unet, holder, None
    clora = M.ControlLoRA(**cases.CASES[case])
    clora.load_state_dict(o_params.state_dict())
    clora.to(dev)
    unet.set_attn_processor(M.map_processors_to_unet(unet, clora))
    return unet, clora, clora


def run_product_step(case, dev):
    inp = {k: v.to(dev) for k, v in cases.seeded_inputs().items()}
    unet, params, clora = build_product_case(case, dev)
    sched = unet_ref.DDPMSchedule()
    noisy = sched.add_noise(inp["latents"].cpu(), inp["noise"].cpu(), inp["timesteps"].cpu()).to(dev).to(f16)
    out = {}

    class _NoCtrl(torch.nn.Module):
        def forward(self, x):
            return None

    trainer = ControlLoRATrainer(unet, params, init_scale=128.0, dynamic_scale=False)
    if clora is None:
        trainer.control_lora = _NoCtrl()
    pred = trainer.forward_backward(noisy, inp["timesteps"], inp["ehs"].to(f16), inp["guide"].to(f16), inp["noise"])
    if clora is not None:
        # control maps as injected into the processors
        ctrl = clora(inp["guide"].to(f16)).control_states
        for i, c in enumerate(ctrl):
            out[f"control_{i}"] = c.detach().float()
    out["pred"] = pred.detach().float()
    out["loss"] = torch.tensor([trainer.loss(pred.numel())])
    out["grads"] = torch.cat([p.grad.reshape(-1) for p in params.parameters()]).detach().float() / float(trainer.state[3])
    return out, trainer


def check_against_golden(case, dev, golden_dir, tol_pred=5e-3, tol_grad=2.5e-2):
    """Tolerances: fp16 activations through the ~100-layer small UNet vs the fp32 reference.  `pred`
    rel-L2 <= 5e-3, control maps <= 4e-3, flat adapter/hint gradient rel-L2 <= 2.5e-2, loss <= 2e-3 rel."""
    gold = load_file(os.path.join(golden_dir, f"case_{case}.safetensors"))
    out, _ = run_product_step(case, dev)
    errs = {}
    for k, v in out.items():
        errs[k] = rel(v, gold[k])
    for k, e in errs.items():
        lim = tol_grad if k == "grads" else (tol_pred if k == "pred" else (2e-3 if k == "loss" else 4e-3))
        assert e < lim, f"{case}:{k} rel-L2 {e:.3e} (limit {lim})  all={errs}"
    return errs
