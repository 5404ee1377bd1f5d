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


def check_trainer_features(dev, golden_dir, real_backward=True):
    """gradient accumulation (train...:174-178), LR-schedule multiplier and checkpoint/resume (train...:713-735)
    of the device-resident trainer: accumulated grads equal the single-batch golden grads, the optimizer only
    runs on the last micro-batch, and a trainer restored from `state_dict` continues identically
    (bit-exact with synthetic gradients; to fp32 round-off with real backward passes, whose wgrad uses atomics)."""
    inp = cases.seeded_inputs()
    noisy = unet_ref.DDPMSchedule().add_noise(inp["latents"], inp["noise"], inp["timesteps"]).to(dev).to(f16)
    args = (noisy, inp["timesteps"].to(dev), inp["ehs"].to(dev).to(f16), inp["guide"].to(dev).to(f16), inp["noise"].to(dev))
    sched = lambda step: 0.5 if step == 0 else 1.0

    def make(**kw):
        unet, params, _ = build_product_case("v1", dev)
        return ControlLoRATrainer(unet, params, init_scale=128.0, dynamic_scale=False, lr_lambda=sched, **kw)

    gen = torch.Generator().manual_seed(3)

    def backward(tr):
        if real_backward:
            tr.forward_backward(*args)
        else:                                    # synthetic scaled gradients: same values for every trainer / call
            if tr._micro == 0:
                tr.flat.zero_grad()
            g = torch.Generator().manual_seed(5)
            tr.flat.grad += (torch.randn(tr.flat.numel, generator=g) * 128.0 / tr.accum).to(dev)

    a = make(gradient_accumulation_steps=2)
    p0 = a.flat.data.clone()
    backward(a)
    assert a.optimizer_step() is False and torch.equal(a.flat.data, p0)          # first micro-batch: no update
    backward(a)
    if real_backward:
        gold = load_file(os.path.join(golden_dir, "case_v1.safetensors"))["grads"]
        assert rel(a.unscaled_grads_module_order(), gold) < 2.5e-2               # two half-weighted micro-batches
    assert a.optimizer_step() is True and not torch.equal(a.flat.data, p0)
    assert float(a.state[10]) == 0.5 and a.global_step == 1
    # Adam's first update has magnitude lr * multiplier (+ decoupled weight decay): 0.5e-4 here
    delta = (a.flat.data - p0 * (1 - 0.5e-4 * 1e-2)).abs().max()
    assert 0.45e-4 < float(delta) < 0.51e-4

    sd = a.state_dict()
    a.accum = 1
    backward(a)
    a.optimizer_step()
    b = make()
    b.load_state_dict(sd)
    assert b.global_step == 1
    backward(b)
    b.optimizer_step()
    assert float(b.state[10]) == 1.0
    if real_backward:      # the hint-encoder wgrad kernel combines its M-chunks with fp32 atomics: last-bit run-to-run noise
        assert rel(a.flat.data, b.flat.data) < 1e-5 and rel(a.flat.exp_avg_sq, b.flat.exp_avg_sq) < 1e-3
    else:
        assert torch.equal(a.flat.data, b.flat.data) and torch.equal(a.flat.exp_avg_sq, b.flat.exp_avg_sq)


def check_inference_broadcast(case, dev, tol=6e-3):
    """Inference call pattern (reference apps/gradio_canny2image.py:66-92): the hint encoder sees ONE guide image, the
    UNet a classifier-free-guidance batch of 2 -- the control states are repeated over the batch (quirk C6, reference
    models.py:209-213).  Product forward (no_grad; batched / cached control terms, broadcast second adapter input)
    against the oracle restatement, twice (the second call reuses the cached control states like a scheduler loop)."""
    inp = cases.seeded_inputs()
    o_unet, _, o_clora = cases.build_oracle_case(case)
    unet, _, clora = build_product_case(case, dev)
    guide = inp["guide"][:1]
    lat = torch.cat([inp["latents"][:1]] * 2)
    ehs = inp["ehs"][:2]
    errs = []
    with torch.no_grad():
        o_clora(guide)
        clora(guide.to(dev).to(f16))
        for t in (801, 401):
            ref = o_unet(lat, t, ehs).sample
            out = unet(lat.to(dev).to(f16), t, ehs.to(dev).to(f16)).sample
            errs.append(rel(out, ref))
    assert max(errs) < tol, errs
    return errs
