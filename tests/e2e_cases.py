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


def check_control_batch_repeat_interleave(case, dev, tol=6e-3):
    """1 < control batch < UNet batch on the concat-hidden processors (v2 / sketch): the reference repeat-interleaves the control
    batch (models.py:209-213, 343-347: guide 0 for samples 0-1, guide 1 for samples 2-3), the product materialises the same
    tensor -- forward vs the oracle restatement; a tiled order (g0,g1,g0,g1) must NOT match.  The plain path raises like the
    reference's broadcast does."""
    inp = cases.seeded_inputs(batch=4)
    o_unet, _, o_clora = cases.build_oracle_case(case)
    unet, _, clora = build_product_case(case, dev)
    guide = inp["guide"][:2]
    with torch.no_grad():
        o_clora(guide)
        clora(guide.to(dev).to(f16))
        ref = o_unet(inp["latents"], 501, inp["ehs"]).sample
        out = unet(inp["latents"].to(dev).to(f16), 501, inp["ehs"].to(dev).to(f16)).sample
        err = rel(out, ref)
        o_clora(guide[[0, 1, 0, 1]])                  # what tiling would compute: explicitly permuted guides, batch 4
        tiled = o_unet(inp["latents"], 501, inp["ehs"]).sample
    assert err < tol, err
    assert rel(tiled, ref) > 2 * err, ("the two orders must be distinguishable on this case", rel(tiled, ref), err)
    return err


def check_pre_post_chain(kind, dev, B=2, side=4, C=64, heads=4, ctx=48, ctrl_c=32, tol_y=4e-3, tol_dh=1e-2, tol_dc=2e-2, tol_w=3e-2):
    """pre_loras / post_loras chaining (reference models.py:232-243, 249-265, 276-282; mix_lora_and_control_lora.py:111-123):
    the product processors (unfused generic path) vs the oracle restatement -- site output, d(hidden), d(control) and
    every adapter weight gradient of the main / pre / post processors; self- and cross-attention sites."""
    from oracle import controllora_ref as cr
    torch.manual_seed(0)
    N = side * side
    worst = {}
    for self_attn in (True, False):
        cad = None if self_attn else ctx
        o_attn = unet_ref.CrossAttention(C, cad, heads=heads, dim_head=C // heads)
        cases.seeded_weights_(o_attn, seed=5)
        p_attn = U.CrossAttention(C, cad, heads=heads, dim_head=C // heads)
        with torch.no_grad():
            for k, v in p_attn.state_dict().items():
                v.copy_(o_attn.state_dict()[k].to(v.dtype))
        p_attn.to(dev)
        if kind == "v1":
            o_main, p_main = cr.ControlLoRAProcRef(C, cad, rank=4), M.ControlLoRACrossAttnProcessor(C, cad, rank=4)
        else:
            o_main = cr.ControlLoRAProcV2Ref(C, cad, rank=4, control_channels=ctrl_c)
            p_main = M.ControlLoRACrossAttnProcessorV2(C, cad, rank=4, control_channels=ctrl_c)
        o_pre, p_pre = cr.LoRAProcRef(C, cad, rank=4), M.LoRACrossAttnProcessor(C, cad, rank=4)
        o_post, p_post = cr.LoRAProcRef(C, cad, rank=8, post_add=True), M.LoRACrossAttnProcessor(C, cad, rank=8, post_add=True)
        for o, p_, sd in ((o_main, p_main, 1), (o_pre, p_pre, 2), (o_post, p_post, 3)):
            cases.seeded_weights_(o, seed=sd)
            p_.load_state_dict(o.state_dict())
            p_.to(dev)
        o_main.inject_pre_lora(o_pre); o_main.inject_post_lora(o_post)
        p_main.inject_pre_lora(p_pre); p_main.inject_post_lora(p_post)
        h = torch.randn(B, N, C).half()
        e = None if self_attn else torch.randn(B, 5, ctx).half()
        ctrl = torch.randn(B, C if kind == "v1" else ctrl_c, side, side).half()
        go = torch.randn(B, N, C).half()
        for q in o_attn.parameters():                      # oracle: fp32 math on fp16-rounded inputs / frozen weights
            q.data = q.data.half().float()
        ho = h.float().requires_grad_(True)
        co = ctrl.float().requires_grad_(True)
        o_main.inject_control_states(co)
        yo = o_main(o_attn, ho, None if e is None else e.float(), None, 0.7)
        yo.backward(go.float())
        hp = h.clone().to(dev).requires_grad_(True)
        cp = ctrl.permute(0, 2, 3, 1).reshape(B, N, -1).contiguous().to(dev).requires_grad_(True)
        p_main.inject_control_states(cp)
        yp = p_main(p_attn, hp, None if e is None else e.to(dev), None, 0.7)
        yp.backward(go.to(dev))
        errs = {"y": rel(yp, yo.detach()), "dh": rel(hp.grad, ho.grad),
                "dctrl": rel(cp.grad, co.grad.permute(0, 2, 3, 1).reshape(B, N, -1)), "dw": 0.0}
        assert errs["y"] < tol_y and errs["dh"] < tol_dh and errs["dctrl"] < tol_dc, errs
        for (n, a), (_, b_) in zip(list(p_main.named_parameters()) + list(p_pre.named_parameters()) + list(p_post.named_parameters()),
                                   list(o_main.named_parameters()) + list(o_pre.named_parameters()) + list(o_post.named_parameters())):
            if b_.grad is not None and float(b_.grad.norm()) > 0:
                e_ = rel(a.grad, b_.grad)
                errs["dw"] = max(errs["dw"], e_)
                assert e_ < tol_w, (n, e_)
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
    return worst


def check_site_real_shape(dev, hidden=320, side=32, B=2, rank=4, control_rank=256, control_channels=256, concat=True,
                          cross=False, scale=1.0, tol=4e-3, tol_g=1.2e-2):
    """ONE attention site at a REAL SD-1.5 shape with the danbooru-sketch adapter geometry (reference
    configs/danbooru-sketch.json: lora_control_rank 256, lora_concat_hidden, control channels 256; models.py:185-188,
    209-218): rank-256 `to_control` goes through the large-rank down / up / wgrad kernels (no batched path).
    Product processor on the HIP kernels vs the oracle processor: output, d(hidden), d(control), weight grads."""
    from oracle import controllora_ref as cr
    torch.manual_seed(0)
    N, heads = side * side, 8
    cad = 768 if cross else None
    o_attn = unet_ref.CrossAttention(hidden, cad, heads=heads, dim_head=hidden // heads)
    cases.seeded_weights_(o_attn, seed=5)
    p_attn = U.CrossAttention(hidden, cad, heads=heads, dim_head=hidden // heads)
    with torch.no_grad():
        for k, v in p_attn.state_dict().items():
            v.copy_(o_attn.state_dict()[k].to(v.dtype))
        for q in o_attn.parameters():
            q.data = q.data.half().float()
    p_attn.to(dev)
    kw = dict(rank=rank, control_rank=control_rank, concat_hidden=concat, control_channels=control_channels)
    o_p, p_p = cr.ControlLoRAProcRef(hidden, cad, **kw), M.ControlLoRACrossAttnProcessor(hidden, cad, **kw)
    cases.seeded_weights_(o_p, seed=9, up_std=0.02)
    p_p.load_state_dict(o_p.state_dict())
    p_p.to(dev)
    g = torch.Generator().manual_seed(2)
    h = torch.randn(B, N, hidden, generator=g).half()
    e = torch.randn(B, 77, 768, generator=g).half() if cross else None
    ctrl = torch.randn(B, control_channels, side, side, generator=g).half()
    go = (torch.randn(B, N, hidden, generator=g) * 0.1).half()
    ho, co = h.float().requires_grad_(True), ctrl.float().requires_grad_(True)
    o_p.inject_control_states(co)
    yo = o_p(o_attn, ho, None if e is None else e.float(), None, scale)
    yo.backward(go.float())
    hp = h.clone().to(dev).requires_grad_(True)
    cp = ctrl.permute(0, 2, 3, 1).reshape(B, N, -1).contiguous().to(dev).requires_grad_(True)
    p_p.inject_control_states(cp)
    yp = p_p(p_attn, hp, None if e is None else e.to(dev), None, scale)
    yp.backward(go.to(dev))
    errs = {"y": rel(yp, yo.detach()), "dh": rel(hp.grad, ho.grad),
            "dctrl": rel(cp.grad, co.grad.permute(0, 2, 3, 1).reshape(B, N, -1))}
    for (n, a), (_, b_) in zip(p_p.named_parameters(), o_p.named_parameters()):
        errs["dw:" + n] = rel(a.grad, b_.grad)
    assert errs["y"] < tol, errs
    bad = {k: v for k, v in errs.items() if k != "y" and v > tol_g}
    assert not bad, (bad, errs)
    return errs


def check_stock_attention_host(kind, dev, B=2, side=4, C=64, heads=4, ctx=48, ctrl_c=32, tol_y=4e-3, tol_g=1.2e-2):
    """The product processors installed on a module that has ONLY the stock diffusers `CrossAttention` surface (reference
    models.py:122-150: to_q / to_k / to_v / to_out / heads / scale; here oracle/unet_ref.CrossAttention with fp16 weights), called the way
    diffusers calls them (`attn(hidden, encoder_hidden_states=..., scale=...)` -> `processor(attn, ...)`): models.StockAttentionHost packs
    the frozen weights and the site runs on the same kernels -- output and every gradient (a) equal to the same processor on this
    repository's own unet.CrossAttention, bit for bit, and (b) within the site tolerances of the oracle processor on the fp32 module."""
    import copy
    from oracle import controllora_ref as cr
    torch.manual_seed(0)
    N = side * side
    worst = {}
    for self_attn in (True, False):
        cad = None if self_attn else ctx
        o_attn = unet_ref.CrossAttention(C, cad, heads=heads, dim_head=C // heads)
        cases.seeded_weights_(o_attn, seed=5)
        own = U.CrossAttention(C, cad, heads=heads, dim_head=C // heads)
        with torch.no_grad():
            for k, v in own.state_dict().items():
                v.copy_(o_attn.state_dict()[k].to(v.dtype))
        own.to(dev)
        stock = copy.deepcopy(o_attn).half().to(dev).requires_grad_(False)          # what `unet.to(device, dtype=fp16)` leaves (train...:493)
        assert not hasattr(stock, "fused_packs")
        for q in o_attn.parameters():
            q.data = q.data.half().float()
        if kind == "v1":
            mk_o, mk_p = (lambda: cr.ControlLoRAProcRef(C, cad, rank=4)), (lambda: M.ControlLoRACrossAttnProcessor(C, cad, rank=4))
        elif kind == "v2":
            mk_o = lambda: cr.ControlLoRAProcV2Ref(C, cad, rank=4, control_channels=ctrl_c)
            mk_p = lambda: M.ControlLoRACrossAttnProcessorV2(C, cad, rank=4, control_channels=ctrl_c)
        else:
            mk_o, mk_p = (lambda: cr.LoRAProcRef(C, cad, rank=4)), (lambda: M.LoRACrossAttnProcessor(C, cad, rank=4))
        o_p = mk_o()
        cases.seeded_weights_(o_p, seed=3, up_std=0.05)
        h = torch.randn(B, N, C).half()
        e = None if self_attn else torch.randn(B, 5, ctx).half()
        ctrl = torch.randn(B, C if kind == "v1" else ctrl_c, side, side).half()
        go = torch.randn(B, N, C).half()
        ho, co = h.float().requires_grad_(True), ctrl.float().requires_grad_(True)
        if kind != "lora":
            o_p.inject_control_states(co)
        o_attn.set_processor(o_p)
        yo = o_attn(ho, encoder_hidden_states=None if e is None else e.float(), scale=0.7)
        yo.backward(go.float())

        def run(attn_mod):
            p = mk_p()
            p.load_state_dict(o_p.state_dict())
            p.to(dev)
            hp = h.clone().to(dev).requires_grad_(True)
            cp = ctrl.permute(0, 2, 3, 1).reshape(B, N, -1).contiguous().to(dev).requires_grad_(True)
            if kind != "lora":
                p.inject_control_states(cp)
            attn_mod.set_processor(p)
            y = attn_mod(hp, encoder_hidden_states=None if e is None else e.to(dev), scale=0.7)
            y.backward(go.to(dev))
            return y.detach(), hp.grad, (cp.grad if kind != "lora" else None), [q.grad for q in p.parameters()]

        ys, dhs, dcs, dws = run(stock)
        yw, dhw, dcw, dww = run(own)
        assert torch.equal(ys, yw) and torch.equal(dhs, dhw), "stock-module host differs from unet.CrossAttention"
        assert dcs is None or torch.equal(dcs, dcw)
        assert all((a is None and b is None) or torch.equal(a, b) for a, b in zip(dws, dww))
        errs = {"y": rel(ys, yo.detach()), "dh": rel(dhs, ho.grad)}
        if kind != "lora":
            errs["dctrl"] = rel(dcs, co.grad.permute(0, 2, 3, 1).reshape(B, N, -1))
        for (n, b_), a in zip(o_p.named_parameters(), dws):
            if b_.grad is not None and float(b_.grad.norm()) > 0:
                errs["dw:" + n] = rel(a, b_.grad)
        assert errs["y"] < tol_y, errs
        bad = {k: v for k, v in errs.items() if k != "y" and v > (2.5e-2 if k.startswith("dw") else tol_g)}
        assert not bad, (bad, errs)
        assert "_clora_host" in stock.__dict__ and "_clora_host" not in dict(stock.named_modules())
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
    return worst
