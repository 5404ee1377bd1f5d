"""Parity on BASELINE.json's OWN configurations: the full SD-1.5 topology at real widths
(320/640/1280 channels, head dims 40/80/160, 2560-channel concat resnets, the tuned GEMM table) -- product path on
the GPU against the CPU oracle (fp32 restatement of reference train_text_to_image_control_lora.py:751-796 and
apps/gradio_canny2image.py:66-92).  BASELINE configs[0] geometry: 256x256, batch 1.

The oracle's frozen UNet weights and all inputs are fp16-rounded values held in fp32 (SURVEY.md section 8c "Tolerance
reading"): the two sides then differ by accumulation order and by the product's fp16 activation storage only.
Everything is seeded, nothing is read from /root/reference (absent on the GPU box)."""
import os

import torch

from controllora_amd import models as M
from controllora_amd import unet as U
from controllora_amd.train import ControlLoRATrainer
from oracle import cases, unet_ref
from oracle.controllora_ref import ControlLoRARef, map_processors_to_unet, randomize_adapters_

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f16 = torch.float16


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


_ORACLE_UNET = {}


def oracle_unet_sd15():
    """SD-1.5-shaped oracle UNet with seeded weights, rounded to fp16 values (859.5 M parameters, built once per
    process: ~15 s)."""
    if "u" not in _ORACLE_UNET:
        u = unet_ref.UNet2DConditionModel()
        unet_ref.init_unet_weights_(u, seed=0)
        with torch.no_grad():
            for p in u.parameters():
                p.copy_(p.half().float())
                p.requires_grad_(False)
        _ORACLE_UNET["u"] = u
    return _ORACLE_UNET["u"]


_PRODUCT_UNET = {}


def product_unet_sd15(dev):
    if "u" not in _PRODUCT_UNET:
        u = U.UNet2DConditionModel()
        U.load_from_oracle_(u, oracle_unet_sd15())
        _PRODUCT_UNET["u"] = u.to(dev)
    return _PRODUCT_UNET["u"]


def build_pair(config_name, dev, up_std=0.02):
    """(oracle unet, oracle clora, product unet, product clora) for configs/<config_name>, identical weights."""
    o_unet = oracle_unet_sd15()
    torch.manual_seed(1)
    o_clora = ControlLoRARef.from_config(os.path.join(ROOT, "configs", config_name))
    randomize_adapters_(o_clora, seed=1, std=up_std)          # non-zero `up`: zero-init would hide adapter bugs
    o_unet.set_attn_processor(map_processors_to_unet(o_unet, o_clora))
    p_unet = product_unet_sd15(dev)
    p_clora = M.ControlLoRA.from_config(os.path.join(ROOT, "configs", config_name))
    p_clora.load_state_dict(o_clora.state_dict())             # strict: same key set as the reference-keyed oracle
    p_clora.to(dev)
    p_unet.set_attn_processor(M.map_processors_to_unet(p_unet, p_clora))
    return o_unet, o_clora, p_unet, p_clora


def inputs(res=256, batch=1, seed=42):
    g = torch.Generator().manual_seed(seed)
    L = res // 8
    guide = (torch.rand(batch, 3, res, res, generator=g) > 0.9).float() * 2 - 1       # sparse +-1 edge-map-like guide
    return dict(guide=guide.half().float(),
                latents=torch.randn(batch, 4, L, L, generator=g).half().float(),
                noise=torch.randn(batch, 4, L, L, generator=g).half().float(),
                timesteps=torch.randint(0, 1000, (batch,), generator=g),
                ehs=torch.randn(batch, 77, 768, generator=g).half().float())


def product_add_noise(inp, dev):
    """The PRODUCT's DDPM `add_noise` (controllora_amd/schedulers.py; reference train...:765) on the device, the way the
    entry point calls it (fp16 latents / noise), asserted against the oracle's: the fp32 oracle result rounded
    to fp16 may differ from the fp16 evaluation by one rounding of each product and of the sum."""
    from controllora_amd.schedulers import DDPMScheduler
    ts = inp["timesteps"].to(dev)
    noisy = DDPMScheduler().add_noise(inp["latents"].to(dev).to(f16), inp["noise"].to(dev).to(f16), ts).to(f16)
    ref = unet_ref.DDPMSchedule().add_noise(inp["latents"], inp["noise"], inp["timesteps"])
    e = rel(noisy, ref)
    assert e < 6e-4, f"product add_noise vs oracle: rel-L2 {e:.2e}"
    return noisy


def train_step_parity(config_name, dev, res=256, batch=1):
    """One reference train step (train...:757-790) on the SD-1.5 topology: returns rel-L2 of the control maps, the UNet
    prediction, the loss and the flat gradient of every trainable parameter (adapters + hint encoder)."""
    o_unet, o_clora, p_unet, p_clora = build_pair(config_name, dev)
    inp = inputs(res, batch)
    gold = cases.oracle_train_step(o_unet, o_clora, o_clora, inp)
    noisy = product_add_noise(inp, dev)
    trainer = ControlLoRATrainer(p_unet, p_clora, init_scale=1024.0, dynamic_scale=False)
    pred = trainer.forward_backward(noisy, inp["timesteps"].to(dev), inp["ehs"].to(dev).to(f16),
                                    inp["guide"].to(dev).to(f16), inp["noise"].to(dev))
    out = {"pred": pred, "loss": torch.tensor([trainer.loss(pred.numel())]),
           "grads": trainer.unscaled_grads_module_order()}
    for i, c in enumerate(p_clora(inp["guide"].to(dev).to(f16)).control_states):
        out[f"control_{i}"] = c
    errs = {k: rel(v, gold[k]) for k, v in out.items()}
    errs["n_trainable"] = trainer.flat.numel
    errs["loss_value"] = float(gold["loss"])
    # per-group gradient parity: hint encoder vs adapters (a broken adapter path must not hide behind the larger group)
    names = [n for n, p in p_clora.named_parameters() if p.requires_grad]
    sizes = [p.numel() for n, p in p_clora.named_parameters() if p.requires_grad]
    gp, go = out["grads"].float().cpu(), gold["grads"].float()
    off, acc = 0, {"adapters": [[], []], "hint": [[], []]}
    for n, k in zip(names, sizes):
        grp = "adapters" if n.startswith("lora_layers") else "hint"
        acc[grp][0].append(gp[off:off + k]); acc[grp][1].append(go[off:off + k])
        off += k
    for grp, (a, b) in acc.items():
        errs[f"grads_{grp}"] = rel(torch.cat(a), torch.cat(b))
    return errs


@torch.no_grad()
def oracle_ddim(o_unet, o_clora, guide, cond, uncond, steps, guidance_scale, latents):
    """CPU restatement of the inference call pattern (apps/gradio_canny2image.py:83-89: hint-encode ONE guide image,
    the pipeline's scheduler loop with classifier-free guidance, UNet batch 2x, uncond first; DDIM eta=0 = BASELINE
    inference config)."""
    sch = unet_ref.DDPMSchedule()
    o_clora(guide)
    ehs = torch.cat([uncond, cond], 0)
    x = latents.clone()
    traj = []
    for t in sch.ddim_timesteps(steps):
        eps = o_unet(torch.cat([x, x], 0), t, ehs).sample
        eu, ec = eps.chunk(2)
        eps = eu + guidance_scale * (ec - eu)
        x = sch.ddim_step(eps, t, x, steps)
        traj.append(x.clone())
    return x, traj


@torch.no_grad()
def oracle_dpm(o_unet, o_clora, guide, cond, uncond, steps, guidance_scale, latents):
    """The reference's validation-time / app sampling loop (train_text_to_image_control_lora.py:811-843, apps/gradio_*2image.py:
    `DPMSolverMultistepScheduler.from_config`, 30 steps, CFG) restated independently of controllora_amd/schedulers.py: DPM-Solver++(2M)
    in the PAPER's form (Lu et al. 2022, Algorithm 2): with lambda = log(alpha / sigma), h = lambda_t - lambda_s, r = h_prev / h,
        x_t = (sigma_t / sigma_s) x_s - alpha_t (e^{-h} - 1) D,   D = (1 + 1/(2r)) x0_s - (1/(2r)) x0_prev   (first step: D = x0_s),
    SD's scaled-linear betas, upstream's timestep grid (linspace(0, 999, n + 1) rounded, reversed, last dropped), the last step goes
    to t = 0; `lower_order_final` (first order on the last step) only below 15 steps.  fp64 coefficients, fp32 state."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
    acp = torch.cumprod(1.0 - betas, 0)
    alpha, sigma = acp.sqrt(), (1.0 - acp).sqrt()
    lam = torch.log(alpha) - torch.log(sigma)
    ts = [int(t) for t in torch.linspace(0, 999, steps + 1).round().long().flip(0)[:-1]]
    o_clora(guide)
    ehs = torch.cat([uncond, cond], 0)
    x = latents.clone()
    x0_prev, lam_prev = None, None
    for i, t in enumerate(ts):
        eps = o_unet(torch.cat([x, x], 0), t, ehs).sample
        eu, ec = eps.chunk(2)
        eps = eu + guidance_scale * (ec - eu)
        t_next = ts[i + 1] if i + 1 < len(ts) else 0
        x0 = (x - float(sigma[t]) * eps) / float(alpha[t])
        h = float(lam[t_next] - lam[t])
        first = x0_prev is None or (steps < 15 and i == len(ts) - 1)
        if first:
            D = x0
        else:
            r = float(lam[t] - lam_prev) / h
            D = (1.0 + 0.5 / r) * x0 - (0.5 / r) * x0_prev
        import math
        x = float(sigma[t_next] / sigma[t]) * x - float(alpha[t_next]) * math.expm1(-h) * D
        x0_prev, lam_prev = x0, lam[t]
    return x, None


def ddim_parity(o_unet, o_clora, p_unet, p_clora, dev, res, steps, guidance_scale=9.0, nb=1, ctx_dim=768, ctx_len=77, seed=5,
                graph=False, fp16_floor=False, sampler="ddim"):
    """denoised-latent parity: product `pipeline.ddim_sample` vs the oracle loop; rel-L2 of the final latents (the
    quantity north_star states 1e-3 for) and of the per-step trajectory."""
    from controllora_amd.pipeline import ddim_sample
    g = torch.Generator().manual_seed(seed)
    L = res // 8
    guide = ((torch.rand(1, 3, res, res, generator=g) > 0.9).float() * 2 - 1)
    cond = torch.randn(nb, ctx_len, ctx_dim, generator=g).half().float()
    uncond = torch.randn(nb, ctx_len, ctx_dim, generator=g).half().float()
    lat0 = torch.randn(nb, 4, L, L, generator=g).half().float()
    loop = oracle_ddim if sampler == "ddim" else oracle_dpm
    # the oracle trajectory (and its fp16-regime floor) depend only on the case, not on how the product is driven (eager / graph): one
    # oracle run per case and process (round 6: the two DPM parametrisations were 2 x 70 s of the same CPU loop on the GPU box)
    key = (sampler, res, steps, float(guidance_scale), nb, ctx_dim, ctx_len, seed, bool(fp16_floor),
           tuple(sorted(k for k, _ in o_clora.named_parameters()))[:4], float(sum(float(p.double().sum()) for p in o_clora.parameters())))
    hit = _ORACLE_LOOPS.get(key)
    if hit is None:
        ref, traj = loop(o_unet, o_clora, guide, cond, uncond, steps, guidance_scale, lat0)
        floor = None
        if fp16_floor:
            floor = fp16_oracle_floor(o_unet, o_clora, guide, cond, uncond, steps, guidance_scale, lat0, ref, loop=loop, dev=dev)
        hit = _ORACLE_LOOPS[key] = (ref, floor)
    ref, floor = hit
    kw = dict(graph=True) if graph else {}
    out = ddim_sample(p_unet, p_clora, guide.to(dev).half(), cond.to(dev).half(), uncond.to(dev).half(), steps=steps,
                      guidance_scale=guidance_scale, latents=lat0.to(dev).half(), sampler=sampler, **kw)
    return {"latents": rel(out, ref), "steps": steps, "latent_norm": float(ref.norm()), "fp16_oracle_vs_fp32_oracle": floor}


_ORACLE_LOOPS = {}


def fp16_oracle_floor(o_unet, o_clora, guide, cond, uncond, steps, guidance_scale, lat0, ref, loop=None, dev="cpu"):
    """What "fp16" costs ANY implementation: the same oracle loop with every module and tensor in fp16 (the arithmetic
    regime of the reference's own fp16 pipeline: fp16 storage of weights / activations / latents) against the fp32 oracle.
    north_star's "within 1e-3 rel fp16" is read against this floor (SURVEY.md section 8c "Tolerance reading").
    On a GPU box the fp16 oracle runs there (stock torch ops, the reference's own regime: fp16 storage, fp32 accumulation inside
    the vendor kernels) -- the host's fp16 kernels made this loop 100 s of a 150 s test."""
    import copy
    from oracle.controllora_ref import map_processors_to_unet as omap
    def run(where):
        h_unet, h_clora = copy.deepcopy(o_unet).half().to(where), copy.deepcopy(o_clora).half().to(where)
        h_unet.set_attn_processor(omap(h_unet, h_clora))
        mv = lambda t: t.half().to(where)
        out, _ = (loop or oracle_ddim)(h_unet, h_clora, mv(guide), mv(cond), mv(uncond), steps, guidance_scale, mv(lat0))
        return rel(out.float().cpu(), ref)
    if torch.device(dev).type == "cuda":
        try:
            return run(dev)
        except RuntimeError as e:                      # an oracle module that builds a tensor on the host: the host loop instead
            print("NOTE fp16_oracle_floor: GPU run of the oracle failed, falling back to the host:", str(e).splitlines()[0][:200])
    return run("cpu")


def full_size_properties(dev, config_name="fill50k.json", res=512, batch=4):
    """BASELINE configs[1] at its FULL size (SD-1.5 topology, 512x512, batch 4: the shapes of the tuned launch table, the
    patch-staged convs, the 8-wave attention blocks) through properties that do not need an oracle run of that size:
      identity   : fresh adapters (zero `up`, reference models.py:45 init) leave the UNet prediction bit-identical to the plain UNet;
      batch      : sample i of the batch-4 prediction equals the batch-1 prediction of sample i (other launch shapes / tiles / split-K);
      additivity : the flat gradient of the batch-4 MSE step equals the mean of the four batch-1 gradients (loss is a mean);
      scaling    : doubling the loss scale doubles the raw gradient (the backward is linear in its seed)."""
    _, o_clora, p_unet, p_clora = build_pair(config_name, dev)
    inp = inputs(res, batch, seed=7)
    g16 = lambda k: inp[k].to(dev).to(f16)
    noisy = unet_ref.DDPMSchedule().add_noise(inp["latents"], inp["noise"], inp["timesteps"]).to(dev).to(f16)
    ts = inp["timesteps"].to(dev)
    out = {}
    with torch.no_grad():
        # identity: zero the up matrices of a copy of the adapters
        fresh = M.ControlLoRA.from_config(os.path.join(ROOT, "configs", config_name)).to(dev)
        p_unet.set_attn_processor(M.map_processors_to_unet(p_unet, fresh))
        fresh(g16("guide"))
        with_adapters = p_unet(noisy, ts, g16("ehs")).sample.clone()
        p_unet.set_attn_processor(U.CrossAttnProcessor())
        plain = p_unet(noisy, ts, g16("ehs")).sample.clone()
        out["identity_bit_exact"] = bool(torch.equal(with_adapters, plain))
        # batch independence with the trained-like (non-zero up) adapters
        p_unet.set_attn_processor(M.map_processors_to_unet(p_unet, p_clora))
        p_clora(g16("guide"))
        pred4 = p_unet(noisy, ts, g16("ehs")).sample.clone()
        # run-to-run: the same forward twice must give the same bits (every forward kernel is atomics-free; a sporadic
        # wrong element in one launch -- seen in round 3 with an epilogue variant -- shows up here at the real shapes)
        out["forward_bit_stable"] = bool(torch.equal(pred4, p_unet(noisy, ts, g16("ehs")).sample)) and \
            bool(torch.equal(pred4, p_unet(noisy, ts, g16("ehs")).sample))
        errs = []
        for i in range(batch):
            p_clora(g16("guide")[i:i + 1])
            errs.append(rel(p_unet(noisy[i:i + 1], ts[i:i + 1], g16("ehs")[i:i + 1]).sample, pred4[i:i + 1]))
        out["batch_vs_single_pred"] = max(errs)
    # gradients
    def grads(sl, scale):
        tr = ControlLoRATrainer(p_unet, p_clora, init_scale=scale, dynamic_scale=False)
        tr.flat.zero_grad()
        tr.forward_backward(noisy[sl], ts[sl], g16("ehs")[sl], g16("guide")[sl], inp["noise"].to(dev)[sl])
        return tr.unscaled_grads_module_order().clone(), tr.flat.grad.clone()
    g4, raw1 = grads(slice(0, batch), 1024.0)
    _, raw2 = grads(slice(0, batch), 2048.0)
    out["seed_linearity"] = rel(raw2, 2.0 * raw1)
    gsum = None
    for i in range(batch):
        gi, _ = grads(slice(i, i + 1), 1024.0)
        gsum = gi if gsum is None else gsum + gi
    out["grad_additivity"] = rel(g4, gsum / batch)
    out["grad_norm"] = float(g4.norm())
    return out


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE's own sizes against COMMITTED oracle fixtures (oracle/make_fullsize_golden.py, written in the build container)
def _checksum(t):
    t = t.detach().double().reshape(-1).cpu()
    return torch.stack([t.sum(), t.abs().sum()])


def _assert_same_inputs(fx, name, t):
    got, want = _checksum(t), fx[name].double()
    assert torch.allclose(got, want, rtol=1e-9, atol=1e-6), f"{name}: seeded input differs from the fixture's ({got} vs {want})"


def load_fixture(name):
    from safetensors import safe_open
    path = os.path.join(ROOT, "tests", "golden", name)
    with safe_open(path, "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}, f.metadata()


def train_step_vs_fixture(dev, fixture="full_train_512_bs4.safetensors", regime_floor=False):
    """BASELINE configs[1] (configs/fill50k.json, SD-1.5 topology, 512x512, batch 4; reference train...:751-796) -- or, with
    fixture="full_train_512_bs8_v2.safetensors", BASELINE configs[3] as quoted (configs/mpii-pose-v2.json, batch 8; reference
    models.py:292-431): the product train step on the GPU against the committed oracle record -- prediction, loss, the four
    control maps (two strided samples with coprime strides + norm), the flat gradient of every trainable parameter (two strided
    samples, norm, per-parameter norms)."""
    from oracle import cases as ocases
    fx, meta = load_fixture(fixture)
    sg, sc = int(meta["stride_grad"]), int(meta["stride_ctrl"])
    o_unet, o_clora, p_unet, p_clora = build_pair(meta["config"], dev)
    assert torch.allclose(ocases.weight_checksum(o_unet), fx["weights_checksum_unet"].double(), rtol=1e-9), "seeded UNet weights differ"
    assert torch.allclose(ocases.weight_checksum(o_clora), fx["weights_checksum_clora"].double(), rtol=1e-9), "seeded adapters differ"
    inp = inputs(int(meta["res"]), int(meta["batch"]), seed=int(meta["input_seed"]))
    for k in ("guide", "latents", "noise", "ehs"):
        _assert_same_inputs(fx, f"in_{k}_checksum", inp[k])
    assert torch.equal(inp["timesteps"], fx["in_timesteps"])
    noisy = product_add_noise(inp, dev)
    trainer = ControlLoRATrainer(p_unet, p_clora, init_scale=1024.0, dynamic_scale=False)
    pred = trainer.forward_backward(noisy, inp["timesteps"].to(dev), inp["ehs"].to(dev).to(f16),
                                    inp["guide"].to(dev).to(f16), inp["noise"].to(dev))
    grads = trainer.unscaled_grads_module_order().float().cpu()
    errs = {"pred": rel(pred, fx["pred"]),
            "loss": abs(trainer.loss(pred.numel()) - float(fx["loss"])) / float(fx["loss"]),
            "grads_sample": rel(grads[::sg], fx["grads_sample"]),
            "grads_norm": abs(float(grads.double().norm()) - float(fx["grads_norm"])) / float(fx["grads_norm"])}
    if "grads_sample2" in fx:                           # second sample, coprime stride (round 4)
        errs["grads_sample2"] = rel(grads[::int(meta["stride_grad2"])], fx["grads_sample2"])
    for i, c in enumerate(p_clora(inp["guide"].to(dev).to(f16)).control_states):
        c = c.float().cpu()                              # NCHW view: same element order as the oracle's maps
        errs[f"control_{i}"] = rel(c.reshape(-1)[::sc], fx[f"control_{i}_sample"])
        if f"control_{i}_sample2" in fx:
            errs[f"control_{i}_s2"] = rel(c.reshape(-1)[::int(meta["stride_ctrl2"])], fx[f"control_{i}_sample2"])
        errs[f"control_{i}_norm"] = abs(float(c.double().norm()) - float(fx[f"control_{i}_norm"])) / float(fx[f"control_{i}_norm"])
    # per-parameter gradient norms: an error confined to one small tensor cannot hide inside the global rel-L2
    names = meta["param_names"].split("\n")
    sizes = [p.numel() for _, p in p_clora.named_parameters()]
    assert names == [n for n, _ in p_clora.named_parameters()] and sum(sizes) == grads.numel()
    want = fx["grads_param_norms"].double()
    got = torch.stack([c.double().norm() for c in torch.split(grads, sizes)])
    big = want > 1e-3 * want.max()
    errs["param_norm_worst"] = float(((got - want).abs() / want)[big].max())
    relerr = ((got - want).abs() / want) * big
    errs["param_norm_worst_name"] = names[int(relerr.argmax())]
    errs["param_norm_small_abs_worst"] = float(((got - want).abs()[~big]).max() / want.max()) if bool((~big).any()) else 0.0
    errs["n_params"] = len(names)
    errs["oracle_seconds"] = float(fx["oracle_seconds"])
    errs["clora_impl"] = meta.get("clora_impl", "restatement")
    if regime_floor:
        errs["fp16_regime"] = fp16_regime_train_step_errs(o_unet, o_clora, inp, fx, meta, sizes, dev)
    return errs


def fp16_regime_train_step_errs(o_unet, o_clora, inp, fx, meta, sizes, dev, loss_scale=1024.0):
    """The ORACLE's train step in the reference's fp16 arithmetic (oracle/precision_regimes.py "fp16": stock torch ops, every
    weight / activation / gradient in fp16, loss scaled like the product's trainer), run on the GPU, against the same committed fp32
    record: the distance any fp16 implementation of this step sits from the fixture -- for the quantities that have no other
    measured regime figure (gradient samples / norms, per-parameter norms, control maps)."""
    from oracle import precision_regimes as PR
    from oracle.unet_ref import DDPMSchedule
    import torch.nn.functional as Fn
    u16, c16 = PR.build_regime(o_unet, o_clora, "fp16", dev)
    for p in u16.parameters():
        p.requires_grad_(False)
    for p in c16.parameters():
        p.requires_grad_(True)
        p.grad = None
    ctrl = c16(inp["guide"].to(dev).half()).control_states
    noisy = DDPMSchedule().add_noise(inp["latents"], inp["noise"], inp["timesteps"]).to(dev).half()
    pred = u16(noisy, inp["timesteps"].to(dev), inp["ehs"].to(dev).half()).sample
    loss = Fn.mse_loss(pred.float(), inp["noise"].to(dev).float(), reduction="mean")
    (loss * loss_scale).backward()
    grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in c16.parameters()]).cpu() / loss_scale
    assert grads.numel() == sum(sizes)
    sg, sc = int(meta["stride_grad"]), int(meta["stride_ctrl"])
    r = {"pred": rel(pred, fx["pred"]), "grads_sample": rel(grads[::sg], fx["grads_sample"]),
         "grads_norm": abs(float(grads.double().norm()) - float(fx["grads_norm"])) / float(fx["grads_norm"])}
    if "grads_sample2" in fx:
        r["grads_sample2"] = rel(grads[::int(meta["stride_grad2"])], fx["grads_sample2"])
    for i, c in enumerate(ctrl):
        c = c.detach().float().cpu()
        r[f"control_{i}"] = rel(c.reshape(-1)[::sc], fx[f"control_{i}_sample"])
        r[f"control_{i}_norm"] = abs(float(c.double().norm()) - float(fx[f"control_{i}_norm"])) / float(fx[f"control_{i}_norm"])
    want = fx["grads_param_norms"].double()
    got = torch.stack([c.double().norm() for c in torch.split(grads, sizes)])
    big = want > 1e-3 * want.max()
    r["param_norm_worst"] = float(((got - want).abs() / want)[big].max())
    r["finite"] = bool(torch.isfinite(grads).all())
    del u16, c16
    torch.cuda.empty_cache()
    return r


class _StopSampling(Exception):
    pass


def infer32_vs_fixture(dev):
    """BASELINE config 5 at its OWN batch (reference apps/gradio_canny2image.py:83-89: 16 images => UNet batch 32, one guide
    broadcast over the batch, 50-step DDIM schedule, CFG 9.0, 512x512): the product's first UNet evaluation (all 32 samples) and
    the latents after scheduler steps 1 and 5, replayed from its hipGraph as shipped, vs tests/golden/full_infer_512_b32.safetensors.
    These are the M = 131072 / 32768 / 8192 / 2048 launch-table entries end to end."""
    from controllora_amd.pipeline import ddim_sample
    from oracle import cases as ocases
    from oracle.make_fullsize_golden import infer32_inputs
    fx, meta = load_fixture("full_infer_512_b32.safetensors")
    o_unet, o_clora, p_unet, p_clora = build_pair(meta["config"], dev)
    assert torch.allclose(ocases.weight_checksum(o_unet), fx["weights_checksum_unet"].double(), rtol=1e-9)
    assert torch.allclose(ocases.weight_checksum(o_clora), fx["weights_checksum_clora"].double(), rtol=1e-9)
    guide, cond, uncond, lat0 = infer32_inputs(int(meta["res"]), int(meta["images"]), int(meta["input_seed"]))
    for k, v in (("guide", guide), ("cond", cond), ("uncond", uncond), ("lat0", lat0)):
        _assert_same_inputs(fx, f"in_{k}_checksum", v)
    keep = [int(k) for k in meta["keep"].split(",")]
    traj = {}

    def cb(i, x, eps):
        traj[i] = (x.float().cpu().clone(), eps.float().cpu().clone() if i == 1 else None)
        if i >= max(keep):
            raise _StopSampling()

    try:
        ddim_sample(p_unet, p_clora, guide.to(dev).half(), cond.to(dev).half(), uncond.to(dev).half(), steps=int(meta["steps"]),
                    guidance_scale=float(meta["guidance_scale"]), latents=lat0.to(dev).half(), graph=True, callback=cb)
    except _StopSampling:
        pass
    errs = {"eps_step01": rel(traj[1][1], fx["eps_step01"]), "unet_batch": int(traj[1][1].shape[0]),
            "oracle_seconds": float(fx["oracle_seconds"])}
    per = (traj[1][1] - fx["eps_step01"]).flatten(1).norm(dim=1) / fx["eps_step01"].flatten(1).norm(dim=1)
    errs["eps_step01_worst_sample"] = float(per.max())                   # no single sample of the 32 may hide in the batch norm
    for i in keep:
        errs[f"latents_step{i:02d}"] = rel(traj[i][0], fx[f"latents_step{i:02d}"])
    return errs


def vae_512_vs_fixture(dev):
    """SD-1.5 VAE topology at 512x512, batch 1 (reference train...:753-754 encode, apps/gradio_canny2image.py:88-92 decode): the
    W = 512 / 256 implicit-GEMM conv path with M = 262144 rows, the 4096-token mid-block attention -- product vs the committed
    outputs of oracle/vae_ref.py."""
    from controllora_amd import vae as V
    from oracle import cases as ocases
    from oracle.make_fullsize_golden import vae_inputs, vae_oracle
    fx, meta = load_fixture("full_vae_512.safetensors")
    o = vae_oracle(int(meta["seed"]))
    x, eps = vae_inputs(int(meta["res"]), int(meta["batch"]))
    assert torch.allclose(ocases.weight_checksum(o), fx["weights_checksum"].double(), rtol=1e-9), "seeded VAE weights differ"
    _assert_same_inputs(fx, "in_x_checksum", x)
    _assert_same_inputs(fx, "in_eps_checksum", eps)
    m = V.AutoencoderKL(**V.SD15_VAE)
    V.load_from_oracle_(m, o)
    m.to(dev)
    with torch.no_grad():
        dist = m.encode(x.to(dev).half()).latent_dist
        errs = {"mean": rel(dist.mean, fx["mean"]), "logvar": rel(dist.logvar, fx["logvar"]),
                "sample": rel(dist.sample(noise=eps.to(dev)), fx["z"])}
        img = m.decode(fx["z"].to(dev).half()).sample.float().cpu().reshape(-1)
    errs["decode"] = rel(img[::int(meta["stride"])], fx["img_sample"])
    errs["decode_s2"] = rel(img[::int(meta["stride2"])], fx["img_sample2"])
    errs["decode_norm"] = abs(float(img.double().norm()) - float(fx["img_norm"])) / float(fx["img_norm"])
    return errs


def ddim_vs_fixture(dev, graph=True):
    """BASELINE inference geometry (reference apps/gradio_canny2image.py:83-89): 50 DDIM steps, CFG 9.0, 512x512, UNet batch 4
    (2 images), product `pipeline.ddim_sample` vs tests/golden/full_ddim_512_50.safetensors: the first UNet evaluation, the
    trajectory at 8 steps and the final denoised latents (the quantity north_star states 1e-3 for)."""
    from controllora_amd.pipeline import ddim_sample
    from oracle import cases as ocases
    from oracle.make_fullsize_golden import ddim_inputs
    fx, meta = load_fixture("full_ddim_512_50.safetensors")
    o_unet, o_clora, p_unet, p_clora = build_pair(meta["config"], dev)
    assert torch.allclose(ocases.weight_checksum(o_unet), fx["weights_checksum_unet"].double(), rtol=1e-9)
    assert torch.allclose(ocases.weight_checksum(o_clora), fx["weights_checksum_clora"].double(), rtol=1e-9)
    guide, cond, uncond, lat0 = ddim_inputs(int(meta["res"]), int(meta["images"]), int(meta["input_seed"]))
    for k, v in (("guide", guide), ("cond", cond), ("uncond", uncond), ("lat0", lat0)):
        _assert_same_inputs(fx, f"in_{k}_checksum", v)
    traj = {}
    out = ddim_sample(p_unet, p_clora, guide.to(dev).half(), cond.to(dev).half(), uncond.to(dev).half(),
                      steps=int(meta["steps"]), guidance_scale=float(meta["guidance_scale"]), latents=lat0.to(dev).half(),
                      graph=graph, callback=lambda i, x, eps: traj.__setitem__(i, (x.float().cpu().clone(), eps.float().cpu().clone())))
    errs = {"latents": rel(out, fx["latents"]), "eps_step01": rel(traj[1][1], fx["eps_step01"]),
            "latent_norm": float(fx["latents"].norm()), "oracle_seconds": float(fx["oracle_seconds"])}
    for k in sorted(fx):
        if k.startswith("latents_step"):
            errs[k] = rel(traj[int(k[-2:])][0], fx[k])
    return errs


def fp16_reference_regime_vs_fixture(dev):
    """The oracle in the reference's own fp16 arithmetic (oracle/precision_regimes.py "fp16": stock torch ops on the GPU, fp16
    weights / activations, scheduler state fp32) against the same fp32 fixture: the noise floor of "fp16" at this configuration."""
    from oracle import precision_regimes as PR
    from oracle.make_fullsize_golden import ddim_inputs
    fx, meta = load_fixture("full_ddim_512_50.safetensors")
    o_unet = oracle_unet_sd15()
    torch.manual_seed(1)
    o_clora = ControlLoRARef.from_config(os.path.join(ROOT, "configs", meta["config"]))
    randomize_adapters_(o_clora, seed=1, std=0.02)
    o_unet.set_attn_processor(map_processors_to_unet(o_unet, o_clora))
    guide, cond, uncond, lat0 = ddim_inputs(int(meta["res"]), int(meta["images"]), int(meta["input_seed"]))
    u, c = PR.build_regime(o_unet, o_clora, "fp16", dev)
    x, _, eps1 = PR.ddim_loop(u, c, guide, cond, uncond, lat0, int(meta["steps"]), float(meta["guidance_scale"]))
    out = {"latents": rel(x, fx["latents"]), "eps_step01": rel(eps1, fx["eps_step01"])}
    del u, c
    torch.cuda.empty_cache()
    return out
