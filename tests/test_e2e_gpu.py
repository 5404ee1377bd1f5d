"""GPU end-to-end parity (-m gpu): one reference-style train step of the product path (real gfx950 library)
against the golden vectors made by the reference's own models.py, plus the substitute pins of SURVEY.md
section 8c that need the device (zero-init identity, optimizer behaviour, inference loop)."""
import pytest
import torch

from tests import e2e_cases as E

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["v1", "v2", "sketch", "lora", "postadd", "postadd_concat"])
def test_train_step_matches_reference_golden(case, golden_dir):
    errs = E.check_against_golden(case, "cuda", golden_dir)
    print(case, errs)


def test_native_library_is_what_ran():
    from controllora_amd import capi
    assert capi.lib().require_device and capi.lib().path.endswith("_build/libclora.so")
    assert b"gfx950" in capi.lib().cdll.clora_build_info()


@pytest.mark.parametrize("case", ["v1", "v2"])
def test_zero_init_identity_on_gpu(case):
    """a freshly initialised ControlLoRA leaves the UNet output bit-identical."""
    from controllora_amd import models as M, unet as U
    from oracle import cases, unet_ref
    o = unet_ref.UNet2DConditionModel(**cases.SMALL_UNET)
    cases.seeded_weights_(o, seed=11)
    unet = U.UNet2DConditionModel(**cases.SMALL_UNET)
    U.load_from_oracle_(unet, o)
    unet.cuda()
    inp = {k: v.cuda() for k, v in cases.seeded_inputs().items()}
    with torch.no_grad():
        plain = unet(inp["latents"].half(), inp["timesteps"], inp["ehs"].half()).sample.clone()
        torch.manual_seed(0)
        clora = M.ControlLoRA(**cases.CASES[case]).cuda()          # default init: every `up` is zero
        unet.set_attn_processor(M.map_processors_to_unet(unet, clora))
        clora(inp["guide"].half())
        with_adapters = unet(inp["latents"].half(), inp["timesteps"], inp["ehs"].half()).sample
    assert torch.equal(plain, with_adapters)


def test_optimizer_step_moves_parameters_and_is_finite():
    out, trainer = E.run_product_step("v1", "cuda")
    before = trainer.flat.data.clone()
    trainer.optimizer_step()
    torch.cuda.synchronize()
    after = trainer.flat.data
    assert torch.isfinite(after).all() and float((after - before).abs().max()) > 0
    assert float(trainer.state[6]) == 0.0 and float(trainer.state[2]) == 1.0
    # AdamW first step moves every parameter with a non-zero gradient by ~lr
    assert float((after - before).abs().max()) < 2e-4


def test_ddim_inference_loop_runs():
    """Inference call pattern (apps/gradio_canny2image.py:66-92): hint-encode once, CFG batch 2, DDIM steps."""
    from controllora_amd.pipeline import ddim_sample
    unet, params, clora = E.build_product_case("v1", "cuda")
    inp = {k: v.cuda() for k, v in __import__("oracle.cases", fromlist=["x"]).seeded_inputs().items()}
    lat = ddim_sample(unet, clora, inp["guide"][:1].half(), inp["ehs"][:1].half(), inp["ehs"][1:2].half(),
                      steps=4, guidance_scale=9.0, latents=inp["latents"][:1].half())
    assert lat.shape == (1, 4, 16, 16) and torch.isfinite(lat.float()).all()


def test_graph_replay_matches_eager_step():
    """The hipGraph-captured step (bench default) must do exactly the work of the eager step."""
    out = []
    for graphed in (False, True):
        torch.manual_seed(0)
        unet, params, clora = E.build_product_case("v1", "cuda")
        from controllora_amd.train import ControlLoRATrainer
        from oracle import cases, unet_ref
        inp = {k: v.cuda() for k, v in cases.seeded_inputs().items()}
        noisy = unet_ref.DDPMSchedule().add_noise(inp["latents"].cpu(), inp["noise"].cpu(), inp["timesteps"].cpu()).cuda().half()
        tr = ControlLoRATrainer(unet, params, init_scale=128.0, dynamic_scale=False)
        args = (noisy, inp["timesteps"], inp["ehs"].half(), inp["guide"].half(), inp["noise"].half())
        start = tr.flat.data.clone()
        if graphed:
            tr.capture(*args, warmup=1)          # warm-up steps move the parameters: restore them
            tr.flat.data.copy_(start); tr.flat.exp_avg.zero_(); tr.flat.exp_avg_sq.zero_(); tr.state[2] = 0
            from controllora_amd import ops
            ops.repack_adapters()                # a write to the flat buffer behind torch's back: refresh the fp16 operands derived from
            #                                      it (adapter blocks, hint-encoder conv operands), as ControlLoRATrainer.load_state_dict does
            for _ in range(2):
                tr.step_graphed(*args)
        else:
            for _ in range(2):
                tr.step(*args)
        torch.cuda.synchronize()
        out.append((tr.flat.data.clone(), tr.loss(noisy.numel()), float(tr.state[2])))
    (p0, l0, s0), (p1, l1, s1) = out
    assert s0 == s1 == 2.0
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0))
    assert float((p0 - p1).norm() / p0.norm()) < 1e-5


def test_trainer_accumulation_lr_schedule_and_resume(golden_dir):
    E.check_trainer_features("cuda", golden_dir, real_backward=True)


def test_vae_encode_decode_matches_oracle():
    from tests import vae_cases
    print(vae_cases.check_vae("cuda", res=64, batch=2))


def test_clip_text_encoder_matches_transformers_sd15_shape():
    """the frozen CLIP ViT-L/14 text encoder (12 layers, 768 wide, 12 heads of 64, 77 tokens) on the clora
    kernels -- fused q|k|v GEMM, causal flash attention, quick_gelu MLP -- vs the stock transformers model in fp32 on the host
    (reference train...:768 `text_encoder(batch["input_ids"])[0]`)"""
    from controllora_amd import clip as C
    from tests import clip_cases
    print("CLIP_SMALL", clip_cases.check_clip("cuda", batch=2, seq=77))
    print("CLIP_SD15_SHAPE", clip_cases.check_clip("cuda", cfg=C.SD15_CLIP, batch=4, seq=77, tol=3e-3))


def test_vae_sd15_topology_real_widths_matches_oracle():
    """SURVEY section 8 (f)1 / U8 at the bar of the hot path: the SD-1.5 VAE topology at its real widths (128 / 256 / 512 / 512
    channels, two resnets per level, the single-head d = 512 attention over 1024 tokens) at 256x256, product on the GPU vs
    oracle/vae_ref.py on the host: encoder moments, the sampled latents, the decoded image (reference train...:753-754,
    apps/gradio_canny2image.py:88-92)."""
    from controllora_amd import vae as V
    from tests import vae_cases
    errs = vae_cases.check_vae("cuda", res=256, batch=1, cfg=V.SD15_VAE, tol=6e-3)
    print("VAE_SD15_TOPOLOGY_256", {k: f"{v:.2e}" for k, v in errs.items()})


@pytest.mark.parametrize("case", ["v1", "v2", "sketch"])
def test_inference_with_control_batch_broadcast(case):
    print(case, E.check_inference_broadcast(case, "cuda"))


def test_pipeline_prompt_to_image(tmp_path):
    """reference apps/gradio_canny2image.py:66-92 call pattern end to end on the small random model: text encode, hint
    encode once, CFG DDIM loop, VAE decode; deterministic for a fixed seed."""
    import json
    from controllora_amd import models as M
    from controllora_amd.pipeline import ControlLoRAPipeline
    from oracle import cases
    torch.manual_seed(0)
    clora = M.ControlLoRA(**cases.SMALL_CLORA_V1)
    pipe = ControlLoRAPipeline.from_pretrained("random:small", clora, "cuda")
    guide = torch.rand(1, 3, 64, 64) * 2 - 1
    a = pipe("red circle", guide, num_samples=2, ddim_steps=3, scale=7.5, seed=5)
    b = pipe("red circle", guide, num_samples=2, ddim_steps=3, scale=7.5, seed=5)
    assert a.shape == (2, 64, 64, 3) and a.dtype == torch.uint8 and torch.equal(a, b)
    c = pipe("red circle", guide, num_samples=2, ddim_steps=3, scale=7.5, seed=6)
    assert not torch.equal(a, c)


def test_canny2image_app_process():
    """apps/canny2image.py `process()` (reference apps/gradio_canny2image.py:66-92): image -> Canny -> control -> samples"""
    import importlib.util
    import os
    import numpy as np
    from controllora_amd import models as M
    from controllora_amd.pipeline import ControlLoRAPipeline
    from oracle import cases
    spec = importlib.util.spec_from_file_location("canny2image", os.path.join(os.path.dirname(os.path.dirname(__file__)), "apps", "canny2image.py"))
    app = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(app)
    torch.manual_seed(0)
    pipe = ControlLoRAPipeline.from_pretrained("random:small", M.ControlLoRA(**cases.SMALL_CLORA_V1), "cuda")
    img = np.zeros((80, 100, 3), np.uint8)
    img[20:60, 30:70] = 220
    out = app.process(pipe, img, "a square", "best quality", "lowres", 2, 64, 4, 7.5, 3, 0.0, 100, 200)
    assert len(out) == 3 and all(o.dtype == np.uint8 for o in out)
    assert out[0].shape == (64, 64, 3) or out[0].shape == (64, 128, 3)       # resized to multiples of 64
    assert out[1].shape == out[0].shape and (out[0] < 255).any()             # inverted edge map shows the square


def test_c_abi_rccl_exchange_single_rank():
    """SURVEY section 8b `allreduce_flat`: the C ABI's own RCCL communicator (clora_comm_unique_id / clora_comm_init /
    clora_allreduce_flat_f32; reference train...:683-685, 790).  One GPU here: a world of 1 -- the sum over one rank leaves the
    flat gradient buffer bit-identical, on the launch stream, and the trainer path with comm="clora" runs a step."""
    from controllora_amd import capi
    from controllora_amd.train import ControlLoRATrainer
    unet, _, clora = E.build_product_case("v1", "cuda")
    tr = ControlLoRATrainer(unet, clora, init_scale=128.0, dynamic_scale=False, comm="clora")
    assert capi.lib().cdll.clora_comm_world() == 1
    g = tr.flat.grad
    g.copy_(torch.randn(g.shape, generator=torch.Generator().manual_seed(0)).to(g.device))
    before = g.clone()
    tr._all_reduce_grads()
    torch.cuda.synchronize()
    assert torch.equal(g, before)
    tr._init_clora_comm(None, 1)                                   # idempotent: a second trainer joins the existing communicator
    assert capi.lib().cdll.clora_comm_destroy() == 0 and capi.lib().cdll.clora_comm_world() == 0


def test_exchange_captured_inside_the_optimizer_graph_single_rank():
    """Round 6: with the C ABI's communicator the all-reduce is a node of the optimizer hipGraph (RCCL enqueues on the capturing
    stream; reference train...:683-685, 790: DDP reduces inside `accelerator.backward`) -- a rank's step is two graph replays and
    no host-issued collective.  One GPU here: a communicator of one rank (`exchange_at_world_1`), graph replay == eager steps
    on parameters, loss and step count, and the capture must really contain the exchange."""
    from controllora_amd import capi
    from controllora_amd.train import ControlLoRATrainer
    from oracle import cases, unet_ref
    out = []
    for graphed in (False, True):
        torch.manual_seed(0)
        unet, params, clora = E.build_product_case("v1", "cuda")
        inp = {k: v.cuda() for k, v in cases.seeded_inputs().items()}
        noisy = unet_ref.DDPMSchedule().add_noise(inp["latents"].cpu(), inp["noise"].cpu(), inp["timesteps"].cpu()).cuda().half()
        tr = ControlLoRATrainer(unet, params, init_scale=128.0, dynamic_scale=False, comm="clora")
        tr.exchange_at_world_1 = True
        args = (noisy, inp["timesteps"], inp["ehs"].half(), inp["guide"].half(), inp["noise"].half())
        start = tr.flat.data.clone()
        if graphed:
            tr.capture(*args, warmup=1)
            assert tr._exchange_in_graph, getattr(tr, "exchange_capture_error", "the all-reduce was not captured")
            tr.flat.data.copy_(start); tr.flat.exp_avg.zero_(); tr.flat.exp_avg_sq.zero_(); tr.state[2] = 0
            from controllora_amd import ops
            ops.repack_adapters()
            for _ in range(2):
                tr.step_graphed(*args)
        else:
            for _ in range(2):
                tr.forward_backward(*args)
                tr._all_reduce_grads()                 # what world > 1 does eagerly: the sum over one rank
                tr.optimizer_step()
        torch.cuda.synchronize()
        out.append((tr.flat.data.clone(), tr.loss(noisy.numel()), float(tr.state[2])))
        tr.close()
    (p0, l0, s0), (p1, l1, s1) = out
    assert s0 == s1 == 2.0
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0))
    assert float((p0 - p1).norm() / p0.norm()) < 1e-5
