"""CPU end-to-end test of the HOST LOGIC (autograd functions, module plumbing, processor mapping, trainer)
with the kernels running under the host emulator: one reference-style train step on the seeded small
cases must reproduce the golden vectors produced by the reference's own models.py."""
import pytest
import torch

from tests import e2e_cases as E
from tests.emu_fixture import use_emulator


@pytest.fixture(autouse=True)
def _emu():
    with use_emulator():
        yield


@pytest.mark.parametrize("case", ["v1", "v2", "sketch", "lora", "postadd", "postadd_concat"])
def test_train_step_matches_reference_golden(case, golden_dir):
    errs = E.check_against_golden(case, "cpu", golden_dir)
    print(case, errs)


@pytest.mark.parametrize("kind", ["v1", "v2"])
def test_pre_post_lora_chain_matches_oracle(kind):
    """host logic of the chained-adapter path on the emulated kernels (the GPU suite runs the same check on the real
    library: tests/test_e2e_gpu.py)"""
    print(kind, E.check_pre_post_chain(kind, "cpu"))


@pytest.mark.parametrize("kind", ["v1", "v2", "lora"])
def test_processors_run_on_a_stock_cross_attention_module(kind):
    """reference models.py:122-150: the processors on a module with only the diffusers `CrossAttention` surface (host logic of
    models.StockAttentionHost on the emulated kernels; the GPU suite repeats it at a real width)"""
    print(kind, E.check_stock_attention_host(kind, "cpu"))


def test_trainer_accumulation_lr_schedule_and_resume(golden_dir):
    """host logic of gradient accumulation / LR multiplier / checkpoint-resume with synthetic gradients (the full
    version with real forward/backward passes runs in the GPU suite: tests/test_e2e_gpu.py)"""
    E.check_trainer_features("cpu", golden_dir, real_backward=False)


def test_resume_from_an_accelerate_format_checkpoint(tmp_path):
    """reference train_text_to_image_control_lora.py:713-735 (`accelerator.save_state`) / :705-722 (`load_state`): a `checkpoint-N`
    directory in accelerate's layout -- pytorch_model.bin + the state_dict of a REAL torch.optim.AdamW built from
    `control_lora.parameters()` + GradScaler / LambdaLR state -- resumes the flat trainer: same moments, same step count, and
    the NEXT optimizer step lands on the same parameters as torch's AdamW does.  Round trip through save_accelerate_state."""
    from controllora_amd.train import ControlLoRATrainer
    from oracle import cases
    unet, clora, _ = E.build_product_case("v1", "cpu")
    ref = __import__("copy").deepcopy(clora).float()
    params = [p for p in ref.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    g = torch.Generator().manual_seed(5)
    for _ in range(3):
        for p_ in params:
            p_.grad = torch.randn(p_.shape, generator=g) * 1e-2
        opt.step(); sched.step()
    d = tmp_path / "checkpoint-3"
    d.mkdir()
    torch.save(ref.state_dict(), d / "pytorch_model.bin")
    torch.save(opt.state_dict(), d / "optimizer.bin")
    torch.save(sched.state_dict(), d / "scheduler.bin")
    torch.save({"scale": 1024.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 7}, d / "scaler.pt")
    tr = ControlLoRATrainer(unet, clora, lr=1e-3, max_grad_norm=0.0, dynamic_scale=False)
    tr.load_state(str(d))
    assert tr.global_step == 3 and float(tr.state[2]) == 3.0 and float(tr.state[3]) == 1024.0 and float(tr.state[4]) == 7.0
    for (p_, off), q in zip(tr._module_order_views(), params):
        assert torch.equal(p_.detach().cpu(), q.detach()), "weights"
        assert torch.equal(tr.flat.exp_avg[off:off + q.numel()].cpu().reshape(q.shape), opt.state[q]["exp_avg"])
        assert torch.equal(tr.flat.exp_avg_sq[off:off + q.numel()].cpu().reshape(q.shape), opt.state[q]["exp_avg_sq"])
    # the next step: the same gradient through torch's AdamW and through the flat kernels (scaled by the loss scale)
    tr.flat.zero_grad()
    for (p_, off), q in zip(tr._module_order_views(), params):
        q.grad = torch.randn(q.shape, generator=g) * 1e-2
        p_.grad.copy_(q.grad * 1024.0)
    opt.step()
    tr._optimizer_kernels()
    worst = max(float((p_.detach().cpu() - q.detach()).abs().max()) for (p_, _), q in zip(tr._module_order_views(), params))
    assert worst < 2e-7, worst
    # and back: what save_accelerate_state writes loads into a fresh torch AdamW
    tr.save_accelerate_state(str(tmp_path / "checkpoint-4"))
    opt2 = torch.optim.AdamW([torch.nn.Parameter(q.detach().clone()) for q in params], lr=1e-3)
    opt2.load_state_dict(torch.load(tmp_path / "checkpoint-4" / "optimizer.bin"))
    st = opt2.state_dict()["state"]
    assert len(st) == len(params) and int(st[0]["step"]) == 4
    assert torch.allclose(st[0]["exp_avg"], opt.state[params[0]]["exp_avg"], atol=1e-8)
    # the step count comes from the optimizer state, scheduler.bin is in accelerate's unit (the wrapped scheduler is
    # stepped num_processes times per optimizer step): a checkpoint of an 8-process run resumes at step 3, not 24
    torch.save({"last_epoch": 24, "_step_count": 25}, d / "scheduler.bin")
    tr.load_state(str(d))
    assert tr.global_step == 3 and tr.resumed_scheduler_ratio == 8.0
    assert tr.sched_epoch == 24                    # the schedule continues where the saving run's wrapped LambdaLR stood
    # the LR multiplier is evaluated in accelerate's units: `last_epoch` advances by the process count per optimizer step
    seen = []
    tr.lr_lambda = lambda e: (seen.append(e), 1.0)[1]
    tr.world = 4
    tr._all_reduce_grads = lambda: None          # no process group in this test: only the schedule arithmetic is exercised
    tr.optimizer_step()
    assert seen == [24] and tr.sched_epoch == 28 and tr.global_step == 4
    tr.save_accelerate_state(str(tmp_path / "checkpoint-5"))
    sd = torch.load(tmp_path / "checkpoint-5" / "scheduler.bin", weights_only=False)
    assert sd["last_epoch"] == 28 and sd["_step_count"] == 29
    # ... and the file is a complete LambdaLR state dict: torch's own scheduler loads it (what accelerate.load_state does)
    sch = torch.optim.lr_scheduler.LambdaLR(torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=1e-3), lambda e: 1.0)
    sch.load_state_dict(sd)
    assert sch.last_epoch == 28


def test_adapter_packs_are_refreshed_behind_flat_updates(monkeypatch):
    """(a) a pack group registered after the optimizer graph was captured is unknown to the captured repack launch --
    `_AdapterPacks.epoch` moves and step_graphed repacks eagerly; (b) ControlLoRATrainer.load_state_dict copies into the flat
    buffer without bumping any parameter's _version, so it must repack itself."""
    from controllora_amd import ops
    from controllora_amd.train import ControlLoRATrainer
    from tests.emu_fixture import use_emulator
    unet, clora, _ = E.build_product_case("v1", "cpu")
    with use_emulator():
        packs = ops._AdapterPacks()
        D1, D2 = torch.randn(4, 64), torch.randn(4, 64)
        e0 = packs.epoch
        a = packs.get([D1]).clone()
        assert packs.epoch == e0 + 1
        packs.get([D1])
        assert packs.epoch == e0 + 1                      # same group: no new epoch
        packs.get([D1, D2])
        assert packs.epoch == e0 + 2
        with torch.no_grad():
            D1.data.mul_(2.0)                             # what the flat AdamW kernel does: the bytes change, _version does not
        v = D1._version
        assert torch.equal(packs.get([D1]), a) and D1._version == v     # stale until someone repacks ...
        packs.repack_all()
        assert torch.equal(packs.get([D1])[:4].float(), D1.half().float())         # ... rows 0..3 = fp16(D)
        tr = ControlLoRATrainer(unet, clora, lr=1e-3)
        calls = []
        monkeypatch.setattr(ops, "repack_adapters", lambda: calls.append(1))
        tr.load_state_dict(tr.state_dict())
        assert calls, "load_state_dict must refresh the adapter operand packs"


def test_hint_encoder_conv_operands_are_persistent_and_refreshed():
    """Round 6: the fp16 GEMM operands of the trainable hint-encoder convolutions (reference models.py:470, 529, 594-597, 684) are
    persistent buffers refreshed by ONE multi-job launch per optimizer step (ops.TRAIN_CONV_PACKS) instead of one launch per
    convolution per forward, and the staging -> OIHW gradient unpack of all of them is ONE launch at the end of the backward.
    (a) multi-job pack == single-job pack, bit for bit; (b) an in-place torch update (the _version moves) is caught at the point of
    use; (c) a flat-buffer update (no _version change) is stale until repack_adapters(), which the trainer calls after AdamW;
    (d) deferred multi-job unpack == immediate single-job unpack on the gradients of a whole hint-encoder backward."""
    from controllora_amd import kernels as K
    from controllora_amd import ops
    from tests.emu_fixture import use_emulator
    with use_emulator():
        g = torch.Generator().manual_seed(3)
        ws = [torch.nn.Parameter(torch.randn(16, 3, 3, 3, generator=g)), torch.nn.Parameter(torch.randn(24, 16, 3, 3, generator=g)),
              torch.nn.Parameter(torch.randn(40, 24, 1, 1, generator=g))]
        cips = [8, 16, 24]
        single = [K.conv_weight_pack(w.detach(), cip, True) for w, cip in zip(ws, cips)]
        packs = ops._TrainConvPacks()
        got = [packs.get(w, cip, True) for w, cip in zip(ws, cips)]
        for (f1, d1), (f2, d2) in zip(single, got):
            assert torch.equal(f1, f2) and torch.equal(d1, d2)
        ptrs = [f.data_ptr() for f, _ in got]
        with torch.no_grad():
            ws[0].mul_(2.0)                                   # torch-visible update: caught by the version check
        f0, _ = packs.get(ws[0], cips[0], True)
        assert f0.data_ptr() == ptrs[0] and torch.equal(f0, K.conv_weight_pack(ws[0].detach(), cips[0], True)[0])
        with torch.no_grad():
            ws[1].data.mul_(0.5)                              # what the flat AdamW kernel does: bytes change, _version does not
            ws[2].data.add_(1.0)
        stale, _ = packs.get(ws[1], cips[1], True)
        assert torch.equal(stale, single[1][0])
        packs.repack_all()                                    # ONE launch for all three
        for w, cip, p0 in zip(ws, cips, ptrs):
            f, d = packs.get(w, cip, True)
            f1, d1 = K.conv_weight_pack(w.detach(), cip, True)
            assert f.data_ptr() == p0 and torch.equal(f, f1) and torch.equal(d, d1)
        # (d) the same hint-encoder backward with the deferred multi-job unpack and with one unpack per convolution
        grads = {}
        for defer in (True, False):
            K.DEFER_UNPACK = defer
            _, clora, _ = E.build_product_case("v1", "cpu")
            from oracle import cases
            outs = clora(cases.seeded_inputs()["guide"].half()).control_states
            loss = sum((o.float() ** 2).sum() for o in outs)
            loss.backward()
            K.lora_wgrad_flush()
            grads[defer] = [p.grad.clone() for n, p in clora.named_parameters() if p.grad is not None and p.ndim == 4]
        K.DEFER_UNPACK = True
        assert len(grads[True]) == len(grads[False]) >= 5
        for a, b in zip(grads[True], grads[False]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))      # fp32 atomics: order differs run to run


@pytest.mark.parametrize("case", ["v1", "v2", "lora"])
def test_grouped_text_kv_projection_equals_per_site(case):
    """Round 6: in training the k | v projections of all cross-attention sites of one width run as ONE GEMM at the head of the UNet
    forward (ops.TextKVGroup; reference models.py:128-136, 249-265 evaluates them site by site) and their adapters' backward as one
    multi-job launch after the group's last site: same prediction and same gradients as the per-site path."""
    from controllora_amd import ops
    from tests.emu_fixture import use_emulator
    res = {}
    with use_emulator():
        for grouped in (True, False):
            ops.GROUP_TEXT_KV = grouped
            calls = []
            orig = ops.TextKVGroup.__init__

            def spy(self, *a, **k):
                calls.append(1)
                return orig(self, *a, **k)
            ops.TextKVGroup.__init__ = spy
            try:
                out, _ = E.run_product_step(case, "cpu")
            finally:
                ops.TextKVGroup.__init__ = orig
                ops.GROUP_TEXT_KV = True
            assert bool(calls) == grouped, (grouped, len(calls))
            res[grouped] = out
    assert E.rel(res[True]["pred"], res[False]["pred"]) < 1e-6
    assert E.rel(res[True]["grads"], res[False]["grads"]) < 2e-5, E.rel(res[True]["grads"], res[False]["grads"])


@pytest.mark.parametrize("case", ["v1", "v2"])
def test_deferred_finishes_and_in_place_concat_do_not_change_a_train_step(case):
    """Round 6: (a) split-K finishes folded into the GroupNorm / LayerNorm launch that reads the GEMM's output (kernels._PENDING,
    include/clora.h clora_deferred_t) and (b) the up path's cat([x, skip]) read in place by norm1 (ops._GroupNormCatFn) are pure
    launch merges: prediction, loss and every gradient of a train step are BIT-identical to the unmerged path."""
    from controllora_amd import kernels as K
    from controllora_amd import unet as U
    from tests.emu_fixture import use_emulator
    res = {}
    with use_emulator():
        for merged in (True, False):
            K.DEFER_FINISH, U.CAT_IN_PLACE = merged, merged
            taken = []
            orig = K.take_pending

            def spy(t):
                r = orig(t)
                taken.append(r[0] is not None)
                return r
            K.take_pending = spy
            try:
                out, _ = E.run_product_step(case, "cpu")
            finally:
                K.take_pending = orig
                K.DEFER_FINISH, U.CAT_IN_PLACE = True, True
            assert any(taken) == merged and not K._PENDING
            res[merged] = out
    for k in ("pred", "loss", "grads"):
        assert torch.equal(res[True][k], res[False][k]), k


def test_vae_encode_decode_matches_oracle():
    from tests import vae_cases
    print(vae_cases.check_vae("cpu", res=32, batch=1))


def test_clip_text_encoder_matches_transformers():
    """controllora_amd/clip.py on the emulated kernels vs the stock transformers CLIPTextModel (fp32 CPU): small config, causal mask"""
    from tests import clip_cases
    print(clip_cases.check_clip("cpu", batch=2, seq=77))
    print(clip_cases.check_clip("cpu", batch=1, seq=20, seed=3))


def test_text_encoder_loads_a_transformers_checkpoint_folder(tmp_path):
    """`text_encoder/` in the SD-1.5 layout (config.json + model.safetensors written by transformers' save_pretrained) loads
    into the clora CLIP and reproduces the transformers model's last hidden state"""
    from transformers import CLIPTextConfig, CLIPTextModel
    from controllora_amd import text
    from tests import clip_cases
    torch.manual_seed(5)
    ref = CLIPTextModel(CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=0, eos_token_id=999, pad_token_id=999,
                                       **clip_cases.SMALL_CLIP)).eval()
    with torch.no_grad():
        for p_ in ref.parameters():
            p_.copy_(p_.half().float())
    ref.save_pretrained(tmp_path / "text_encoder")
    enc = text.load_text_encoder(str(tmp_path), "cpu")
    ids = torch.randint(0, 1000, (2, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = ref(input_ids=ids).last_hidden_state
    assert clip_cases.rel(enc(ids)[0], want) < 3e-3


@pytest.mark.parametrize("case", ["v1", "v2"])
def test_inference_with_control_batch_broadcast(case):
    print(case, E.check_inference_broadcast(case, "cpu"))


@pytest.mark.parametrize("case", ["v2", "sketch"])
def test_control_batch_repeat_interleave(case):
    print(case, E.check_control_batch_repeat_interleave(case, "cpu"))


def test_control_batch_mismatch_raises_in_the_plain_path():
    """reference models.py:237-238: `hidden_states + process_control_states(...)` cannot broadcast 2 control samples over 4"""
    from oracle import cases
    unet, _, clora = E.build_product_case("v1", "cpu")
    inp = cases.seeded_inputs(batch=4)
    with torch.no_grad():
        clora(inp["guide"][:2].half())
        with pytest.raises((ValueError, RuntimeError, AssertionError)):
            unet(inp["latents"].half(), 501, inp["ehs"].half())


def test_ddim_text_kv_cache_is_exact():
    """the per-loop cache of the cross-attention K/V projections (pipeline.ddim_sample) changes nothing but the work"""
    from controllora_amd import models as M
    from controllora_amd.pipeline import ddim_sample
    from oracle import cases
    unet, _, clora = E.build_product_case("v1", "cpu")
    inp = cases.seeded_inputs()
    args = (unet, clora, inp["guide"][:1].half(), inp["ehs"][:1].half(), inp["ehs"][1:2].half())
    a = ddim_sample(*args, steps=3, guidance_scale=5.0, latents=inp["latents"][:1].half(), cache_text_kv=False)
    calls = []
    orig = M.ops.lora_proj
    M.ops.lora_proj = lambda x, pack, segs, residual=None, **kw: (calls.append(x.shape[0]), orig(x, pack, segs, residual, **kw))[1]
    try:
        b = ddim_sample(*args, steps=3, guidance_scale=5.0, latents=inp["latents"][:1].half(), cache_text_kv=True)
    finally:
        M.ops.lora_proj = orig
    assert torch.equal(a, b)
    kv_rows = 2 * cases.CTX_LEN                         # CFG batch 2 x text length: the K/V projections' row count
    assert sum(1 for m in calls if m == kv_rows) == 16  # 16 cross-attention sites, projected once for all 3 steps


def test_pipeline_prompt_to_image_on_the_emulator():
    """reference apps/gradio_canny2image.py:66-92 end to end on the small random model with the kernels emulated: CLIP text encode
    (controllora_amd/clip.py), hint encode, CFG DDIM loop, VAE decode; one guide broadcast over the samples, one guide per image
    (tiled like the CFG batch), and a guide batch with no defined pairing is rejected."""
    from controllora_amd import models as M
    from controllora_amd.pipeline import ControlLoRAPipeline
    from oracle import cases
    torch.manual_seed(0)
    clora = M.ControlLoRA(**cases.SMALL_CLORA_V1)
    with torch.no_grad():                                  # fresh adapters have zero `up` matrices: the guide would have no effect
        for n, p_ in clora.named_parameters():
            if n.endswith(".up.weight"):
                p_.normal_(0.0, 0.2)
    pipe = ControlLoRAPipeline.from_pretrained("random:small", clora, "cpu")
    guide = torch.rand(2, 3, 64, 64) * 2 - 1
    a = pipe("red circle", guide[:1], num_samples=2, ddim_steps=2, scale=5.0, seed=5)
    assert a.shape == (2, 64, 64, 3) and a.dtype == torch.uint8
    b = pipe("red circle", guide, num_samples=2, ddim_steps=2, scale=5.0, seed=5)          # one guide per image
    # image 0 sees the same guide both times, image 1 does not.  Not bit-equal: the hint encoder runs at control batch 1 vs 2 and the
    # GroupNorm team plan (like the GEMM launch table) partitions its fp32 sums by the problem size -- a uint8 level here and there
    d0 = (a[0].int() - b[0].int()).abs().float()
    d1 = (a[1].int() - b[1].int()).abs().float()
    # (measured: image 0 differs by <= 2 levels, 0.18 on average; image 1 by 5.9 on average, up to 40)
    assert b.shape == a.shape and d0.max() <= 3 and d0.mean() < 0.5 and d1.mean() > 10 * d0.mean() and d1.max() > 20
    with pytest.raises(ValueError):
        pipe("red circle", torch.rand(3, 3, 64, 64), num_samples=2, ddim_steps=2, seed=5)


def test_pose_app_process_call_pattern_on_the_emulator():
    """reference apps/gradio_pose2image.py:68-96 with the pose map supplied instead of detected (the OpenPose annotator is out of
    scope): nearest-neighbour resize of the map to the generation size (line 77), BGR flip + /127.5 - 1 control tensor (line 79),
    the reference's return convention [detected_map] + images, the 30-step DPM-Solver++ default of the app; a different pose map
    gives a different image, the same one the same image."""
    import importlib.util
    import os
    import numpy as np
    from controllora_amd import models as M
    from controllora_amd.pipeline import ControlLoRAPipeline
    from oracle import cases
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pose2image", os.path.join(root, "apps", "pose2image.py"))
    app = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(app)
    # nearest resize == cv2.INTER_NEAREST's index rule on an up- and a down-scale
    src = np.arange(6 * 4 * 3, dtype=np.uint8).reshape(6, 4, 3)
    up = app.nearest_resize(src, 8, 12)
    assert up.shape == (12, 8, 3) and np.array_equal(up[::2, ::2], src) and np.array_equal(up[1::2, 1::2], src)
    assert np.array_equal(app.nearest_resize(src, 2, 3), src[::2, ::2])
    torch.manual_seed(0)
    clora = M.ControlLoRA(**cases.SMALL_CLORA_V1)
    with torch.no_grad():
        for n, p_ in clora.named_parameters():
            if n.endswith(".up.weight"):
                p_.normal_(0.0, 0.2)
    pipe = ControlLoRAPipeline.from_pretrained("random:small", clora, "cpu")
    rng = np.random.default_rng(0)
    pose = (rng.random((32, 32, 3)) > 0.8).astype(np.uint8) * 255            # a small "skeleton" map, resized to 64 x 64 by process()
    photo = np.zeros((128, 128, 3), np.uint8)
    seen = {}
    orig = pipe.__class__.__call__

    def spy(self, prompt, control, **kw):
        seen.update(control=control.clone(), kw=kw)
        return orig(self, prompt, control, **kw)
    pipe.__class__.__call__ = spy
    try:
        out = app.process(pipe, photo, "a man", "best quality", "lowres", 1, 64, 64, 2, 5.0, 7, 0.0, pose_map=pose)
    finally:
        pipe.__class__.__call__ = orig
    assert len(out) == 2 and out[0].shape == (64, 64, 3) and out[1].shape == (64, 64, 3) and out[1].dtype == np.uint8
    assert np.array_equal(out[0], app.nearest_resize(pose, 64, 64))                      # [detected_map] first, un-inverted (pose app)
    want = torch.from_numpy(out[0][..., ::-1].copy().transpose(2, 0, 1)).float()[None] / 127.5 - 1.0
    assert torch.equal(seen["control"], want) and seen["kw"]["sampler"] == "dpm" and seen["kw"]["ddim_steps"] == 2
    again = app.process(pipe, photo, "a man", "best quality", "lowres", 1, 64, 64, 2, 5.0, 7, 0.0, pose_map=pose)
    other = app.process(pipe, photo, "a man", "best quality", "lowres", 1, 64, 64, 2, 5.0, 7, 0.0, pose_map=255 - pose)
    assert np.array_equal(out[1], again[1]) and not np.array_equal(out[1], other[1])


@pytest.mark.parametrize("M,fuse", [(128, "0"), (96, "1")])
def test_lora_proj_counts_a_precomputed_second_input_once(M, fuse, monkeypatch):
    """on the NON-fused path `_LoraProjFn.forward` handed the down-projection job both the materialised second input
    (X2 = c) and its precomputed share (T_in = t_pre = c . D^T), so T = (h + c) . D^T + c . D^T.  With a t_pre the job must read h
    alone; the result equals the call without t_pre (which evaluates L(h + c) = L(h) + L(c) itself).  M = 128 / 96, N = K = 320
    have no fused plan (too few blocks / down-projection fusion switched off)."""
    from controllora_amd import ops
    from tests.emu_fixture import use_emulator
    monkeypatch.setattr(ops, "FUSE_DOWN", fuse == "1")
    g = torch.Generator().manual_seed(3)
    Kd = N = 320
    x = (torch.randn(M, Kd, generator=g) * 0.5).half()
    c = (torch.randn(M, Kd, generator=g) * 0.5).half()
    W = torch.randn(N, Kd, generator=g) / Kd ** 0.5
    D = torch.randn(4, Kd, generator=g) / Kd ** 0.5
    U = torch.randn(N, 4, generator=g) * 0.3
    pack = ops.LinearPack(W, torch.zeros(N))
    with use_emulator():
        assert ops._fuse_plan(M, N, Kd, N) is None or fuse == "1"
        if ops._fusable(pack, ((((0, 1)), 1.0),), [4], N, M) is not None:
            pytest.skip("shape became fusable: the test needs the non-fused path")
        ref = ops.lora_proj(x, pack, [((x, c), D, U, 1.0)])
        t_pre = c.float() @ D.t()
        got = ops.lora_proj(x, pack, [((x, c), D, U, 1.0)], t_pre=t_pre.contiguous())
    want = (x.float() @ W.t().half().float()) + ((x.float() + c.float()) @ D.t()) @ U.t()
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(ref, want) < 2e-3
    assert rel(got, want) < 2e-3, rel(got, want)                       # 0.35 with the double count


@pytest.mark.parametrize("steps", [4, 16])
def test_dpm_solver_sampling_loop_matches_an_independent_restatement(steps):
    """reference train_text_to_image_control_lora.py:811-843 / the apps' `DPMSolverMultistepScheduler`: the product loop
    (`ddim_sample(sampler="dpm")`: CFG batch, uncond first, scheduler state in fp32) against tests/full_cases.oracle_dpm -- the
    DPM-Solver++(2M) update restated from the paper's form around the fp32 oracle UNet.  4 steps: first-order start, one
    second-order step, `lower_order_final`; 16 steps: second order to the end.  Kernels emulated."""
    from oracle import cases
    from tests import full_cases as F
    from tests.emu_fixture import use_emulator
    o_unet, _, o_clora = cases.build_oracle_case("v1")
    p_unet, _, p_clora = E.build_product_case("v1", "cpu")
    with torch.no_grad():
        for p in o_unet.parameters():
            p.copy_(p.half().float())
    with use_emulator():
        r = F.ddim_parity(o_unet, o_clora, p_unet, p_clora, "cpu", res=64, steps=steps, guidance_scale=7.5, nb=1, ctx_dim=64, ctx_len=7,
                          sampler="dpm")
    print("DPM_LOOP_PARITY emu", steps, r)
    assert r["latents"] < 6e-3, r


@pytest.mark.parametrize("concat,Mc,scale", [(True, 96, 1.0), (True, 48, 1.0), (False, 96, 0.5)])
def test_wide_rank_control_adapter_on_the_gemm_kernels(concat, Mc, scale, monkeypatch):
    """configs/danbooru-sketch.json's rank-256 `to_control` (reference models.py:209-218) runs as two Linear layers on the MFMA GEMM /
    weight-gradient kernels (ops._ControlAddWideFn) instead of the rank-r kernels: output, d(h), d(ctrl) and both weight gradients
    against torch autograd of the reference formula in fp32, and against the rank-r path (ops._ControlAddFn) -- rank 32, with and
    without cat(h, ctrl), control batch broadcast (Mc < M), scale != 1."""
    from controllora_amd import ops
    from tests.emu_fixture import use_emulator
    g = torch.Generator().manual_seed(11)
    M, C_, Cc, R = 96, 64, 32, 32
    h = (torch.randn(M, C_, generator=g) * 0.5).half()
    ctrl = (torch.randn(Mc, Cc, generator=g) * 0.5).half()
    Kin = C_ + Cc if concat else Cc
    dy = (torch.randn(M, C_, generator=g) * 0.1).half()

    def run(wide):
        monkeypatch.setattr(ops, "WIDE_RANK", wide)
        D = torch.nn.Parameter(torch.randn(R, Kin, generator=torch.Generator().manual_seed(1)) / Kin ** 0.5)
        U = torch.nn.Parameter(torch.randn(C_, R, generator=torch.Generator().manual_seed(2)) * 0.2)
        hh, cc = h.clone().requires_grad_(True), ctrl.clone().requires_grad_(True)
        with use_emulator():
            y = ops.control_add(hh, cc, D, U, scale, concat)
            y.backward(dy)
        return y.detach().float(), hh.grad.float(), cc.grad.float(), D.grad.clone(), U.grad.clone()

    D = torch.randn(R, Kin, generator=torch.Generator().manual_seed(1)) / Kin ** 0.5
    U = torch.randn(C_, R, generator=torch.Generator().manual_seed(2)) * 0.2
    D.requires_grad_(True); U.requires_grad_(True)
    hf, cf = h.float().requires_grad_(True), ctrl.float().requires_grad_(True)
    cm = cf.repeat(M // Mc, 1)
    x = torch.cat([hf, cm], 1) if concat else cm
    yref = hf + scale * ((x @ D.t()) @ U.t())
    yref.backward(dy.float())
    ref = (yref.detach(), hf.grad, cf.grad, D.grad, U.grad)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    wide, narrow = run(True), run(False)
    names = ("y", "dh", "dctrl", "dD", "dU")
    errs_w = {n: rel(a, b) for n, a, b in zip(names, wide, ref)}
    errs_n = {n: rel(a, b) for n, a, b in zip(names, narrow, ref)}
    print("WIDE_RANK", errs_w, "rank-r path", errs_n)
    for n in names:
        assert errs_w[n] < 1e-3, (n, errs_w)                 # fp16 T / dT between the two GEMMs, as in the reference's fp16 path
        assert errs_n[n] < 1e-3, (n, errs_n)
