"""The drop-in training entry point (train_text_to_image_control_lora.py): flag surface of the reference's parse_args
(reference train_text_to_image_control_lora.py:84-326), LR schedules, the synthetic fill50k data path (CPU), and on
the GPU a short run with checkpoint + resume + final artefacts (reference :713-735, 796-809, 922-929)."""
import json
import os

import pytest
import torch

import train_text_to_image_control_lora as T
from controllora_amd import data, text

REFERENCE_FLAGS = """pretrained_model_name_or_path revision dataset_name dataset_config_name train_data_dir image_column
guide_column caption_column validation_prompt num_validation_images validation_epochs max_train_samples output_dir cache_dir
seed resolution train_batch_size num_train_epochs max_train_steps gradient_accumulation_steps gradient_checkpointing
learning_rate scale_lr lr_scheduler lr_warmup_steps use_8bit_adam allow_tf32 dataloader_num_workers adam_beta1 adam_beta2
adam_weight_decay adam_epsilon max_grad_norm push_to_hub hub_token hub_model_id logging_dir mixed_precision report_to
local_rank checkpointing_steps resume_from_checkpoint enable_xformers_memory_efficient_attention control_lora_config""".split()


def test_all_reference_flags_are_accepted_with_reference_defaults():
    assert len(REFERENCE_FLAGS) == 44
    a = T.parse_args(["--pretrained_model_name_or_path", "x", "--dataset_name", "d", "--control_lora_config", "c.json"])
    for f in REFERENCE_FLAGS:
        assert hasattr(a, f), f
    assert (a.resolution, a.train_batch_size, a.num_train_epochs, a.learning_rate, a.lr_scheduler, a.lr_warmup_steps) == \
        (512, 16, 100, 1e-4, "constant", 500)
    assert (a.adam_beta1, a.adam_beta2, a.adam_weight_decay, a.adam_epsilon, a.max_grad_norm) == (0.9, 0.999, 1e-2, 1e-8, 1.0)
    assert (a.checkpointing_steps, a.output_dir, a.image_column, a.guide_column, a.caption_column) == \
        (500, "sd-fill50k-model-control-lora", "image", "guide", "text")
    with pytest.raises(ValueError):                      # reference :322-324
        T.parse_args(["--pretrained_model_name_or_path", "x", "--control_lora_config", "c.json"])
    with pytest.raises(SystemExit):                      # --control_lora_config is required (reference :315)
        T.parse_args(["--pretrained_model_name_or_path", "x", "--dataset_name", "d"])


def test_lr_schedules():
    assert [data.lr_lambda("constant", 5, 100)(s) for s in (0, 50)] == [1.0, 1.0]
    f = data.lr_lambda("constant_with_warmup", 10, 100)
    assert f(0) == 0.0 and f(5) == 0.5 and f(10) == 1.0 and f(99) == 1.0
    f = data.lr_lambda("linear", 10, 110)
    assert f(5) == 0.5 and f(10) == 1.0 and abs(f(60) - 0.5) < 1e-12 and f(110) == 0.0
    f = data.lr_lambda("cosine", 0, 100)
    assert f(0) == 1.0 and abs(f(50) - 0.5) < 1e-12 and abs(f(100)) < 1e-12
    f = data.lr_lambda("polynomial", 0, 100)
    assert f(0) == 1.0 and abs(f(100) - 1e-7) < 1e-12 and f(1000) == 1e-7
    assert data.lr_lambda("cosine_with_restarts", 0, 100)(100) == 0.0
    with pytest.raises(ValueError):
        data.lr_lambda("nope", 0, 1)


def test_synthetic_fill50k_and_tokenizer():
    tok = text.HashTokenizer()
    ds = data.SyntheticFill50k(64, 8, seed=42, tokenizer=tok)
    a, b = ds[3], ds[3]
    assert torch.equal(a["pixel_values"], b["pixel_values"]) and a["caption"] == b["caption"]       # deterministic
    assert a["pixel_values"].shape == (3, 64, 64) and float(a["pixel_values"].min()) >= -1 and float(a["pixel_values"].max()) <= 1
    g = a["guide_values"]
    assert set(g.unique().tolist()) == {-1.0, 1.0} and 0 < float((g > 0).float().mean()) < 0.2       # thin outline
    ids = a["input_ids"]
    assert ids.shape == (77,) and ids[0] == text.BOS and ids[-1] == text.EOS
    batch = data.collate([ds[0], ds[1]])
    assert batch["pixel_values"].shape == (2, 3, 64, 64) and batch["input_ids"].shape == (2, 77)


def test_entry_point_refuses_to_run_without_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="MI355X"):
        T.main(["--pretrained_model_name_or_path", "random:small", "--dataset_name", "synthetic:fill50k",
                "--control_lora_config", "configs/fill50k.json", "--output_dir", str(tmp_path)])


@pytest.mark.gpu
def test_short_run_checkpoint_resume_and_artifacts(tmp_path):
    from oracle import cases
    cfg = tmp_path / "small.json"
    cfg.write_text(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cases.SMALL_CLORA_V1.items()}))
    out = tmp_path / "run"
    common = ["--pretrained_model_name_or_path", "random:small", "--dataset_name", "synthetic:fill50k", "--control_lora_config", str(cfg),
              "--output_dir", str(out), "--resolution", "64", "--train_batch_size", "2", "--max_train_samples", "16", "--seed", "3",
              "--mixed_precision", "fp16", "--checkpointing_steps", "2", "--lr_scheduler", "constant_with_warmup", "--lr_warmup_steps", "2"]
    assert T.main(common + ["--max_train_steps", "4"]) == 4
    assert sorted(d for d in os.listdir(out) if d.startswith("checkpoint-")) == ["checkpoint-2", "checkpoint-4"]
    for f in ("config.json", "diffusion_pytorch_model.bin", "diffusion_pytorch_model.safetensors"):
        assert (out / f).exists(), f
    assert (out / "checkpoint-4" / "trainer_state.safetensors").exists()
    logs = [json.loads(l) for l in open(out / "logs" / "train_log.jsonl")]
    assert logs[0]["step"] == 1 and all(l["step_loss"] == l["step_loss"] for l in logs)            # finite losses
    # resume: continues at step 5 from checkpoint-4 and trains the adapters further (hipGraph and eager paths)
    assert T.main(common + ["--max_train_steps", "6", "--resume_from_checkpoint", "latest", "--no_hipgraph",
                            "--validation_prompt", "red circle with blue background", "--num_validation_images", "1"]) == 6
    assert (out / "checkpoint-6").exists() and len(os.listdir(out / "validation")) == 1
    from controllora_amd import models as M
    m = M.ControlLoRA.from_pretrained(str(out))
    ups = [p for n, p in m.named_parameters() if n.endswith("to_q_lora.up.weight")]
    assert any(float(p.abs().max()) > 0 for p in ups)                                                # zero-init `up` has moved


def test_checkpoint_directory_loading(tmp_path):
    """`--pretrained_model_name_or_path <dir>`: diffusers-layout `unet/` and `vae/` folders (config.json +
    diffusion_pytorch_model.safetensors / .bin, newer VAE attention key names accepted) load into the product models."""
    from safetensors.torch import save_file
    from controllora_amd import loading, unet as U, vae as V
    u = U.UNet2DConditionModel(**loading.SMALL_UNET)
    U.init_random_(u, seed=5)
    v = V.AutoencoderKL(**loading.SMALL_VAE)
    V.init_random_(v, seed=6)
    (tmp_path / "unet").mkdir(), (tmp_path / "vae").mkdir()
    ucfg = {k: (list(x) if isinstance(x, tuple) else x) for k, x in loading.SMALL_UNET.items()}
    (tmp_path / "unet" / "config.json").write_text(json.dumps(dict(ucfg, _class_name="UNet2DConditionModel", sample_size=8)))
    save_file({k: t.contiguous() for k, t in u.state_dict().items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    vcfg = {k: (list(x) if isinstance(x, tuple) else x) for k, x in loading.SMALL_VAE.items()}
    (tmp_path / "vae" / "config.json").write_text(json.dumps(dict(vcfg, scaling_factor=0.18215)))
    renamed = {}
    for k, t in v.state_dict().items():            # write the VAE with the newer diffusers attention key spelling, as .bin
        for new, old in loading._VAE_RENAMES.items():
            k = k.replace(old, new)
        renamed[k] = t.clone()
    torch.save(renamed, tmp_path / "vae" / "diffusion_pytorch_model.bin")
    u2 = loading.load_unet(str(tmp_path), "cpu")
    v2 = loading.load_vae(str(tmp_path), "cpu")
    assert all(torch.equal(a, b) for a, b in zip(u.state_dict().values(), u2.state_dict().values()))
    assert all(torch.equal(v.state_dict()[k], t) for k, t in v2.state_dict().items())
    (tmp_path / "unet" / "diffusion_pytorch_model.safetensors").unlink()
    with pytest.raises(FileNotFoundError):
        loading.load_unet(str(tmp_path), "cpu")


def test_image_guide_dataset_transform():
    """reference preprocess_train (train_text_to_image_control_lora.py:606-630): short side resized to `resolution`
    (bilinear), Normalize(0.5, 0.5), ONE random crop shared by image and guide, caption tokenised"""
    from PIL import Image
    import numpy as np
    rng = np.random.default_rng(0)
    rows = []
    for w, h in ((96, 64), (64, 80), (64, 64)):
        a = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        rows.append({"image": Image.fromarray(a), "guide": Image.fromarray(255 - a), "text": ["a cat", "a dog"]})
    ds = data.ImageGuideDataset(rows, "image", "guide", "text", 32, text.HashTokenizer())
    for i in range(3):
        ex = ds[i]
        assert ex["pixel_values"].shape == (3, 32, 32) and ex["guide_values"].shape == (3, 32, 32)
        assert float(ex["pixel_values"].abs().max()) <= 1.0
        # guide = inverted image and both got the SAME crop: they stay (almost exactly) opposite in [-1, 1]
        assert float((ex["pixel_values"] + ex["guide_values"]).abs().max()) < 0.05
        assert ex["caption"] in ("a cat", "a dog") and ex["input_ids"].shape == (77,)
    b = data.collate([ds[0], ds[1]])
    assert b["pixel_values"].shape == (2, 3, 32, 32) and b["input_ids"].shape == (2, 77)


def test_canny_annotator_and_app_helpers():
    """the CPU side of the app entry point (apps/canny2image.py): Canny finds the outline of a square and nothing else,
    hysteresis keeps weak pixels only when connected to strong ones, resize rounds to multiples of 64"""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("canny2image", os.path.join(os.path.dirname(os.path.dirname(__file__)), "apps", "canny2image.py"))
    app = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(app)
    img = np.zeros((64, 64, 3), np.uint8)
    img[16:48, 16:48] = 200
    e = app.canny(img, 100, 200)
    assert e.dtype == np.uint8 and set(np.unique(e)) == {0, 255}
    ys, xs = np.nonzero(e)
    assert ys.min() >= 14 and ys.max() <= 49 and xs.min() >= 14 and xs.max() <= 49       # edges hug the square
    assert e[30:34, 30:34].sum() == 0 and e[:10].sum() == 0                               # flat regions stay empty
    assert 100 <= (e > 0).sum() <= 300                                                    # a thin outline, not a band
    weak = np.zeros((32, 32, 3), np.uint8)
    weak[:, 16:] = 30                                                                     # gradient 120 (L1 Sobel): weak only
    assert app.canny(weak, 100, 200).sum() == 0
    assert app.resize_image(np.zeros((100, 150, 3), np.uint8), 128).shape == (128, 192, 3)
    assert app.hwc3(np.zeros((4, 4), np.uint8)).shape == (4, 4, 3)


@pytest.mark.gpu
def test_validation_after_graph_replayed_steps_sees_the_current_adapters(tmp_path):
    """ADVICE r04 (stale adapter packs), end to end through the entry point: `run_validation` (reference train...:811-843, 30-step
    DPM-Solver++) samples at the CFG batch, whose projections register NEW fused-adapter pack groups after the train step's
    optimizer graph was captured; the flat AdamW of the following replayed steps changes the weights without bumping a parameter
    version, so the next validation would mix stale down matrices with fresh up matrices unless step_graphed repacks.  The hipGraph
    run and the eager run (which repacks every group every step) must paint the same validation strips at both checkpoints."""
    import numpy as np
    from PIL import Image
    from oracle import cases
    cfg = tmp_path / "small.json"
    cfg.write_text(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cases.SMALL_CLORA_V1.items()}))
    imgs = {}
    for mode, extra in (("graph", []), ("eager", ["--no_hipgraph"])):
        out = tmp_path / mode
        args = ["--pretrained_model_name_or_path", "random:small", "--dataset_name", "synthetic:fill50k", "--control_lora_config", str(cfg),
                "--output_dir", str(out), "--resolution", "64", "--train_batch_size", "2", "--max_train_samples", "16", "--seed", "3",
                "--mixed_precision", "fp16", "--checkpointing_steps", "3", "--learning_rate", "1e-2", "--max_train_steps", "6",
                "--validation_prompt", "red circle with blue background", "--num_validation_images", "1"] + extra
        assert T.main(args) == 6
        files = sorted(os.listdir(out / "validation"))
        assert len(files) == 2, files
        imgs[mode] = [np.asarray(Image.open(out / "validation" / f)).astype(np.int32) for f in files]
    for a, b in zip(imgs["graph"], imgs["eager"]):
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 2, int(np.abs(a - b).max())          # uint8 strips: identical up to a rounding of the fp16 path
    assert np.abs(imgs["graph"][0] - imgs["graph"][1]).max() > 2           # and the adapters DID move between the two checkpoints
