"""Product CLIP text encoder (controllora_amd/clip.py: gfx950 kernels through the C ABI) against the stock
`transformers.CLIPTextModel` in fp32 on the CPU with the same (fp16-representable) weights: the last hidden state the reference
feeds to the UNet (reference train_text_to_image_control_lora.py:768).  transformers IS the upstream implementation, so this
boundary is pinned by the real thing, not by a restatement."""
import torch

from controllora_amd import clip as C

SMALL_CLIP = dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                  max_position_embeddings=77)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def check_clip(dev, cfg=None, batch=2, seq=77, seed=0, tol=3e-3):
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = SMALL_CLIP if cfg is None else cfg
    torch.manual_seed(seed)
    ref = CLIPTextModel(CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=0, eos_token_id=cfg["vocab_size"] - 1,
                                       pad_token_id=cfg["vocab_size"] - 1, **cfg)).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():                   # non-trivial biases / norms, fp16-representable values
            if p.ndim == 1:
                p.copy_(0.1 * torch.randn_like(p) + (1.0 if "norm" in n and n.endswith("weight") else 0.0))
            p.copy_(p.half().float())
    m = C.CLIPTextModel(**cfg)
    missing = m.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    # the older checkpoint layout (SD-1.5 `text_encoder/`: keys under `text_model.`) loads too
    m.load_state_dict({("text_model." + k): v for k, v in ref.state_dict().items()}, strict=True)
    m.to(dev)
    g = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(0, cfg["vocab_size"], (batch, seq), generator=g)
    with torch.no_grad():
        want = ref(input_ids=ids).last_hidden_state
    got = m(ids.to(dev))[0]
    assert got.shape == want.shape and got.dtype == torch.float16
    e = rel(got, want)
    assert e < tol, e
    # causality: changing a later token must not change earlier positions (bit-exact), and must change later ones
    ids2 = ids.clone()
    ids2[:, seq // 2] = (ids2[:, seq // 2] + 1) % cfg["vocab_size"]
    got2 = m(ids2.to(dev))[0]
    assert torch.equal(got2[:, :seq // 2], got[:, :seq // 2]) and not torch.equal(got2[:, seq // 2:], got[:, seq // 2:])
    return {"last_hidden_state": e}
