"""Product VAE (controllora_amd/vae.py, HIP kernels through the C ABI) against the fp32 restatement
oracle/vae_ref.py on a small seeded configuration: encoder moments, sampled latents, decoder output."""
import torch

from controllora_amd import vae as V
from oracle import vae_ref as R

SMALL_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(16, 32, 32, 64),
                 layers_per_block=1, norm_num_groups=8)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def check_vae(dev, res=32, batch=2, cfg=None, tol=6e-3):
    """cfg=None: the small seeded configuration; cfg=V.SD15_VAE: the SD-1.5 VAE topology at its real widths (128-512 channels,
    the single-head d=512 mid-block attention over (res/8)^2 tokens) -- reference train...:753-754 / apps/gradio_canny2image.py:88-92."""
    cfg = SMALL_VAE if cfg is None else cfg
    torch.manual_seed(3)
    o = R.AutoencoderKL(**cfg)
    with torch.no_grad():
        for n, p in o.named_parameters():                  # non-trivial norms / biases, fp16-representable values
            if p.ndim == 1:
                p.copy_((0.2 * torch.randn_like(p) + (1.0 if "norm" in n and n.endswith("weight") else 0.0)))
            p.copy_(p.half().float())
    m = V.AutoencoderKL(**cfg)
    V.load_from_oracle_(m, o)
    m.to(dev)
    x = (torch.rand(batch, 3, res, res) * 2 - 1).half().float()
    eps = torch.randn(batch, 4, res // 8, res // 8)
    with torch.no_grad():
        mean_o, logvar_o = o.moments(x)
        z_o = o.encode_sample(x, eps)
        img_o = o.decode(z_o.half().float())
    dist = m.encode(x.to(dev).half()).latent_dist
    assert dist.mean.shape == (batch, 4, res // 8, res // 8) and dist.mean.dtype == torch.float32
    errs = {"mean": rel(dist.mean, mean_o), "logvar": rel(dist.logvar, logvar_o)}
    z = dist.sample(noise=eps.to(dev))
    errs["sample"] = rel(z, z_o)
    img = m.decode(z_o.to(dev).half()).sample
    assert img.shape == (batch, 3, res, res)
    errs["decode"] = rel(img, img_o)
    assert max(errs.values()) < tol, errs
    return errs
