"""TEST INFRASTRUCTURE (oracle): CPU fp32 restatement of the frozen VAE the reference calls around its hot path
(`AutoencoderKL.from_pretrained(..., subfolder="vae")`, train_text_to_image_control_lora.py:403;
`vae.encode(pixel_values).latent_dist.sample() * 0.18215`, :753-754; the pipeline's decode in
apps/gradio_canny2image.py:88-92).  The arithmetic lives in third-party diffusers (git HEAD >= 0.13.0.dev0,
requirements.txt:4 -- ABSENT from /root/reference and from this image), so this file restates the published
AutoencoderKL / Encoder / Decoder / DownEncoderBlock2D / UpDecoderBlock2D / UNetMidBlock2D / AttentionBlock
algorithm of that version with its state-dict key names.  PARITY UNPINNED: no reference-side golden vector
exists for this component; the product VAE (controllora_amd/vae.py) is tested against this restatement only.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32)
SCALING_FACTOR = 0.18215


class ResnetBlock(nn.Module):
    """ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1)"""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class AttentionBlock(nn.Module):
    """single head over all H*W positions; q and k are each scaled by (C/heads)^-1/4; softmax in fp32"""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.query(h), self.key(h), self.value(h)
        s = 1.0 / math.sqrt(math.sqrt(C))
        p = torch.softmax(torch.bmm(q * s, (k * s).transpose(1, 2)).float(), dim=-1).to(q.dtype)
        o = self.proj_attn(torch.bmm(p, v))
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, cin, cout, layers, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        if down:
            self.downsamplers = nn.ModuleList([_Down(cout)])
        if up:
            self.upsamplers = nn.ModuleList([_Up(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        for m in getattr(self, "downsamplers", []):
            x = m(x)
        for m in getattr(self, "upsamplers", []):
            x = m(x)
        return x


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock(c, c, groups), ResnetBlock(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        boc = block_out_channels
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([
            _Block(boc[max(i - 1, 0)], boc[i], layers_per_block, groups, down=i != len(boc) - 1) for i in range(len(boc))])
        self.mid_block = _Mid(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        boc = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, boc[0], 3, padding=1)
        self.mid_block = _Mid(boc[0], groups)
        self.up_blocks = nn.ModuleList([
            _Block(boc[max(i - 1, 0)], boc[i], layers_per_block + 1, groups, up=i != len(boc) - 1) for i in range(len(boc))])
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def moments(self, x):
        """(mean, logvar) of DiagonalGaussianDistribution; logvar clamped to [-30, 20]"""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def encode_sample(self, x, eps):
        """latent_dist.sample() with the noise passed in (so product and oracle can share it)"""
        mean, logvar = self.moments(x)
        return mean + torch.exp(0.5 * logvar) * eps

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
