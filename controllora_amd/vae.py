"""Frozen, forward-only VAE on the gfx950 kernels: what the reference calls as
`vae.encode(pixel_values).latent_dist.sample() * 0.18215` every train step
(train_text_to_image_control_lora.py:403-406, 753-754) and as the pipeline's final decode
(apps/gradio_canny2image.py:88-92).  Same state-dict keys as upstream `AutoencoderKL` (diffusers >= 0.13:
encoder/decoder `.down_blocks/.up_blocks.{i}.resnets.{j}`, `.mid_block.attentions.0.{group_norm,query,key,value,
proj_attn}`, `quant_conv`, `post_quant_conv`), NHWC fp16 activations, every conv an implicit GEMM with the
bias / residual fused, GroupNorm+SiLU fused.  The single-head d=512 mid-block attention materialises its
scores with the GEMM kernel (head dim is beyond the flash kernels) and normalises them with
clora_softmax_rows_f16; the value bias is applied after P.V (softmax rows sum to 1).
Oracle: oracle/vae_ref.py (restated from the published diffusers algorithm -- parity unpinned, see there).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import kernels as K
from .unet import Conv1x1, Conv3x3, GroupNorm, Linear

f16, f32 = torch.float16, torch.float32

SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32)


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, 1e-6)
        self.conv1 = Conv3x3(cin, cout, need_dgrad=False)
        self.norm2 = GroupNorm(groups, cout, 1e-6)
        self.conv2 = Conv3x3(cout, cout, need_dgrad=False)
        self.conv_shortcut = Conv1x1(cin, cout) if cin != cout else None

    def forward(self, x, H, W):
        B, N, Cin = x.shape
        h = self.conv1(self.norm1(x, True).reshape(B * N, Cin), B, H, W)
        Cout = h.shape[1]
        h = self.norm2(h.reshape(B, N, Cout), True).reshape(B * N, Cout)
        x2 = x.reshape(B * N, Cin)
        sc = self.conv_shortcut(x2) if self.conv_shortcut is not None else x2
        return self.conv2(h, B, H, W, residual=sc).reshape(B, N, Cout)


class VaeAttention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = GroupNorm(groups, c, 1e-6)
        self.query, self.key, self.value, self.proj_attn = Linear(c, c), Linear(c, c), Linear(c, c), Linear(c, c)

    def forward(self, x):
        B, N, C_ = x.shape
        h = self.group_norm(x, False)
        x2 = x.reshape(B * N, C_)
        hq = self.query(h.reshape(B * N, C_))
        hk = self.key(h.reshape(B * N, C_))
        vp = self.value.pack()
        out = torch.empty_like(x2)
        scores = torch.empty((N, N), dtype=f16, device=x.device)
        for b in range(B):
            q, k, hb = hq[b * N:(b + 1) * N], hk[b * N:(b + 1) * N], h[b]
            K.gemm(q, k, N, N, C_, out=scores)                        # q k^T; the (C^-1/4)^2 scale goes into softmax
            K.softmax_rows(scores, 1.0 / math.sqrt(C_), out=scores)
            vt = K.gemm(vp.w, hb, C_, N, C_)                          # V^T = Wv h^T  [C, N]  (bias added after P.V)
            K.gemm(scores, vt, N, C_, N, bias=vp.bias, out=out[b * N:(b + 1) * N])
        return self.proj_attn(out, residual=x2).reshape(B, N, C_)


class _VaeBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        if down:
            self.downsamplers = nn.ModuleList([_Sampler(Conv3x3(cout, cout, stride=2, pad=0, asym_pad=True, need_dgrad=False))])
        if up:
            self.upsamplers = nn.ModuleList([_Sampler(Conv3x3(cout, cout, upsample=True, need_dgrad=False))])

    def forward(self, x, H, W):
        for r in self.resnets:
            x = r(x, H, W)
        B, N, C_ = x.shape
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0].conv(x.reshape(B * N, C_), B, H, W).reshape(B, N // 4, C_)
            H, W = H // 2, W // 2
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0].conv(x.reshape(B * N, C_), B, H, W).reshape(B, N * 4, C_)
            H, W = 2 * H, 2 * W
        return x, H, W


class _Sampler(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.conv = conv


class _VaeMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])
        self.resnets = nn.ModuleList([VaeResnet(c, c, groups), VaeResnet(c, c, groups)])

    def forward(self, x, H, W):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, H, W)), H, W)


def _to_tokens(x_nchw, cpad):
    B, C_, H, W = x_nchw.shape
    t = x_nchw.new_zeros((B, H, W, cpad), dtype=f16)
    t[..., :C_] = x_nchw.permute(0, 2, 3, 1)
    return t.reshape(B * H * W, cpad)


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = Conv3x3(in_channels, boc[0], need_dgrad=False)
        self.down_blocks = nn.ModuleList([
            _VaeBlock(boc[max(i - 1, 0)], boc[i], layers, groups, down=i != len(boc) - 1) for i in range(len(boc))])
        self.mid_block = _VaeMid(boc[-1], groups)
        self.conv_norm_out = GroupNorm(groups, boc[-1], 1e-6)
        self.conv_out = Conv3x3(boc[-1], 2 * latent_channels, need_dgrad=False)

    def forward(self, x):
        B, _, H, W = x.shape
        h = self.conv_in(_to_tokens(x, self.conv_in.pack().Cip), B, H, W).reshape(B, H * W, -1)
        for blk in self.down_blocks:
            h, H, W = blk(h, H, W)
        h = self.mid_block(h, H, W)
        h = self.conv_norm_out(h, True)
        return self.conv_out(h.reshape(B * H * W, -1), B, H, W), H, W          # [B*H*W, 8] tokens


class Decoder(nn.Module):
    def __init__(self, out_channels, latent_channels, boc, layers, groups):
        super().__init__()
        boc = list(reversed(boc))
        self.conv_in = Conv3x3(latent_channels, boc[0], need_dgrad=False)
        self.mid_block = _VaeMid(boc[0], groups)
        self.up_blocks = nn.ModuleList([
            _VaeBlock(boc[max(i - 1, 0)], boc[i], layers + 1, groups, up=i != len(boc) - 1) for i in range(len(boc))])
        self.conv_norm_out = GroupNorm(groups, boc[-1], 1e-6)
        self.conv_out = Conv3x3(boc[-1], out_channels, need_dgrad=False)

    def forward(self, z_tokens, B, H, W):
        h = self.conv_in(z_tokens, B, H, W).reshape(B, H * W, -1)
        h = self.mid_block(h, H, W)
        for blk in self.up_blocks:
            h, H, W = blk(h, H, W)
        h = self.conv_norm_out(h, True)
        return self.conv_out(h.reshape(B * H * W, -1), B, H, W), H, W


class DiagonalGaussian:
    """`latent_dist` of AutoencoderKL.encode: mean / logvar (clamped to [-30, 20]) with .sample() / .mode()"""

    def __init__(self, mean, logvar):
        self.mean, self.logvar = mean, logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _EncodeOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class _DecodeOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, **unused):
        super().__init__()
        self.latent_channels, self.out_channels, self.scaling_factor = latent_channels, out_channels, scaling_factor
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = Conv1x1(2 * latent_channels, 2 * latent_channels)
        self.post_quant_conv = Conv1x1(latent_channels, latent_channels)

    @torch.no_grad()
    def encode(self, x):
        """x [B,3,H,W] in [-1,1] -> latent_dist over [B,4,H/8,W/8] (fp32 moments, as the reference samples them)"""
        B = x.shape[0]
        m, H, W = self.encoder(x)
        L = self.latent_channels
        m = self.quant_conv(m.contiguous())[:, :2 * L].reshape(B, H, W, 2 * L).permute(0, 3, 1, 2).float()
        return _EncodeOutput(DiagonalGaussian(m[:, :L].contiguous(), m[:, L:].contiguous()))

    @torch.no_grad()
    def decode(self, z):
        """z [B,4,h,w] (already divided by the scaling factor) -> image [B,3,8h,8w]"""
        B, _, H, W = z.shape
        zt = self.post_quant_conv(_to_tokens(z, self.post_quant_conv.pack().K))   # channels zero-padded to 8 in and out
        y, H, W = self.decoder(zt, B, H, W)
        return _DecodeOutput(y.reshape(B, H, W, -1)[..., :self.out_channels].permute(0, 3, 1, 2))


def load_from_oracle_(vae: AutoencoderKL, oracle_vae: nn.Module) -> None:
    sd, own = oracle_vae.state_dict(), vae.state_dict()
    assert set(sd) == set(own), (sorted(set(own) - set(sd))[:5], sorted(set(sd) - set(own))[:5])
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[k].reshape(v.shape).to(v.dtype))


def init_random_(vae: AutoencoderKL, seed: int = 0) -> None:
    """seeded synthetic weights (no SD-1.5 checkpoint offline): fan-in scaled normal filters, zero biases, unit norms"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in vae.named_parameters():
            if p.ndim >= 2:
                p.copy_((torch.randn(p.shape, generator=g, dtype=f32) / math.sqrt(p[0].numel())).to(p.dtype))
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
