"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU / fp32 / eager-autograd restatement of the *upstream* half of the ControlLoRA hot
path: the SD-1.5 ``UNet2DConditionModel`` and the ``CrossAttention`` /
``LoRALinearLayer`` / ``Downsample2D`` helpers the reference imports from
``diffusers>=0.13.0.dev0`` (reference ``models.py:7-12``, ``requirements.txt:4``,
call sites ``train_text_to_image_control_lora.py:407-409, 487, 782``).

``diffusers`` is NOT under /root/reference and NOT installed here, so this file restates its
published semantics.  PARITY UNPINNED at this boundary: the
reference holds no golden vectors and diffusers itself cannot be run.  What
*is* pinned: the 859,520,964-parameter count of the SD-1.5 topology, the diffusers
state-dict key names, and the behaviour of the reference's own ``models.py`` executed in
place on top of these classes (see ``oracle/diffusers_shim`` and ``oracle/make_golden.py``).

Materialised attention (baddbmm -> softmax -> bmm) exactly like the reference
(``models.py:270-271``); this is also the code timed as ``cpu_baseline`` in bench.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- A1
class LoRALinearLayer(nn.Module):
    """Appendix A1: down N(0, 1/rank), up zeros, fp32 math inside, cast back."""

    def __init__(self, in_features, out_features, rank=4):
        super().__init__()
        if rank > min(in_features, out_features):
            raise ValueError(f"LoRA rank {rank} must be less or equal than {min(in_features, out_features)}")
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states):
        orig_dtype = hidden_states.dtype
        dtype = self.down.weight.dtype
        down_hidden_states = self.down(hidden_states.to(dtype))
        up_hidden_states = self.up(down_hidden_states)
        return up_hidden_states.to(orig_dtype)


# --------------------------------------------------------------------------- A2
class CrossAttnProcessor:
    """Plain attention (what a UNet without adapters runs)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        batch_size, sequence_length, _ = hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
        query = attn.head_to_batch_dim(attn.to_q(hidden_states))
        ehs = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        key = attn.head_to_batch_dim(attn.to_k(ehs))
        value = attn.head_to_batch_dim(attn.to_v(ehs))
        probs = attn.get_attention_scores(query, key, attention_mask)
        hidden_states = attn.batch_to_head_dim(torch.bmm(probs, value))
        hidden_states = attn.to_out[0](hidden_states)
        return attn.to_out[1](hidden_states)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim), nn.Dropout(dropout)])
        self.processor = CrossAttnProcessor()

    def set_processor(self, processor):
        # an nn.Module processor becomes a registered sub-module (same as upstream)
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(
            self, hidden_states, encoder_hidden_states=encoder_hidden_states,
            attention_mask=attention_mask, **cross_attention_kwargs)

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        h = self.heads
        return t.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None):
        if attention_mask is None:
            return None
        raise NotImplementedError("attention masks are never used on this path (SURVEY.md A2)")


# --------------------------------------------------------------------------- A9
class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.padding = padding
        if use_conv:
            self.conv = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.conv = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        if self.use_conv and self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if self.use_conv:
            x = self.conv(x)
        return x


# --------------------------------------------------------------------------- A5
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


# --------------------------------------------------------------------------- A8
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


# --------------------------------------------------------------------------- A7
class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        kw = cross_attention_kwargs or {}
        x = self.attn1(self.norm1(x), **kw) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states, **kw) + x
        x = self.ff(self.norm3(x)) + x
        return x


# --------------------------------------------------------------------------- A6
class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        b, c, h, w = x.shape
        res = x
        x = self.proj_in(self.norm(x))
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states, cross_attention_kwargs)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(x) + res


# --------------------------------------------------------------------------- blocks (A4)
class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, in_c, out_c, temb_c, layers, heads, ctx, groups, eps, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, groups, eps) for i in range(layers)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, out_c // heads, out_c, ctx, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_c, True, out_c, 1, "op")]) if add_downsample else None

    def forward(self, x, temb, ehs, kw):
        outs = ()
        for r, a in zip(self.resnets, self.attentions):
            x = a(r(x, temb), ehs, kw)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class DownBlock2D(nn.Module):
    def __init__(self, in_c, out_c, temb_c, layers, groups, eps, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, groups, eps) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_c, True, out_c, 1, "op")]) if add_downsample else None

    def forward(self, x, temb, ehs=None, kw=None):
        outs = ()
        for r in self.resnets:
            x = r(x, temb)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, temb_c, heads, ctx, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, c // heads, c, ctx, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps), ResnetBlock2D(c, c, temb_c, groups, eps)])

    def forward(self, x, temb, ehs, kw):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ehs, kw)
        return self.resnets[1](x, temb)


class UpBlock2D(nn.Module):
    def __init__(self, in_c, prev_c, out_c, temb_c, layers, groups, eps, add_upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip_c = in_c if i == layers - 1 else out_c
            rin = prev_c if i == 0 else out_c
            rs.append(ResnetBlock2D(rin + skip_c, out_c, temb_c, groups, eps))
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample2D(out_c, True, out_c)]) if add_upsample else None

    def forward(self, x, skips, temb, ehs=None, kw=None):
        for r in self.resnets:
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, in_c, prev_c, out_c, temb_c, layers, heads, ctx, groups, eps, add_upsample):
        super().__init__()
        rs, at = [], []
        for i in range(layers):
            skip_c = in_c if i == layers - 1 else out_c
            rin = prev_c if i == 0 else out_c
            rs.append(ResnetBlock2D(rin + skip_c, out_c, temb_c, groups, eps))
            at.append(Transformer2DModel(heads, out_c // heads, out_c, ctx, groups))
        self.attentions = nn.ModuleList(at)
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample2D(out_c, True, out_c)]) if add_upsample else None

    def forward(self, x, skips, temb, ehs, kw):
        for r, a in zip(self.resnets, self.attentions):
            x = a(r(torch.cat([x, skips.pop()], dim=1), temb), ehs, kw)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """A4: flip_sin_to_cos=True, freq_shift=0 -> cat[cos, sin]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    arg = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_c, time_c):
        super().__init__()
        self.linear_1 = nn.Linear(in_c, time_c)
        self.linear_2 = nn.Linear(time_c, time_c)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


@dataclass
class UNetOutput:
    sample: torch.Tensor


SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5)


class UNet2DConditionModel(nn.Module):
    """SD-1.5 topology (A3/A4); ``attention_head_dim`` is the NUMBER of heads, as upstream."""

    def __init__(self, **cfg):
        super().__init__()
        c = dict(SD15_CONFIG)
        c.update(cfg)
        self.config = SimpleNamespace(**c)
        boc = tuple(c["block_out_channels"])
        heads, ctx, groups, eps, layers = (c["attention_head_dim"], c["cross_attention_dim"],
                                           c["norm_num_groups"], c["norm_eps"], c["layers_per_block"])
        temb_c = boc[0] * 4
        self.conv_in = nn.Conv2d(c["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, t in enumerate(c["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            last = i == len(boc) - 1
            if t == "CrossAttnDownBlock2D":
                self.down_blocks.append(CrossAttnDownBlock2D(in_c, out_c, temb_c, layers, heads, ctx, groups, eps, not last))
            else:
                self.down_blocks.append(DownBlock2D(in_c, out_c, temb_c, layers, groups, eps, not last))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb_c, heads, ctx, groups, eps)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(c["up_block_types"]):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            last = i == len(boc) - 1
            if t == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(in_c, prev_c, out_c, temb_c, layers + 1, heads, ctx, groups, eps, not last))
            else:
                self.up_blocks.append(UpBlock2D(in_c, prev_c, out_c, temb_c, layers + 1, groups, eps, not last))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], c["out_channels"], 3, padding=1)

    # -- attention-processor protocol (the plugin seam)
    @property
    def attn_processors(self) -> Dict[str, object]:
        procs = {}

        def walk(name, module):
            if hasattr(module, "set_processor"):
                procs[f"{name}.processor"] = module.processor
            for sub, child in module.named_children():
                if sub == "processor":
                    continue
                walk(f"{name}.{sub}", child)

        for name, module in self.named_children():
            walk(name, module)
        return procs

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does "
                             f"not match the number of attention layers: {count}.")

        def walk(name, module):
            if hasattr(module, "set_processor"):
                module.set_processor(processor if not isinstance(processor, dict) else processor[f"{name}.processor"])
            for sub, child in module.named_children():
                if sub == "processor":
                    continue
                walk(f"{name}.{sub}", child)

        for name, module in self.named_children():
            walk(name, module)

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, return_dict=True):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        t_emb = timestep_embedding(timestep, self.config.block_out_channels[0]).to(sample.dtype)
        emb = self.time_embedding(t_emb)
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states, cross_attention_kwargs)
            skips.extend(outs)
        x = self.mid_block(x, emb, encoder_hidden_states, cross_attention_kwargs)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states, cross_attention_kwargs)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return UNetOutput(sample=x) if return_dict else (x,)


# --------------------------------------------------------------------------- A10 / A11
class DDPMSchedule:
    """scaled_linear betas, T=1000, epsilon prediction (A10)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    def add_noise(self, x0, noise, t):
        ac = self.alphas_cumprod.to(device=x0.device, dtype=x0.dtype)
        a = ac[t].sqrt().reshape(-1, 1, 1, 1)
        s = (1 - ac[t]).sqrt().reshape(-1, 1, 1, 1)
        return a * x0 + s * noise

    # DDIM, eta = 0, steps_offset = 1, set_alpha_to_one = False (A11)
    def ddim_timesteps(self, n):
        ratio = self.num_train_timesteps // n
        return [int(i * ratio) + 1 for i in reversed(range(n))]

    def ddim_step(self, eps, t, x, n):
        prev = t - self.num_train_timesteps // n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.alphas_cumprod[0]
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps


def init_unet_weights_(unet: nn.Module, seed: int = 0, gain: float = 1.0) -> None:
    """Seeded synthetic init at SD-1.5 shapes (no checkpoint is available offline).

    Variance-preserving fan-in init so activations stay O(1) through ~100 layers in fp16
    (SURVEY.md §7 hard part iv); zero biases; norm affine = (1, 0)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if ".processor." in name:
                continue
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (gain / math.sqrt(fan_in)))
            elif name.endswith("bias"):
                p.zero_()
