"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU / fp32 restatement of the reference-owned half of the hot path:

* attention processors   -- reference ``models.py:72-152`` (P1), ``:155-287`` (P2/P3), ``:292-431`` (P4)
* hint encoder           -- reference ``models.py:434-547`` (H1), ``:550-610`` (H2), ``:618-808`` (H3), ``:810-835`` (H4)
* processor <-> UNet map -- reference ``train_text_to_image_control_lora.py:469-487`` (M1)

Written independently (chain-of-adapters formulation) with the SAME parameter names so the
state-dict keys match the reference.  Pinned against the reference's own code
executed in place under ``oracle/diffusers_shim`` by ``tests/test_oracle_vs_reference.py`` and
frozen into ``tests/golden/*.safetensors`` by ``oracle/make_golden.py``.  Quirks C1-C9 of
SURVEY.md Appendix C are preserved on purpose.
"""
from __future__ import annotations

import inspect
import json
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import Downsample2D, LoRALinearLayer


# ------------------------------------------------------------------ processors (P1-P4)
class LoRAProcRef(nn.Module):
    """P1 -- reference models.py:72-152."""

    version = 0

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, post_add=False,
                 key_states_skipped=False, value_states_skipped=False, output_states_skipped=False):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.rank, self.post_add = hidden_size, cross_attention_dim, rank, post_add
        kv_in = hidden_size if post_add else (cross_attention_dim or hidden_size)
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank)
        if not key_states_skipped:
            self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank)
        if not value_states_skipped:
            self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank)
        if not output_states_skipped:
            self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank)
        self.key_states_skipped, self.value_states_skipped, self.output_states_skipped = (
            key_states_skipped, value_states_skipped, output_states_skipped)
        self.pre_loras: List["LoRAProcRef"] = []
        self.post_loras: List["LoRAProcRef"] = []

    # -- toggles (models.py:103-116; C4: the value toggle asserts on to_q_lora)
    def skip_key_states(self, is_skipped=True):
        if not is_skipped:
            assert hasattr(self, "to_k_lora")
        self.key_states_skipped = is_skipped

    def skip_value_states(self, is_skipped=True):
        if not is_skipped:
            assert hasattr(self, "to_q_lora")
        self.value_states_skipped = is_skipped

    def skip_output_states(self, is_skipped=True):
        if not is_skipped:
            assert hasattr(self, "to_out_lora")
        self.output_states_skipped = is_skipped

    def _chain(self):
        return [*self.pre_loras, self, *self.post_loras]

    # hooks specialised by the control variants
    def _q_extra(self, owner, h, scale):
        return 0

    def _pre(self, h, scale):
        return h

    def _post_attn(self, a, scale):
        return a

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0):
        if self.version:
            assert self.control_states is not None          # models.py:227 / :362
        h = self._pre(hidden_states, scale)
        chain = self._chain()
        q = attn.to_q(h)
        for p in chain:
            src = (q if p.post_add else h) + self._q_extra(p, h, scale)
            q = q + scale * p.to_q_lora(src)
        e = h if encoder_hidden_states is None else encoder_hidden_states
        k = attn.to_k(e)
        for p in chain:
            if not p.key_states_skipped:
                k = k + scale * p.to_k_lora(k if p.post_add else e)
        v = attn.to_v(e)
        for p in chain:
            if not p.value_states_skipped:
                # C3: borrowed (pre/post) value adapters are NOT multiplied by scale
                v = v + (scale if (p is self or self.version == 0) else 1.0) * p.to_v_lora(v if p.post_add else e)
        probs = attn.get_attention_scores(attn.head_to_batch_dim(q), attn.head_to_batch_dim(k), None)
        a = attn.batch_to_head_dim(torch.bmm(probs, attn.head_to_batch_dim(v)))
        a = self._post_attn(a, scale)
        out = attn.to_out[0](a)
        for p in chain:
            # C2: a control processor applies its OWN out adapter unconditionally
            if (p is self and self.version) or not p.output_states_skipped:
                out = out + scale * p.to_out_lora(out if p.post_add else a)
        return attn.to_out[1](out)


class _ControlMixin:
    def inject_pre_lora(self, lora_layer):
        self.pre_loras.append(lora_layer)

    def inject_post_lora(self, lora_layer):
        self.post_loras.append(lora_layer)

    def inject_control_states(self, control_states):
        self.control_states = control_states

    def process_control_states(self, hidden_states, scale=1.0, is_out=False):
        """P2 -- models.py:201-220 / :336-355 (flatten-and-cache quirk C5, repeat order C6)."""
        ctrl = self.control_states.to(hidden_states.dtype)
        if hidden_states.ndim == 3 and ctrl.ndim == 4:
            b, _, hh, ww = ctrl.shape
            ctrl = ctrl.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
            self.control_states = ctrl
        x = ctrl
        if self.concat_hidden:
            b1, b2 = ctrl.shape[0], hidden_states.shape[0]
            if b1 != b2:
                ctrl = ctrl.repeat_interleave(b2 // b1, dim=0)
            x = torch.cat([hidden_states, ctrl], dim=-1)
        layer = self.to_control_out if is_out else self.to_control
        y = scale * layer(x)
        return ctrl + y if self.control_self_add else y


class ControlLoRAProcRef(_ControlMixin, LoRAProcRef):
    """P3 -- reference models.py:155-287."""

    version = 1

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, control_rank=None, post_add=False,
                 concat_hidden=False, control_channels=None, control_self_add=True, key_states_skipped=False,
                 value_states_skipped=False, output_states_skipped=False, **kwargs):
        super().__init__(hidden_size, cross_attention_dim, rank, post_add, key_states_skipped,
                         value_states_skipped, output_states_skipped)
        control_rank = rank if control_rank is None else control_rank
        control_channels = hidden_size if control_channels is None else control_channels
        self.concat_hidden = concat_hidden
        self.control_self_add = False          # C1: the reference's conditional always yields False
        self.control_states = None
        self.to_control = LoRALinearLayer(control_channels + (hidden_size if concat_hidden else 0), hidden_size, control_rank)

    def _q_extra(self, owner, h, scale):
        if isinstance(owner, ControlLoRAProcRef):
            return owner.process_control_states(h, scale)
        return 0


class ControlLoRAProcV2Ref(_ControlMixin, LoRAProcRef):
    """P4 -- reference models.py:292-431."""

    version = 2

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, control_rank=None, control_channels=None, **kwargs):
        super().__init__(hidden_size, cross_attention_dim, rank, False, True, True, False)
        control_rank = rank if control_rank is None else control_rank
        control_channels = hidden_size if control_channels is None else control_channels
        self.concat_hidden, self.control_self_add, self.control_states = True, False, None
        self.to_control = LoRALinearLayer(hidden_size + control_channels, hidden_size, control_rank)
        self.to_control_out = LoRALinearLayer(hidden_size + control_channels, hidden_size, control_rank)

    def _pre(self, h, scale):
        for p in self._chain():
            if isinstance(p, ControlLoRAProcV2Ref):
                h = h + p.process_control_states(h, scale)
        return h

    def _post_attn(self, a, scale):
        for p in self._chain():
            if isinstance(p, ControlLoRAProcV2Ref):
                a = a + p.process_control_states(a, scale, is_out=True)
        return a


# ------------------------------------------------------------------ hint encoder (H1-H4)
class HintConvBlockRef(nn.Module):
    """H1 -- non-residual GN -> SiLU -> conv_k -> GN -> SiLU (models.py:512-547 with temb=None)."""

    def __init__(self, cin, cout, k, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, k, padding=k // 2)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)

    def forward(self, x):
        return F.silu(self.norm2(self.conv1(F.silu(self.norm1(x)))))


class HintStageRef(nn.Module):
    """H2 -- models.py:550-610 (downsample pad (0,1,0,1) because padding=0, A9)."""

    def __init__(self, cin, cout, layers, k, groups, downsample):
        super().__init__()
        self.convnets = nn.ModuleList([HintConvBlockRef(cin if i == 0 else cout, cout, k, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, True, cout, 0, "op")]) if downsample else None

    def forward(self, x):
        for c in self.convnets:
            x = c(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


@dataclass
class ControlLoRAOutputRef:
    control_states: Tuple[torch.Tensor, ...]


DEFAULT_CROSS_DIMS = ([None, 768] * 5, [None, 768] * 5, [None, 768] * 5, [None, 768])


class ControlLoRARef(nn.Module):
    """H3/H4 -- reference models.py:618-835."""

    def __init__(self, in_channels=3, down_block_types=("SimpleDownEncoderBlock2D",) * 4,
                 block_out_channels=(32, 64, 128, 256), layers_per_block=1, act_fn="silu", norm_num_groups=32,
                 lora_pre_down_block_types=(None,) + ("SimpleDownEncoderBlock2D",) * 3,
                 lora_pre_down_layers_per_block=1, lora_pre_conv_skipped=False,
                 lora_pre_conv_types=("SimpleDownEncoderBlock2D",) * 4, lora_pre_conv_layers_per_block=1,
                 lora_pre_conv_layers_kernel_size=1, lora_block_in_channels=(256, 256, 256, 256),
                 lora_block_out_channels=(320, 640, 1280, 1280), lora_cross_attention_dims=DEFAULT_CROSS_DIMS,
                 lora_rank=4, lora_control_rank=None, lora_post_add=False, lora_concat_hidden=False,
                 lora_control_channels=(None, None, None, None), lora_control_self_add=True,
                 lora_key_states_skipped=False, lora_value_states_skipped=False, lora_output_states_skipped=False,
                 lora_control_version=1):
        super().__init__()
        self.config = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        cls = ControlLoRAProcV2Ref if lora_control_version == 2 else ControlLoRAProcRef
        assert lora_block_in_channels[0] == block_out_channels[-1]            # models.py:674
        if lora_pre_conv_skipped:                                              # models.py:676-678 (C13)
            lora_control_channels = lora_block_in_channels
            lora_control_self_add = False
        g = norm_num_groups
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.down_blocks, self.pre_lora_layers, self.lora_layers = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        stages, c = [], block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            stages.append(HintStageRef(c, co, layers_per_block, 3, g, i != len(block_out_channels) - 1))
            c = co
        n_ids = len(lora_pre_down_block_types)
        for i in range(n_ids):
            if i == 0:
                self.down_blocks.append(nn.Sequential(*stages))
                cin = lora_block_in_channels[0]
            else:
                cin_prev, cin = lora_block_in_channels[i - 1], lora_block_in_channels[i]
                self.down_blocks.append(HintStageRef(cin_prev, cin, lora_pre_down_layers_per_block, 3, g, True))
            cc = lora_control_channels[i]
            if lora_pre_conv_skipped:
                self.pre_lora_layers.append(nn.Identity())
            else:
                self.pre_lora_layers.append(HintStageRef(
                    cin, lora_block_out_channels[i] if cc is None else cc,
                    lora_pre_conv_layers_per_block, lora_pre_conv_layers_kernel_size, g, False))
            self.lora_layers.append(nn.ModuleList([
                cls(lora_block_out_channels[i], cross_attention_dim=cad, rank=lora_rank, control_rank=lora_control_rank,
                    post_add=lora_post_add, concat_hidden=lora_concat_hidden, control_channels=cc,
                    control_self_add=lora_control_self_add, key_states_skipped=lora_key_states_skipped,
                    value_states_skipped=lora_value_states_skipped, output_states_skipped=lora_output_states_skipped)
                for cad in lora_cross_attention_dims[i]]))

    @classmethod
    def from_config(cls, config, **kwargs):
        if not isinstance(config, dict):
            path = os.path.join(config, "config.json") if os.path.isdir(config) else config
            with open(path) as f:
                config = json.load(f)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        init = {k: v for k, v in config.items() if k in accepted}
        init.update(kwargs)
        return cls(**init)

    def forward(self, x, return_dict=True):
        orig_dtype = x.dtype
        h = self.conv_in(x.to(self.conv_in.weight.dtype))
        outs = []
        for down, pre, procs in zip(self.down_blocks, self.pre_lora_layers, self.lora_layers):
            h = down(h)
            ctrl = pre(h).to(orig_dtype)                                       # C9
            for p in procs:
                p.inject_control_states(ctrl)
            outs.append(ctrl)
        return ControlLoRAOutputRef(tuple(outs)) if return_dict else tuple(outs)


# ------------------------------------------------------------------ mapping (M1)
def map_processors_to_unet(unet, control_lora) -> dict:
    """M1 -- train_text_to_image_control_lora.py:469-487 (also apps/gradio_canny2image.py:43-63)."""
    n = len(unet.config.block_out_channels)
    pools = [list(l) for l in control_lora.lora_layers]
    procs = {}
    for name in unet.attn_processors.keys():
        if name.startswith("mid_block"):
            cid = n - 1
        elif name.startswith("up_blocks"):
            cid = n - 1 - int(name[len("up_blocks.")])
        else:
            cid = int(name[len("down_blocks.")])
        if pools[cid]:
            procs[name] = pools[cid].pop(0)
    return procs


def randomize_adapters_(control_lora: nn.Module, seed=1, std=0.02) -> None:
    """Give every adapter ``up`` a non-zero value: zero-init would hide adapter bugs (SURVEY §8c pin 7)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in control_lora.named_parameters():
            if name.endswith(".up.weight"):
                p.copy_(torch.randn(p.shape, generator=g) * std)
