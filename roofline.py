"""Algorithmic work of the hot path, counted from layer shapes only (FLOPs = 2 x MACs), independent of how the
kernels are written.  SURVEY.md section 8(d) asks for this counter in-repo; bench.py divides these numbers by the
measured step time to report the whole-step MFMA fraction, and tests/test_roofline.py pins the totals
(UNet forward 803.3 GFLOP / image at 512^2, 180.1 at 256^2, hint encoder 25.3 / 24.1 GFLOP).

SD-1.5 topology as in the reference's call sites (train_text_to_image_control_lora.py:407-409) and its upstream
UNet2DConditionModel config: blocks (320, 640, 1280, 1280), 2 resnets per block, attention in the first three
down / last three up blocks and the mid block, 8 heads, text context 77 x 768, GEGLU feed-forward (x4).
"""
from __future__ import annotations

import json
import sys
from collections import OrderedDict

BLOCKS = (320, 640, 1280, 1280)
CTX_LEN, CTX_DIM, TEMB = 77, 768, 1280


class Counter(OrderedDict):
    def add(self, key, flops):
        self[key] = self.get(key, 0.0) + float(flops)

    @property
    def total(self):
        return sum(self.values())


def _conv(c, key, hw, cin, cout, k=3):
    c.add(key, 2.0 * hw * cin * cout * k * k)


def _linear(c, key, rows, cin, cout):
    c.add(key, 2.0 * rows * cin * cout)


def _resnet(c, hw, cin, cout):
    _conv(c, "conv3x3", hw, cin, cout)
    _linear(c, "time_emb", 1, TEMB, cout)
    _conv(c, "conv3x3", hw, cout, cout)
    if cin != cout:
        _conv(c, "shortcut_1x1", hw, cin, cout, 1)


def _transformer(c, hw, ch):
    _conv(c, "proj_1x1", hw, ch, ch, 1)                     # proj_in
    # attn1 (self)
    for _ in range(4):
        _linear(c, "qkvo", hw, ch, ch)
    c.add("scores_v", 2.0 * 2.0 * hw * hw * ch)             # QK^T and PV over all heads
    # attn2 (cross): q, out on hw rows; k, v on the 77 context rows
    _linear(c, "qkvo", hw, ch, ch)
    _linear(c, "qkvo", hw, ch, ch)
    _linear(c, "qkvo", CTX_LEN, CTX_DIM, ch)
    _linear(c, "qkvo", CTX_LEN, CTX_DIM, ch)
    c.add("scores_v", 2.0 * 2.0 * hw * CTX_LEN * ch)
    # GEGLU feed-forward
    _linear(c, "ff", hw, ch, 8 * ch)
    _linear(c, "ff", hw, 4 * ch, ch)
    _conv(c, "proj_1x1", hw, ch, ch, 1)                     # proj_out


def unet_forward_flops(res: int) -> Counter:
    """one image, one UNet forward"""
    c = Counter()
    L = res // 8
    hw = L * L
    _linear(c, "time_emb", 1, 320, TEMB)
    _linear(c, "time_emb", 1, TEMB, TEMB)
    _conv(c, "conv_in_out", hw, 4, 320)
    skips = [320]
    ch = 320
    for i, co in enumerate(BLOCKS):
        for _ in range(2):
            _resnet(c, hw, ch, co)
            ch = co
            if i < 3:
                _transformer(c, hw, ch)
            skips.append(ch)
        if i < 3:
            hw //= 4
            _conv(c, "updown_conv", hw, ch, ch)             # stride-2 conv, counted at the output resolution
            skips.append(ch)
    _resnet(c, hw, ch, ch)
    _transformer(c, hw, ch)
    _resnet(c, hw, ch, ch)
    for i, co in enumerate(reversed(BLOCKS)):
        for _ in range(3):
            _resnet(c, hw, ch + skips.pop(), co)
            ch = co
            if i > 0:
                _transformer(c, hw, ch)
        if i < 3:
            hw *= 4
            _conv(c, "updown_conv", hw, ch, ch)             # nearest x2 then conv at the upsampled resolution
    _conv(c, "conv_in_out", hw, 320, 4)
    return c


def hint_encoder_forward_flops(cfg: dict, res: int) -> Counter:
    """one image through ControlLoRA.conv_in / down_blocks / pre_lora_layers (reference models.py:669-700, 810-835).
    A ConvBlock2D holds ONE conv (models.py:529); stride-2 downsamplers are counted at their output resolution."""
    c = Counter()
    hw = res * res
    blocks = cfg["block_out_channels"]
    _conv(c, "hint_conv", hw, cfg.get("in_channels", 3), blocks[0])
    ch = blocks[0]
    for i, co in enumerate(blocks):                                   # down_blocks[0] = the 4 encoder stages
        for _ in range(cfg.get("layers_per_block", 1)):
            _conv(c, "hint_conv", hw, ch, co)
            ch = co
        if i < len(blocks) - 1:
            hw //= 4
            _conv(c, "hint_conv", hw, ch, ch)
    lora_in, lora_out = cfg["lora_block_in_channels"], cfg["lora_block_out_channels"]
    ctrl = cfg.get("lora_control_channels") or [None] * len(lora_in)
    for i in range(len(lora_in)):
        if i > 0:                                                     # down_blocks[i]: conv block(s) + downsample
            for _ in range(cfg.get("lora_pre_down_layers_per_block", 1)):
                _conv(c, "hint_conv", hw, ch, lora_in[i])
                ch = lora_in[i]
            hw //= 4
            _conv(c, "hint_conv", hw, ch, ch)
        if not cfg.get("lora_pre_conv_skipped", False):
            cin = ch
            for _ in range(cfg.get("lora_pre_conv_layers_per_block", 1)):
                cout = lora_out[i] if ctrl[i] is None else ctrl[i]
                _conv(c, "hint_pre_lora", hw, cin, cout, cfg.get("lora_pre_conv_layers_kernel_size", 1))
                cin = cout
    return c


def train_step_flops_per_image(res: int, cfg: dict, adapter_fwd_gflop: float) -> dict:
    """forward + backward with frozen UNet weights (dgrad only: 1x the forward GEMM/conv
    FLOPs, 2x the scores.V term which also needs dP and dS), 3x for the trainable hint encoder and adapters."""
    u = unet_forward_flops(res)
    h = hint_encoder_forward_flops(cfg, res)
    fwd = u.total
    bwd = u.total + u["scores_v"]
    return {"unet_fwd": fwd, "unet_bwd": bwd, "hint": 3.0 * h.total, "adapters": 3.0 * adapter_fwd_gflop * 1e9,
            "total": fwd + bwd + 3.0 * h.total + 3.0 * adapter_fwd_gflop * 1e9}


if __name__ == "__main__":
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    u = unet_forward_flops(res)
    print(json.dumps({"res": res, "unet_fwd_gflop": round(u.total / 1e9, 1),
                      "by_class_gflop": {k: round(v / 1e9, 1) for k, v in u.items()}}, indent=1))
