"""The in-repo algorithmic work counter (roofline.py) against the figures SURVEY.md section 8(d) fixes."""
import json
import os

import pytest

import roofline as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(name):
    return json.load(open(os.path.join(ROOT, "configs", name + ".json")))


def test_unet_forward_flops():
    u512, u256 = R.unet_forward_flops(512), R.unet_forward_flops(256)
    assert round(u512.total / 1e9, 1) == 803.3 and round(u256.total / 1e9, 1) == 180.1
    by = {k: round(v / 1e9, 1) for k, v in u512.items()}
    assert by["ff"] == 153.5 and by["scores_v"] == 126.1 and by["qkvo"] == 79.7 and by["updown_conv"] == 73.6
    assert by["proj_1x1"] == 25.6 and by["shortcut_1x1"] == 18.0
    assert abs(by["conv3x3"] + by["conv_in_out"] - 326.7) < 0.11


@pytest.mark.parametrize("name,res,gflop", [("fill50k", 512, 25.3), ("mpii-pose-v2", 512, 24.1), ("fill50k", 256, 6.33)])
def test_hint_encoder_flops(name, res, gflop):
    assert abs(R.hint_encoder_forward_flops(_cfg(name), res).total / 1e9 - gflop) < 0.06


def test_train_step_total():
    t = R.train_step_flops_per_image(512, _cfg("fill50k"), 1.51)
    assert abs(t["total"] / 1e12 - 1.81) < 0.01 and abs(t["unet_bwd"] / 1e9 - 929.4) < 0.2
    t = R.train_step_flops_per_image(256, _cfg("fill50k"), 0.40)
    assert abs(t["total"] / 1e12 - 0.39) < 0.005
