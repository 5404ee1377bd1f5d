"""The ``configs/*.json`` schema of the reference is a drop-in contract (SURVEY.md section 5 "Config"): each
file is the kwargs of ``ControlLoRA.__init__`` in diffusers' config format.  The files under ``configs/`` are
GENERATED from the constructor defaults plus the per-config deltas below (``python -m controllora_amd.configs``)."""
from __future__ import annotations

import json
import os

_CROSS = [[None, 768] * 5, [None, 768] * 5, [None, 768] * 5, [None, 768]]
_SDB = ["SimpleDownEncoderBlock2D"] * 4

BASE = {
    "_class_name": "ControlLoRA", "_diffusers_version": "0.13.0.dev0", "act_fn": "silu",
    "block_out_channels": [32, 64, 128, 256], "down_block_types": _SDB, "in_channels": 3, "layers_per_block": 1,
    "lora_block_in_channels": [256, 256, 256, 256], "lora_block_out_channels": [320, 640, 1280, 1280],
    "lora_control_rank": None, "lora_cross_attention_dims": _CROSS, "lora_post_add": False,
    "lora_pre_conv_layers_kernel_size": 1, "lora_pre_conv_layers_per_block": 1, "lora_pre_conv_types": _SDB,
    "lora_pre_down_block_types": [None] + _SDB[1:], "lora_pre_down_layers_per_block": 1, "lora_rank": 4,
    "norm_num_groups": 32,
}
_V2 = {"lora_concat_hidden": True, "lora_control_channels": [256, 256, 256], "lora_control_self_add": False,
       "lora_control_version": 2, "lora_key_states_skipped": True, "lora_output_states_skipped": False,
       "lora_pre_conv_skipped": True, "lora_value_states_skipped": True}
_SKETCH = {"lora_control_channels": [256, 256, 256], "lora_control_rank": 256, "lora_control_self_add": False,
           "lora_concat_hidden": True, "lora_pre_conv_skipped": True}

CONFIGS = {
    "base": {}, "fill50k": {}, "diffusiondb-canny": {}, "mpii-pose": {}, "post-add": {"lora_post_add": True},
    "danbooru-sketch": _SKETCH, "mpii-pose-v2": _V2, "diffusiondb-canny-v2": _V2,
}


def config_dict(name: str) -> dict:
    d = dict(BASE)
    d.update(CONFIGS[name])
    return d


def write_configs(out_dir: str) -> None:
    os.makedirs(out_dir, exist_ok=True)
    for name in CONFIGS:
        with open(os.path.join(out_dir, f"{name}.json"), "w") as f:
            json.dump(config_dict(name), f, indent=2, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    write_configs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs"))
