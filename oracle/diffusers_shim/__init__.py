"""ORACLE tooling (test infrastructure only).

A minimal stand-in for the 7 ``diffusers`` symbols the reference ``models.py`` imports
(reference ``models.py:7-12``), so that the reference's OWN code (``ControlLoRA``, the three
attention processors, ``ConvBlock2D`` ...) can be imported *in place* from /root/reference and
executed on CPU in the build container.  It is used only by ``oracle/make_golden.py`` and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent, e.g. on the
GPU box).  Semantics follow SURVEY.md Appendix A1/A2/A9; real diffusers is not available,
so this boundary is "parity unpinned" (see oracle/unet_ref.py header).
"""
from __future__ import annotations

import functools
import inspect
import json
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import unet_ref


class BaseOutput:
    """dataclass-friendly output holder with tuple/dict style access."""

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(getattr(self, f) for f in self.__dataclass_fields__)


class FrozenDict(OrderedDict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for p in sig.parameters.values() if p.name != "self"]
        cfg = {p.name: p.default for p in params}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        init(self, *args, **{k: v for k, v in kwargs.items() if not k.startswith("_")})
        self._internal_dict = FrozenDict(cfg)
    return inner


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def load_config(cls, path_or_dict):
        if isinstance(path_or_dict, dict):
            return dict(path_or_dict)
        path = path_or_dict
        if os.path.isdir(path):
            path = os.path.join(path, cls.config_name)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = cls.load_config(config)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        init = {k: v for k, v in cfg.items() if k in accepted}
        init.update(kwargs)
        return cls(**init)

    def save_config(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        d = dict(self.config)
        d["_class_name"] = type(self).__name__
        d["_diffusers_version"] = "0.13.0.dev0"
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)


class ModelMixin(nn.Module):
    pass


class Mish(nn.Module):
    def forward(self, x):
        return x * torch.tanh(torch.nn.functional.softplus(x))


def _unsupported(*a, **k):
    raise NotImplementedError("dead branch under every shipped config (SURVEY.md C7)")


def install():
    """Register the fake ``diffusers`` package tree in sys.modules (idempotent)."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_clora_shim", False):
        return

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    d = mod("diffusers")
    d._clora_shim = True
    utils = mod("diffusers.utils")
    outputs = mod("diffusers.utils.outputs")
    outputs.BaseOutput = BaseOutput
    utils.outputs = outputs
    cu = mod("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    models = mod("diffusers.models")
    mu = mod("diffusers.models.modeling_utils")
    mu.ModelMixin = ModelMixin
    blocks = mod("diffusers.models.unet_2d_blocks")
    blocks.get_down_block = _unsupported
    resnet = mod("diffusers.models.resnet")
    resnet.Mish, resnet.Upsample2D, resnet.Downsample2D = Mish, unet_ref.Upsample2D, unet_ref.Downsample2D
    resnet.upsample_2d = resnet.downsample_2d = _unsupported
    resnet.partial = functools.partial
    ca = mod("diffusers.models.cross_attention")
    ca.CrossAttention, ca.LoRALinearLayer = unet_ref.CrossAttention, unet_ref.LoRALinearLayer
    d.utils, d.configuration_utils, d.models = utils, cu, models
    models.modeling_utils, models.unet_2d_blocks, models.resnet, models.cross_attention = mu, blocks, resnet, ca


def import_reference_models(reference_root="/root/reference"):
    """Import the reference ``models.py`` in place (never copied) under the shim."""
    install()
    import importlib.util
    spec = importlib.util.spec_from_file_location("clora_reference_models", os.path.join(reference_root, "models.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["clora_reference_models"] = m
    spec.loader.exec_module(m)
    return m
