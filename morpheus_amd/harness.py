"""Shared setup for bench.py, __graft_entry__.smoke() and the GPU tests: config, model with a
closed-form weight state, synthetic rays of one frame, renderer."""
from __future__ import annotations

import os
from typing import Optional

import torch
import yaml

from . import synth
from .model import scene_representation
from .render import HotPathRenderer, PresetSampler, UniformSampler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_config(name: str = "snoopy") -> dict:
    with open(os.path.join(ROOT, "configs", f"{name}.yaml")) as f:
        return yaml.safe_load(f)


def build_model(state_kind: str = "b", device="cuda", max_level=None, config: Optional[dict] = None,
                num_frames: int = 200) -> scene_representation:
    cfg = config or load_config()
    m = cfg["model"]
    model = scene_representation(cfg, 1.01, num_frames=num_frames, deform_dim=m["deform_dim"], use_app=m["use_app"],
                                 use_t=m["use_t"], amb_dim=m["amb_dim"], color_grid=m["color_grid"],
                                 use_joint=m["use_joint"], encode_topo=m["encode_topo"])
    model.load_state_dict(synth.make_state(state_kind, num_frames))
    model.max_level = max_level
    return model.to(device)


def make_renderer(model, n_samples: int, jitter=None, config: Optional[dict] = None, num_frames: int = 200,
                  samples=None) -> HotPathRenderer:
    cfg = config or model.config
    sampler = PresetSampler(*samples) if samples is not None else UniformSampler(n_samples, model.bound, jitter)
    return HotPathRenderer(model, cfg, sampler, num_frames)


def bench_loss(res, timg, tdep):
    """loss = MSE(image) + MSE(depth) against fixed targets: gradients reach every
    parameter group on the path, both hash tables included."""
    mse = torch.nn.functional.mse_loss           # the reference's own loss operator (morpheus.py:954, 980): one launch each way
    return mse(res["image"][0], timg) + mse(res["depth"][0], tdep)
