"""The object ``FourcastnetV2Model.build_model()`` returns: earth2mip's TimeLoop protocol
(/root/reference/skyrim/core/models/fourcastnet_v2.py:24-28, consumed by models/utils.py:10-49) on the HIP SFNO engine.

    loop(time, x) -> iterator of (time, state (B=1, 73, 721, 1440) on .device, restart);  first yield = the input state.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass

import numpy as np
import torch

from .. import weights
from .engine import SfnoEngine
from .spec import CHANNELS, SfnoConfig, init_synthetic, synthetic_state


@dataclass
class Grid:
    lat: list
    lon: list

    @property
    def shape(self):
        return (len(self.lat), len(self.lon))


class SfnoTimeLoop:
    n_history_levels = 1
    time_step = datetime.timedelta(hours=6)

    def __init__(self, params: dict | None = None, cfg: SfnoConfig | None = None, device: str | torch.device = "cuda:0", seed: int = 0):
        """``params``: state dict keyed by ``spec.param_spec`` (default: ``SKYRIM_SFNO_WEIGHTS`` = a torch file of that dict,
        or seeded random parameters -- the e2mip://fcnv2_sm checkpoint is not obtainable in this environment)."""
        self.cfg = cfg or SfnoConfig()
        self.engine = SfnoEngine(self.cfg, device)
        if params is None:
            params = weights.resolve("SKYRIM_SFNO_WEIGHTS", self._load, lambda: init_synthetic(self.cfg, seed), "fourcastnet_v2")
        self.engine.load_params(params)
        names = CHANNELS if self.cfg.in_chans == len(CHANNELS) else [f"c{i}" for i in range(self.cfg.in_chans)]
        self.in_channel_names = list(names)
        self.out_channel_names = list(names[: self.cfg.out_chans])
        self.grid = Grid(list(np.linspace(90.0, -90.0, self.cfg.n_lat)), list(np.arange(self.cfg.n_lon) * (360.0 / self.cfg.n_lon)))

    def _load(self, path: str) -> dict:
        """A torch file of the slot dict (``spec.param_spec``), or the reference's own package: a directory holding ``weights.tar``,
        ``global_means.npy`` and ``global_stds.npy`` (earth2mip's fcnv2_sm layout), mapped by ``checkpoint.convert``."""
        if os.path.isdir(path):
            from . import checkpoint
            return checkpoint.load(os.path.join(path, "weights.tar"), self.cfg, os.path.join(path, "global_means.npy"), os.path.join(path, "global_stds.npy"))
        return torch.load(path, map_location="cpu")

    @property
    def device(self):
        return self.engine.device

    def to(self, device):
        if torch.device(device) != self.engine.device:
            raise NotImplementedError("the engine's buffers are bound to one GPU; build a new SfnoTimeLoop for another device")
        return self

    def synthetic_state(self, seed: int) -> torch.Tensor:
        """Initial-condition hook of the synthetic DataSource (no network for GFS / ERA5 here)."""
        return synthetic_state(self.cfg, seed)

    def release(self):
        """Drop every prepared matrix and work buffer of the engine (GlobalModel.release_model; the SFNO C ABI is stateless)."""
        self.engine.release()

    def __call__(self, time: datetime.datetime, x: torch.Tensor, restart=None):
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 1 or tuple(x.shape[2:]) != self.engine.state_shape:
            raise ValueError(f"expected x of shape (1, 1, {', '.join(map(str, self.engine.state_shape))}), got {tuple(x.shape)}")
        state = x[0, 0].to(self.device, torch.float32).contiguous()
        yield time, state.unsqueeze(0).clone(), restart
        while True:
            state = self.engine.step(state)                  # new buffer each step: the caller keeps the yielded one
            time = time + self.time_step
            yield time, state.unsqueeze(0), restart
