"""The object ``GraphcastModel.build_model()`` returns (/root/reference/skyrim/core/models/graphcast.py:51-54): a TimeLoop on the
HIP GraphCast engine.  GraphCast conditions on TWO time levels (the reference's state carries ``time=2``, graphcast.py:112-115), so
``n_history_levels = 2``: ``loop(time, x)`` takes x of shape (1, 2, 83, 721, 1440) = states at time - 6 h and time, and yields
(time, state (1, 83, 721, 1440), restart) starting with the input state at ``time``.  The forcings (solar-radiation proxy, day / year
progress) are a closed-form function of the valid time (spec.forcings)."""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass

import numpy as np
import torch

from .engine import GraphcastEngine
from .spec import CHANNELS, GraphcastConfig, forcings, init_synthetic, synthetic_states

_EPOCH = datetime.datetime(2000, 1, 1)


@dataclass
class Grid:
    lat: list
    lon: list

    @property
    def shape(self):
        return (len(self.lat), len(self.lon))


class GraphcastTimeLoop:
    n_history_levels = 2
    time_step = datetime.timedelta(hours=6)

    def __init__(self, params: dict | None = None, cfg: GraphcastConfig | None = None, device: str | torch.device = "cuda:0", seed: int = 0):
        """``params``: state dict keyed by ``spec.param_spec`` (default: ``SKYRIM_GRAPHCAST_WEIGHTS`` = a torch file of that dict, or
        seeded random parameters -- the e2mip://graphcast checkpoint is not obtainable in this environment)."""
        self.cfg = cfg or GraphcastConfig()
        self.engine = GraphcastEngine(self.cfg, device)
        if params is None:
            path = os.environ.get("SKYRIM_GRAPHCAST_WEIGHTS")
            params = torch.load(path, map_location="cpu") if path else init_synthetic(self.cfg, seed)
        self.engine.load_params(params)
        names = CHANNELS if self.cfg.n_vars == len(CHANNELS) else [f"c{i}" for i in range(self.cfg.n_vars)]
        self.in_channel_names = list(names)
        self.out_channel_names = list(names)
        self.grid = Grid(list(np.linspace(90.0, -90.0, self.cfg.n_lat)), list(np.arange(self.cfg.n_lon) * (360.0 / self.cfg.n_lon)))

    @property
    def device(self):
        return self.engine.device

    def to(self, device):
        if torch.device(device) != self.engine.device:
            raise NotImplementedError("the engine's buffers are bound to one GPU; build a new GraphcastTimeLoop for another device")
        return self

    def synthetic_state(self, seed: int) -> torch.Tensor:
        """Initial-condition hook of the synthetic DataSource (no network for GFS / ERA5 here)."""
        return synthetic_states(self.cfg, seed)[1]

    def _forcing(self, time: datetime.datetime) -> torch.Tensor:
        return forcings(self.cfg, (time - _EPOCH).total_seconds() / 3600.0).to(self.device)

    def __call__(self, time: datetime.datetime, x: torch.Tensor, restart=None):
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 2 or tuple(x.shape[2:]) != self.engine.state_shape:
            raise ValueError(f"expected x of shape (1, 2, {', '.join(map(str, self.engine.state_shape))}), got {tuple(x.shape)}")
        prev = x[0, 0].to(self.device, torch.float32).contiguous()
        cur = x[0, 1].to(self.device, torch.float32).contiguous()
        yield time, cur.unsqueeze(0).clone(), restart
        while True:
            nxt = self.engine.step(prev, cur, self._forcing(time))      # new buffer each step: the caller keeps the yielded one
            prev, cur = cur, nxt
            time = time + self.time_step
            yield time, cur.unsqueeze(0), restart
