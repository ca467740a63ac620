"""The object ``GraphcastModel.build_model()`` returns (/root/reference/skyrim/core/models/graphcast.py:51-54): a TimeLoop on the
HIP GraphCast engine, with the ``stepper`` member the reference drives it through (graphcast.py:102-118).

GraphCast conditions on TWO time levels (the reference's state carries ``time=2``, graphcast.py:112-115), so
``n_history_levels = 2``: ``loop(time, x)`` takes x of shape (1, 2, 83, 721, 1440) = states at time - 6 h and time, and yields
(time, state (1, 83, 721, 1440), restart) starting with the input state at ``time``.

``loop.stepper.initialize(x, time) -> state`` and ``loop.stepper.step(state) -> (state, output)`` reproduce earth2mip's stepper
contract: ``state = (time, dataset, rng)`` where ``dataset`` holds DeepMind's variable names over (batch, time = 2, [level,] lat, lon)
with latitude ASCENDING (-90 .. 90; the reference flips it back in ``forecast``, graphcast.py:138).  Here the two time levels stay
in HBM between steps; the dataset view is materialised on the host only when it is read.

The forcings (solar-radiation proxy, day / year progress) are a closed-form function of the valid time, evaluated on the device
from cached latitude / longitude trigonometry planes (no per-step host work, no H2D copy)."""
from __future__ import annotations

import datetime
import math
from dataclasses import dataclass

import numpy as np
import torch

from .. import weights
from ..labeled import DataArray, Dataset
from .engine import GraphcastEngine
from .spec import CHANNELS, N_FORCING, GraphcastConfig, init_synthetic, synthetic_states

_EPOCH = datetime.datetime(2000, 1, 1)
_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
# DeepMind variable name -> (channel code, has levels); the engine's channel order is CHANNELS (graphcast.py:17-26)
DATASET_VARS = [("geopotential", "z", True), ("specific_humidity", "q", True), ("temperature", "t", True),
                ("u_component_of_wind", "u", True), ("v_component_of_wind", "v", True), ("vertical_velocity", "w", True),
                ("10m_u_component_of_wind", "u10m", False), ("10m_v_component_of_wind", "v10m", False),
                ("2m_temperature", "t2m", False), ("mean_sea_level_pressure", "msl", False),
                ("toa_incident_solar_radiation", "tp06", False)]


@dataclass
class Grid:
    lat: list
    lon: list

    @property
    def shape(self):
        return (len(self.lat), len(self.lon))


class StateDataset(Dataset):
    """The stepper state's dataset: two time levels resident on the GPU, DeepMind-named host variables built on first read."""

    def __init__(self, loop: "GraphcastTimeLoop", prev: torch.Tensor, cur: torch.Tensor, times):
        self._loop, self.device_state, self._times = loop, (prev, cur), list(times)
        self._vars = None

    @property
    def data_vars(self):
        if self._vars is None:
            host = torch.stack(self.device_state).cpu().numpy()                        # (2, C, lat 90..-90, lon)
            self._vars = self._loop.dataset_vars(host, self._times)
        return self._vars


class _Stepper:
    """earth2mip's stepper protocol (time_loop.py:114-122 at the commit the reference links, graphcast.py:101)."""

    def __init__(self, loop: "GraphcastTimeLoop"):
        self.loop = loop

    def initialize(self, x: torch.Tensor, time: datetime.datetime):
        loop = self.loop
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != 2 or tuple(x.shape[2:]) != loop.engine.state_shape:
            raise ValueError(f"expected x of shape (1, 2, {', '.join(map(str, loop.engine.state_shape))}), got {tuple(x.shape)}")
        prev = x[0, 0].to(loop.device, torch.float32).contiguous()
        cur = x[0, 1].to(loop.device, torch.float32).contiguous()
        return (time, StateDataset(loop, prev, cur, [time - loop.time_step, time]), np.zeros(2, dtype=np.uint32))

    def step(self, state):
        loop = self.loop
        time, ds, rng = state
        prev, cur = ds.device_state
        nxt = loop.engine.step(prev, cur, loop.forcing(time))          # new buffer each step: the caller keeps the returned one
        time = time + loop.time_step
        return (time, StateDataset(loop, cur, nxt, [time - loop.time_step, time]), rng), nxt.unsqueeze(0)


class GraphcastTimeLoop:
    n_history_levels = 2
    time_step = datetime.timedelta(hours=6)

    def __init__(self, params: dict | None = None, cfg: GraphcastConfig | None = None, device: str | torch.device = "cuda:0", seed: int = 0):
        """``params``: state dict keyed by ``spec.param_spec``; default: the torch file named by ``SKYRIM_GRAPHCAST_WEIGHTS``
        (``skyrim_amd.weights``: seeded random parameters only with SKYRIM_SYNTHETIC_WEIGHTS=1 -- the e2mip://graphcast checkpoint
        is not obtainable in this environment)."""
        self.cfg = cfg or GraphcastConfig()
        self.engine = GraphcastEngine(self.cfg, device)
        if params is None:
            params = weights.resolve("SKYRIM_GRAPHCAST_WEIGHTS", self._load, lambda: init_synthetic(self.cfg, seed), "graphcast")
        self.engine.load_params(params)
        names = CHANNELS if self.cfg.n_vars == len(CHANNELS) else [f"c{i}" for i in range(self.cfg.n_vars)]
        self.in_channel_names = list(names)
        self.out_channel_names = list(names)
        self.grid = Grid(list(np.linspace(90.0, -90.0, self.cfg.n_lat)), list(np.arange(self.cfg.n_lon) * (360.0 / self.cfg.n_lon)))
        self.stepper = _Stepper(self)
        # latitude / longitude planes of the insolation proxy, cached on the device (float64: the proxy is compared with spec.forcings)
        lat = torch.deg2rad(torch.linspace(90.0, -90.0, self.cfg.n_lat, dtype=torch.float64, device=self.device))[:, None]
        lon = torch.deg2rad(torch.arange(self.cfg.n_lon, dtype=torch.float64, device=self.device) * (360.0 / self.cfg.n_lon))[None, :]
        self._sin_lat, self._cos_lat, self._lon = torch.sin(lat), torch.cos(lat), lon

    def _load(self, path: str) -> dict:
        """A torch file of the slot dict (``spec.param_spec``), or a directory in the shape of deepmind's release: ``params.npz`` -- the
        haiku parameters as ``checkpoint.dump`` flattens them (``params:<module path>:<name>``; '/'-joined keys are accepted too) -- and
        ``stats.npz`` with ``mean``, ``std``, ``diff_std`` (per variable, the engine's channel order) and ``static`` (2, n_lat, n_lon:
        normalised surface geopotential and land-sea mask), optionally ``in_perm`` / ``out_perm``; mapped by ``checkpoint.load``."""
        import os
        if os.path.isdir(path):
            from . import checkpoint
            st = np.load(os.path.join(path, "stats.npz"))
            kw = {k: st[k] for k in ("mean", "std", "diff_std", "static", "in_perm", "out_perm") if k in st.files}
            return checkpoint.load(os.path.join(path, "params.npz"), self.cfg, **kw)
        return torch.load(path, map_location="cpu")

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

    def release(self):
        """Drop the engine's graph, weights and latents and the cached forcing planes (GlobalModel.release_model)."""
        self.engine.release()
        self.stepper = None
        self._sin_lat = self._cos_lat = self._lon = None

    def forcing(self, time: datetime.datetime) -> torch.Tensor:
        """(15, n_lat, n_lon) forcings of the step from ``time`` -- the same closed forms as ``spec.forcings``, on the device."""
        hours = (time - _EPOCH).total_seconds() / 3600.0
        out = torch.empty((N_FORCING, self.cfg.n_lat, self.cfg.n_lon), dtype=torch.float32, device=self.device)
        for j, dt in enumerate((-6.0, 0.0, 6.0)):
            h = hours + dt
            day, year = (h / 24.0) % 1.0, (h / (24.0 * 365.25)) % 1.0
            decl = -0.409 * math.cos(2 * math.pi * (year + 10.0 / 365.25))
            cosz = self._sin_lat * math.sin(decl) + self._cos_lat * math.cos(decl) * torch.cos(2 * math.pi * day + self._lon - math.pi)
            out[5 * j] = cosz.clamp_min_(0.0)
            for i, v in enumerate((math.sin(2 * math.pi * day), math.cos(2 * math.pi * day), math.sin(2 * math.pi * year), math.cos(2 * math.pi * year))):
                out[5 * j + 1 + i].fill_(v)
        return out

    def dataset_vars(self, host: np.ndarray, times) -> dict:
        """(2, C, lat descending, lon) host array in CHANNELS order -> DeepMind-named variables with (batch, time, [level,] lat, lon)
        dims and ASCENDING latitude, as views (no copy)."""
        if self.cfg.n_vars != len(CHANNELS):
            raise NotImplementedError("the dataset view needs the operational 83-channel configuration")
        lat, lon = np.asarray(self.grid.lat)[::-1], np.asarray(self.grid.lon)
        tcoord = np.array(times, dtype="datetime64[ns]")
        out, c = {}, 0
        for name, _, levelled in DATASET_VARS:
            n = len(_LEVELS) if levelled else 1
            block = host[:, c:c + n, ::-1, :]
            c += n
            if levelled:
                out[name] = DataArray(block[None], ["batch", "time", "level", "lat", "lon"],
                                      dict(time=tcoord, level=np.array(_LEVELS), lat=lat, lon=lon))
            else:
                out[name] = DataArray(block[None, :, 0], ["batch", "time", "lat", "lon"], dict(time=tcoord, lat=lat, lon=lon))
        return out

    def __call__(self, time: datetime.datetime, x: torch.Tensor, restart=None):
        state = self.stepper.initialize(x, time)
        yield time, state[1].device_state[1].unsqueeze(0).clone(), restart
        while True:
            state, output = self.stepper.step(state)
            yield state[0], output, restart
