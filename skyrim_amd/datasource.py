"""Initial-condition sources with the earth2mip ``DataSource`` protocol the reference relies on
(``.channel_names``, ``.grid``, ``__getitem__(datetime) -> array (C, lat, lon)``;
/root/reference/skyrim/libs/ic/__init__.py:25-34, ifs.py:148-170).  The reference's fetchers download
GFS / IFS / ERA5 (network); here a seeded synthetic source and a file source (restart from a saved step)
provide the same protocol."""
from __future__ import annotations

import datetime
import hashlib

import numpy as np
import torch

from .pangu.spec import CHANNELS, PanguGeometry, synthetic_state

IC_SOURCES = ("cds", "gfs", "ifs", "synthetic", "file")


class SyntheticDataSource:
    """ERA5-magnitude random fields, deterministic in (time, seed)."""

    def __init__(self, channel_names=CHANNELS, geom: PanguGeometry | None = None, seed: int = 0, state_fn=None):
        """``state_fn(seed) -> (C, lat, lon) tensor`` in ``channel_names`` order: the model's own synthetic fields (models other
        than Pangu); default: Pangu's 69-channel generator on ``geom``."""
        self.channel_names = list(channel_names)
        self.geom = geom or PanguGeometry()
        self.seed = seed
        self.state_fn = state_fn

    @property
    def grid(self):
        return self.geom

    def __getitem__(self, time: datetime.datetime) -> np.ndarray:
        h = int(hashlib.sha256(f"{time.isoformat()}|{self.seed}".encode()).hexdigest()[:8], 16)
        if self.state_fn is not None:
            return self.state_fn(h).numpy()
        full = synthetic_state(self.geom, seed=h)
        idx = [CHANNELS.index(c) for c in self.channel_names]
        return full[idx].numpy()


class FileDataSource:
    """Reads the last time entry of a saved forecast file / store (restart, utils.py:24-27 of the reference)."""

    def __init__(self, path, channel_names=CHANNELS):
        self.path = path
        self.channel_names = list(channel_names)

    def __getitem__(self, time) -> np.ndarray:
        from .labeled import open_dataarray
        da = open_dataarray(self.path)
        return da.sel(channel=self.channel_names).values[-1]


def get_data_source(channel_names, initial_condition_source: str = "synthetic", geom: PanguGeometry | None = None, state_fn=None, **kw):
    """Mirror of skyrim.libs.ic.get_data_source.  The network sources of the reference (cds / gfs / ifs)
    are out of scope (no network): they resolve to the synthetic source of the same shape, and the name is
    kept so that file names / logs keep the reference's vocabulary."""
    if initial_condition_source not in IC_SOURCES:
        raise ValueError(f"Invalid initial condition source: {initial_condition_source}")
    if initial_condition_source == "file":
        return FileDataSource(kw["path"], channel_names)
    return SyntheticDataSource(channel_names, geom, state_fn=state_fn)


def get_initial_condition_for_model(model, data_source, time: datetime.datetime) -> torch.Tensor:
    """(B=1, n_history_levels, C, lat, lon) float32 on ``model.device`` -- earth2mip.initial_conditions'
    function of the same name as the reference calls it (utils.py:20)."""
    arrs = []
    for k in range(model.n_history_levels - 1, -1, -1):
        arrs.append(np.asarray(data_source[time - k * model.time_step], dtype=np.float32))
    x = torch.from_numpy(np.stack(arrs))
    return x.to(model.device).unsqueeze(0)
