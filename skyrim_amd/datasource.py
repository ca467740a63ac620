"""Initial-condition sources with the earth2mip ``DataSource`` protocol the reference relies on
(``.channel_names``, ``.grid``, ``__getitem__(datetime) -> array (C, lat, lon)``;
/root/reference/skyrim/libs/ic/__init__.py:25-34, ifs.py:148-170).  The reference's fetchers download
GFS / IFS / ERA5 (network); here a seeded synthetic source and a file source (restart from a saved step)
provide the same protocol."""
from __future__ import annotations

import datetime
import hashlib
import logging
import os
from pathlib import Path

import numpy as np
import torch

from .pangu.spec import CHANNELS, PanguGeometry, synthetic_state

IC_SOURCES = ("cds", "gfs", "ifs", "synthetic", "file")
NETWORK_SOURCES = ("cds", "gfs", "ifs")
logger = logging.getLogger("skyrim_amd")


class SyntheticDataSource:
    """ERA5-magnitude random fields, deterministic in (time, seed).  ``label`` is what saved files are stamped with."""
    label = "synthetic"

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


class ArchiveDataSource:
    """Local mirror of a network source: ``{root}/{source}/{%Y%m%d_%H%M}.nc`` (or a zarr store of that stem), one state per
    file in the saved-forecast layout -- the on-disk form the reference's fetchers cache their downloads in."""

    def __init__(self, root, source: str, channel_names=CHANNELS):
        self.root, self.source = Path(root) / source, source
        self.channel_names = list(channel_names)

    def __getitem__(self, time: datetime.datetime) -> np.ndarray:
        from .labeled import open_dataarray
        stem = self.root / time.strftime("%Y%m%d_%H%M")
        for cand in (stem.with_suffix(".nc"), stem.with_suffix(".zarr"), stem):
            if cand.exists():
                return open_dataarray(cand).sel(channel=self.channel_names).values[-1]
        raise FileNotFoundError(f"no {self.source} initial condition for {time:%Y-%m-%d %H:%M} under {self.root}")


def get_data_source(channel_names, initial_condition_source: str = "synthetic", geom: PanguGeometry | None = None, state_fn=None, **kw):
    """Mirror of skyrim.libs.ic.get_data_source (/root/reference/skyrim/libs/ic/__init__.py:25-34).  The reference's cds / gfs /
    ifs fetchers download from network services, which this build does not ship: those names resolve to a local archive when
    ``SKYRIM_IC_DIR`` points at one, to the seeded synthetic fields ONLY when ``SKYRIM_SYNTHETIC_IC=1`` opts in (files are then
    stamped "synthetic", never with the requested source's name), and raise otherwise."""
    if initial_condition_source not in IC_SOURCES:
        raise ValueError(f"Invalid initial condition source: {initial_condition_source}")
    if initial_condition_source == "file":
        return FileDataSource(kw["path"], channel_names)
    if initial_condition_source in NETWORK_SOURCES:
        root = os.environ.get("SKYRIM_IC_DIR")
        if root:
            return ArchiveDataSource(root, initial_condition_source, channel_names)
        if os.environ.get("SKYRIM_SYNTHETIC_IC") != "1":
            raise RuntimeError(
                f"ic_source={initial_condition_source!r} is a network fetcher of the reference and is not part of this build: point "
                "SKYRIM_IC_DIR at a local archive ({root}/{source}/YYYYMMDD_HHMM.nc), use ic_source='file' / 'synthetic', or set "
                "SKYRIM_SYNTHETIC_IC=1 to run on seeded synthetic fields")
        logger.warning(f"ic_source={initial_condition_source!r}: SKYRIM_SYNTHETIC_IC=1, running on SEEDED SYNTHETIC fields; "
                       "saved files are stamped 'synthetic'")
    return SyntheticDataSource(channel_names, geom, state_fn=state_fn)


def get_initial_condition_for_model(model, data_source, time: datetime.datetime) -> torch.Tensor:
    """(B=1, n_history_levels, C, lat, lon) float32 on ``model.device`` -- earth2mip.initial_conditions'
    function of the same name as the reference calls it (utils.py:20)."""
    arrs = []
    for k in range(model.n_history_levels - 1, -1, -1):
        arrs.append(np.asarray(data_source[time - k * model.time_step], dtype=np.float32))
    x = torch.from_numpy(np.stack(arrs))
    return x.to(model.device).unsqueeze(0)
