"""GraphcastModel wrapper -- /root/reference/skyrim/core/models/graphcast.py, with ``build_model`` returning the HIP GraphCast
TimeLoop instead of ``graphcast.load_time_loop_operational(registry.get_model("e2mip://graphcast"))``.

The reference drives GraphCast through its own loop: ``stepper.initialize / stepper.step`` on a (time, Dataset, rng) state
(graphcast.py:93-120), converts the Dataset with ``_to_global_da`` (:68-91; channel order = CHANNEL_MAP, which differs from
CHANNELS) and flips latitude in ``forecast`` only (:138; ``rollout`` saves the stepper's ascending latitudes, :144-177).  All of
that is mirrored here on ``labeled.Dataset / DataArray``; the two time levels stay in HBM between steps."""
from __future__ import annotations

import datetime
import logging

import numpy as np
import torch

from ...common import generate_forecast_id, save_forecast
from ...datasource import get_initial_condition_for_model
from ...graphcast.spec import CHANNELS  # noqa: F401  (same list as the reference's graphcast.py:17-26)
from ...labeled import DataArray, Dataset, concat, open_dataarray
from .base import GlobalModel

logger = logging.getLogger("skyrim_amd")

# (dataset variable, channel code): the order _to_global_da emits -- graphcast.py:29-41
CHANNEL_MAP = [
    ("specific_humidity", "q"), ("geopotential", "z"), ("temperature", "t"), ("u_component_of_wind", "u"),
    ("v_component_of_wind", "v"), ("vertical_velocity", "w"), ("2m_temperature", "t2m"), ("10m_u_component_of_wind", "u10m"),
    ("10m_v_component_of_wind", "v10m"), ("mean_sea_level_pressure", "msl"), ("toa_incident_solar_radiation", "tp06"),
]
N_LEVEL_VARS = 6


class GraphcastModel(GlobalModel):
    model_name = "graphcast"

    def __init__(self, *args, cfg=None, device="cuda:0", params=None, **kwargs):
        # extras beyond the reference's signature (all optional): network configuration, device, parameter dict
        self._engine_kw = dict(cfg=cfg, device=device, params=params)
        super().__init__(self.model_name, *args, **kwargs)

    def build_model(self):
        from ...graphcast.timeloop import GraphcastTimeLoop
        return GraphcastTimeLoop(**self._engine_kw)

    @property
    def device(self):
        return self.model.device

    @property
    def time_step(self):
        return self.model.time_step

    @property
    def in_channel_names(self):
        return self.model.in_channel_names

    @property
    def out_channel_names(self):
        return self.model.out_channel_names

    # ---- Dataset -> (time, channel, lat, lon) --------------------------------------------------------------- #
    def _to_global_da(self, ds: Dataset) -> DataArray:
        """GraphCast dataset -> the global array layout of the other models: channels named ``{code}{level}`` for the six
        pressure-level variables (levels in the dataset's own order), then the five surface variables, all in CHANNEL_MAP order;
        dims (time, channel, lat, lon); latitude as the dataset has it."""
        ds = ds.squeeze(dim="batch")
        names, blocks, ref = [], [], None
        for k, (var, code) in enumerate(CHANNEL_MAP):
            v = ds[var]
            if k < N_LEVEL_VARS:
                v = v.transpose("time", "level", "lat", "lon")
                names += [f"{code}{lev}" for lev in v._coords["level"].tolist()]
                blocks.append(v.values)
            else:
                v = v.transpose("time", "lat", "lon")
                names.append(code)
                blocks.append(v.values[:, None])
            ref = v
        coords = dict(time=ref._coords["time"], channel=np.array(names), lat=ref._coords["lat"], lon=ref._coords["lon"])
        return DataArray(np.concatenate(blocks, axis=1), ["time", "channel", "lat", "lon"], coords)

    def _state_from_array(self, ic, start_time: datetime.datetime):
        """A saved prediction (path or DataArray, >= 2 time entries, any channel order / latitude direction) -> stepper state."""
        da = open_dataarray(ic) if not isinstance(ic, DataArray) else ic
        da = da.sel(channel=self.in_channel_names)
        vals = np.asarray(da.values[-2:], dtype=np.float32)
        if da._coords["lat"][0] < da._coords["lat"][-1]:
            vals = vals[:, :, ::-1]
        x = torch.from_numpy(np.ascontiguousarray(vals)).unsqueeze(0)
        return self.model.stepper.initialize(x, start_time)

    def _predict_one_step(self, start_time: datetime.datetime, initial_condition: tuple | None = None):
        """One stepper step; returns the new STATE (time, dataset, rng), as the reference does (graphcast.py:93-120)."""
        self.stepper = self.model.stepper
        if initial_condition is None:
            x = get_initial_condition_for_model(self.model, self.data_source, start_time)
            state = self.stepper.initialize(x, start_time)
        elif isinstance(initial_condition, tuple):
            state = initial_condition
        else:
            state = self._state_from_array(initial_condition, start_time)
        state, _ = self.stepper.step(state)
        return state

    def forecast(self, start_time: datetime.datetime, n_steps: int = 4, channels=None):
        """(n_steps + 1, channel, lat 90..-90, lon): the first step contributes both of its time levels (IC + first prediction),
        every later step its newest one; latitude flipped back to descending here and only here (graphcast.py:122-142)."""
        times = [start_time + i * self.time_step for i in range(n_steps + 1)]
        state, parts = None, []
        for n in range(n_steps):
            state = self._predict_one_step(start_time, initial_condition=state)
            ds = state[1] if n == 0 else state[1].isel(time=-1).expand_dims("time")
            da = self._to_global_da(ds)
            if channels:
                da = da.sel(channel=list(channels))
            parts.append(da.isel(lat=slice(None, None, -1)))
            logger.info(f"Forecast step {n + 1}/{n_steps} completed")
        return concat(parts, dim="time").assign_coords(time=times)

    def rollout(self, start_time: datetime.datetime, n_steps: int = 3, save: bool = True, save_config: dict | None = None,
                initial_condition=None):
        """Final two time levels + per-step file paths; the stepper state is fed back step to step (never through a file).
        Files carry the stepper's latitude order (ascending), as the reference writes them."""
        times = [start_time + i * self.time_step for i in range(n_steps + 1)]
        cfg = dict(save_config or {})
        cfg.setdefault("forecast_id", generate_forecast_id())
        if save_config is not None:
            save_config["forecast_id"] = cfg["forecast_id"]
        state, output_paths = initial_condition, []
        source = "file" if initial_condition is not None else self.source_label
        for n in range(n_steps):
            state = self._predict_one_step(start_time, initial_condition=state)
            pred_time = start_time + self.time_step
            if save:
                da = self._to_global_da(state[1]).assign_coords(time=[start_time, pred_time])
                output_paths.append(save_forecast(da, self.model_name, start_time, pred_time, source, config=cfg))
            start_time, source = pred_time, "file"
            logger.info(f"Rollout step {n + 1}/{n_steps} completed")
        return self._to_global_da(state[1]).assign_coords(time=times[-2:]), output_paths
