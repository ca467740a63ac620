"""GlobalModel / GlobalPrediction / GlobalPredictionRollout -- the names, arguments and results of
/root/reference/skyrim/core/models/base.py (:13-15 adjust_lead_time, :18-146 GlobalModel, :149-274
GlobalPrediction, :277-303 GlobalPredictionRollout).  The model side drives a HIP TimeLoop; the prediction side works on
``labeled.DataArray`` with regular-grid index arithmetic (lat / lon are uniform axes, so "nearest" is a division, not a search)."""
from __future__ import annotations

import datetime
import logging
import os
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import List

import numpy as np

from ...common import generate_forecast_id, save_forecast
from ...datasource import IC_SOURCES, get_data_source
from ...labeled import DataArray, open_dataarray
from .utils import run_basic_inference

logger = logging.getLogger("skyrim_amd")

STEPS_AHEAD = 3           # steps ``rollout`` lets the host queue in front of the GPU
SAVE_WORKERS = 8          # per-step netCDF files written at once by ``rollout`` (SKYRIM_SAVE_WORKERS; one file is one pwrite stream of ~6 GB/s, files side by side scale: docs/experiments.md A6.5)


def adjust_lead_time(lead_time: int, step_size: int = 6):
    """Adjust lead time to the nearest multiple of step_size"""
    return max(step_size, (lead_time // step_size) * step_size)


class GlobalModel:
    def __init__(self, model_name: str, ic_source: str = "cds"):
        clock = time.time()
        if ic_source not in IC_SOURCES:
            raise ValueError(f"Invalid initial condition source: {ic_source}")
        self.model_name = model_name
        self.ic_source = ic_source
        self.model = self.build_model()
        self.data_source = self.build_datasource()
        logger.info(f"Initialized {model_name} in {time.time() - clock:.1f} seconds")

    def build_model(self):
        raise NotImplementedError

    def build_datasource(self):
        return get_data_source(self.model.in_channel_names, initial_condition_source=self.ic_source,
                               geom=getattr(self.model, "geom", None), state_fn=getattr(self.model, "synthetic_state", None))

    def release_model(self):
        """Give the model's device memory back, deterministically (the reference's TODO, base.py:50-55; what its ensemble does between
        members, ensemble.py:39-48): the TimeLoop destroys its engine contexts (``sk*_destroy``) and drops every arena / prepared tensor,
        then the caching allocator is emptied.  The wrapper is unusable afterwards (``build_model()`` again for a new one)."""
        m, self.model = getattr(self, "model", None), None
        if getattr(self, "stepper", None) is not None:
            self.stepper = None
        if m is not None and hasattr(m, "__dict__"):
            m._resident_state = None                            # the device copy of the last delivered states (utils.ResidentState)
        if m is not None and hasattr(m, "release"):
            m.release()
        del m
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    @property
    def time_step(self):
        raise NotImplementedError

    def time_steps(self, lead_time: int):
        lead_time = adjust_lead_time(lead_time, step_size=6)
        return int(lead_time // (self.model.time_step.total_seconds() / 3600))

    @property
    def in_channel_names(self):
        raise NotImplementedError

    @property
    def out_channel_names(self):
        raise NotImplementedError

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(model_name={self.model_name})"

    @property
    def source_label(self) -> str:
        """What saved files are stamped with: the requested source, or "synthetic" when the data source is the seeded stand-in."""
        return getattr(self.data_source, "label", None) or self.ic_source

    def predict_one_step(self, start_time: datetime.datetime, initial_condition=None) -> DataArray:
        # if initial_condition is None, it is fetched from the self.ic_source
        return self._step_delivering(start_time, initial_condition, None)

    def _mark_step(self):
        """An event behind everything queued so far on the model's GPU (None for a model that is not on one)."""
        import torch
        dev = getattr(self.model, "device", None)
        if dev is None or torch.device(dev).type != "cuda":
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(torch.device(dev)))
        return ev

    def _step_delivering(self, start_time, initial_condition, deliver, defer_check=False) -> DataArray:
        # (not ``_predict_one_step``: the GraphCast wrapper has a method of that name, as the reference's does)
        return run_basic_inference(model=self.model, n=1, data_source=self.data_source, time=start_time, x=initial_condition, deliver=deliver,
                                   defer_check=defer_check)

    def forecast(self, start_time: datetime.datetime, n_steps: int = 3, channels: List[str] = []):
        da = run_basic_inference(model=self.model, n=n_steps, data_source=self.data_source, time=start_time, x=None)
        return da.sel(channel=channels) if channels else da

    def rollout(self, start_time: datetime.datetime, n_steps: int = 3, save: bool = True, save_config: dict | None = None,
                initial_condition=None):
        """Final prediction (2 time entries) + the paths of the per-step files.

        ``initial_condition`` (the reference's base.py:127 TODO): a saved forecast path or a DataArray whose last
        ``n_history_levels`` time entries are the state at ``start_time``; default: fetched from the data source.
        A ``save_config`` dict handed in by the caller receives the forecast id drawn here, as in the reference
        (base.py:129-130); there is no shared ``{}`` default, so an id never leaks from one call into the next."""
        cfg = dict(save_config or {})
        cfg.setdefault("forecast_id", generate_forecast_id())
        if save_config is not None:
            save_config["forecast_id"] = cfg["forecast_id"]
        pred, pending = initial_condition, []
        source = "file" if initial_condition is not None else self.source_label
        # Step k's file is written by a worker thread while the next steps run (the reference writes 573 MB synchronously between two
        # steps, base.py:134-143).  netCDF targets are one independent file per step, so up to SAVE_WORKERS of them are written at once:
        # the page cache takes ONE file's bytes at ~6 GB/s however many threads push them (buffered writes serialise on the inode,
        # tools/predict_cost.py), different files do not share that lock.  zarr appends to one store and keeps a single worker, in step
        # order.  At most ``workers + 1`` predictions wait for the disk, so a slow target throttles the rollout instead of filling host
        # memory.  Same files, same paths, returned in step order.  Divergences from the reference's serial loop, both opt-out: a
        # user-supplied ``save_config["mapping_func"]`` runs on the save threads, several at once (SKYRIM_SAVE_WORKERS=1 restores one file
        # at a time, in step order); once a step's write has failed no further step is submitted, but writes already running finish.
        workers = 1
        netcdf_local = save and (cfg.get("file_type") or "netcdf") == "netcdf" and "://" not in str(cfg.get("output_dir", ""))
        if netcdf_local:
            workers = max(1, int(os.environ.get("SKYRIM_SAVE_WORKERS", SAVE_WORKERS)))
        pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="skyrim-save") if save else None
        # netCDF-3 holds big-endian floats.  When the file is the prediction as delivered (no mapping_func, no channel filter) its bytes
        # are produced in HBM and copied to the host as they will lie in the file (skyrim_amd/deliver.py): the save threads only pwrite.
        # Intermediate steps bring ONLY that image over PCIe -- the next step reads the state from HBM, nobody reads their ``values``
        # (which would be filled from the image on demand); the last step, returned to the caller, brings both.
        own_step = type(self).predict_one_step is GlobalModel.predict_one_step       # (an overriding subclass is called as the reference would)
        image = netcdf_local and cfg.get("mapping_func") is None and not cfg.get("filter_vars") and own_step
        # The loop never waits for the GPU: a step's non-finite check is read with its copy to the host -- by the save thread, whose
        # exception stops the rollout below, or at the end of the rollout -- instead of between two steps, where the host would sit
        # out every step before queueing the next (17.4 -> 14.6 ms per step at 721x1440, the engine's own pace).
        ahead = []
        try:
            for n in range(n_steps):
                if own_step:
                    last = n == n_steps - 1
                    deliver = ("both" if last else "be") if image else (None if last or save else "skip")
                    pred = self._step_delivering(start_time, pred, deliver, defer_check=True)
                    ahead.append(self._mark_step())
                    if len(ahead) > STEPS_AHEAD:                   # the host queues at most STEPS_AHEAD steps in front of the GPU (each holds its output buffer)
                        mark = ahead.pop(0)
                        if mark is not None:
                            mark.synchronize()
                else:
                    pred = self.predict_one_step(start_time, initial_condition=pred)
                pred_time = start_time + self.time_step
                if save:
                    failed = next((f for f in pending if f.done() and f.exception() is not None), None)
                    if failed is not None:
                        failed.result()                            # a writer failed: stop the rollout here instead of writing later steps
                    pending.append(pool.submit(save_forecast, pred, self.model_name, start_time, pred_time, source, config=cfg))
                    if len(pending) > workers + 1:
                        pending[-(workers + 2)].result()
                start_time, source = pred_time, "file"
                logger.info(f"Rollout step {n + 1}/{n_steps} completed")
            output_paths = [f.result() for f in pending]         # re-raises a writer's exception here, in step order
            if own_step and n_steps > 0 and isinstance(pred, DataArray):
                pred.values                                      # the last state's copy and check: a rollout returns numbers that passed it
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
        return pred, output_paths


def _axis_index(axis: np.ndarray, x: float) -> tuple[int, bool]:
    """(index of the axis value nearest to x, whether it is an exact hit).  Uniform axes (lat 90..-90, lon 0..359.75) are
    inverted arithmetically; anything else falls back to a linear scan."""
    n = axis.shape[0]
    if n > 1:
        step = (float(axis[-1]) - float(axis[0])) / (n - 1)
        if step != 0.0 and np.allclose(np.diff(axis), step, rtol=0, atol=abs(step) * 1e-6):
            i = int(min(max(round((x - float(axis[0])) / step), 0), n - 1))
            return i, bool(axis[i] == x)
    i = int(np.abs(axis - x).argmin())
    return i, bool(axis[i] == x)


class GlobalPrediction:
    """A forecast result, in memory or on disk, with point / wind accessors (reference base.py:149-274)."""

    filepath = None
    prediction = None

    def __init__(self, source, model_name: str = ""):
        self.model = model_name
        if isinstance(source, DataArray):
            data, self.filepath = source, None
        elif isinstance(source, (str, Path)):
            self.filepath = Path(source)
            data = open_dataarray(self.filepath)
        else:
            raise ValueError("Invalid source type.")
        self.prediction = data.squeeze()          # size-1 dims dropped, as the reference does

    # ---- what the array carries ----------------------------------------- #
    @property
    def coords(self):
        return self.prediction.coords

    @property
    def size(self):
        return self.prediction.size

    @property
    def channels(self):
        return self.prediction.channel

    def __repr__(self) -> str:
        where = self.filepath or f"{type(self.prediction).__name__} with shape {self.prediction.shape}"
        return f"GlobalPrediction(model={self.model},source={where})"

    def _need(self, channel: str):
        assert channel in self.channels, f"Variable {channel} not found in dataset."

    # ---- sub-arrays -------------------------------------------------------- #
    def slice(self, lat=None, lon=None, channel=None, n_step=None):
        """Sub-array over whichever of (lat range, lon range, one channel, time positions) are given."""
        out = self.prediction
        if channel is not None:
            self._need(channel)
            out = out.sel(channel=channel)
        for dim, rng in (("lat", lat), ("lon", lon)):
            if rng:
                out = out.sel(**{dim: rng})
        if n_step and "time" in out.dims:
            out = out.isel(time=n_step)
        return out

    # ---- point values --------------------------------------------------------- #
    def point(self, lat: float, lon: float, channel: str, n_step=1):
        """Value of ``channel`` at the grid cell nearest to (lat, lon); negative longitudes wrap to [0, 360)."""
        lon = lon + 360 if lon < 0 else lon
        self._need(channel)
        p = self.prediction
        i, hit_lat = _axis_index(p._coords["lat"], lat)
        j, hit_lon = _axis_index(p._coords["lon"], lon)
        if not (hit_lat and hit_lon):
            logger.warning(f"Exact coordinates not found. Using nearest values: Lat {p._coords['lat'][i]}, Lon {p._coords['lon'][j]}")
        index = []
        for dim in p.dims:
            if dim == "lat":
                index.append(i)
            elif dim == "lon":
                index.append(j)
            elif dim == "channel":
                index.append(p._coords["channel"].tolist().index(channel))
            elif dim == "time":
                index.append(n_step)
            else:
                index.append(slice(None))
        return p.values[tuple(index)].item()

    def point_wind_uv(self, lat: float, lon: float, pressure_level: int = 1000, n_step=1):
        return tuple(self.point(lat=lat, lon=lon, channel=f"{c}{pressure_level}", n_step=n_step) for c in "uv")

    def wind_speed(self, lat: float, lon: float, pressure_level: int, n_step=1):
        """|(u, v)| at a pressure level in hPa."""
        u, v = self.point_wind_uv(lat, lon, pressure_level, n_step)
        return float(np.hypot(u, v))

    def surface_wind_speed(self, lat: float, lon: float, n_step=1):
        return self.wind_speed(lat, lon, pressure_level=1000, n_step=n_step)

    def wind_speed_field(self, pressure_level: int = 1000, n_step=1, device=None):
        """Whole-grid |(u, v)| of one time entry (lat, lon).  With ``device`` (e.g. the model's GPU) the reduction runs there
        through torch; the point accessors above are the reference's surface, this is the field form of the same quantity."""
        u = self.slice(channel=f"u{pressure_level}").isel(time=n_step).values
        v = self.slice(channel=f"v{pressure_level}").isel(time=n_step).values
        if device is None:
            return np.hypot(u, v)
        import torch
        return torch.hypot(torch.as_tensor(u, device=device), torch.as_tensor(v, device=device))


class GlobalPredictionRollout:
    """A list of per-step predictions (paths or arrays) queried together (reference base.py:277-303)."""

    def __init__(self, rollout: list):
        self.rollout = [GlobalPrediction(source) for source in rollout]
        self.time_steps = [r.prediction.time.values[-1] for r in self.rollout]

    def __repr__(self):
        return f"<GlobalPredictionRollout with {len(self.rollout)} predictions, last times: {self.time_steps}>"

    def wind_speed(self, lat: float, lon: float, pressure_level: int, n_step=1):
        return [pred.wind_speed(lat, lon, pressure_level, n_step) for pred in self.rollout]

    def surface_wind_speed(self, lat: float, lon: float, n_step=1):
        return self.wind_speed(lat, lon, pressure_level=1000, n_step=n_step)
