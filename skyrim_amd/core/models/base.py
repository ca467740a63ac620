"""GlobalModel / GlobalPrediction / GlobalPredictionRollout -- same names, arguments and behaviour as
/root/reference/skyrim/core/models/base.py (:13-15 adjust_lead_time, :18-146 GlobalModel, :149-274
GlobalPrediction, :277-303 GlobalPredictionRollout)."""
from __future__ import annotations

import datetime
import logging
import time
from pathlib import Path
from typing import List

from ...common import generate_forecast_id, save_forecast
from ...datasource import IC_SOURCES, get_data_source
from ...labeled import DataArray, open_dataarray
from .utils import run_basic_inference

logger = logging.getLogger("skyrim_amd")


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
        raise NotImplementedError

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

    def predict_one_step(self, start_time: datetime.datetime, initial_condition=None) -> DataArray:
        # if initial_condition is None, it is fetched from the self.ic_source
        return run_basic_inference(model=self.model, n=1, data_source=self.data_source, time=start_time, x=initial_condition)

    def forecast(self, start_time: datetime.datetime, n_steps: int = 3, channels: List[str] = []):
        da = run_basic_inference(model=self.model, n=n_steps, data_source=self.data_source, time=start_time, x=None)
        return da.sel(channel=channels) if channels else da

    def rollout(self, start_time: datetime.datetime, n_steps: int = 3, save: bool = True, save_config: dict = {}):
        # returns the final prediction (2 time entries) and the paths of the intermediate predictions
        pred, output_paths, source = None, [], self.ic_source
        forecast_id = save_config.get("forecast_id", generate_forecast_id())
        save_config.update({"forecast_id": forecast_id})
        for n in range(n_steps):
            pred = self.predict_one_step(start_time, initial_condition=pred)
            pred_time = start_time + self.time_step
            if save:
                output_paths.append(save_forecast(pred, self.model_name, start_time, pred_time, source, config=save_config))
            start_time, source = pred_time, "file"
            logger.info(f"Rollout step {n + 1}/{n_steps} completed")
        return pred, output_paths


class GlobalPrediction:
    filepath = None
    prediction = None

    def __init__(self, source, model_name: str = ""):
        self.model = model_name
        if isinstance(source, (str, Path)):
            self.filepath = Path(source)
            self.prediction = open_dataarray(source).squeeze()
        elif isinstance(source, DataArray):
            self.filepath = None
            self.prediction = source.squeeze()
        else:
            raise ValueError("Invalid source type.")

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
        info = self.filepath if self.filepath else f"{type(self.prediction).__name__} with shape {self.prediction.shape}"
        return f"GlobalPrediction(model={self.model},source={info})"

    def slice(self, lat=None, lon=None, channel=None, n_step=None):
        if channel is None:
            data = self.prediction
        else:
            assert channel in self.channels, f"Variable {channel} not found in dataset."
            data = self.prediction.sel(channel=channel)
        if lat:
            data = data.sel(lat=lat)
        if lon:
            data = data.sel(lon=lon)
        if n_step and "time" in data.dims:
            data = data.isel(time=n_step)
        return data

    def point(self, lat: float, lon: float, channel: str, n_step=1):
        if lon < 0:
            lon = 360 + lon
        assert channel in self.channels, f"Variable {channel} not found in dataset."
        if lat not in self.prediction.coords["lat"].values or lon not in self.prediction.coords["lon"].values:
            lat = self.prediction.sel(lat=lat, method="nearest").lat.item()
            lon = self.prediction.sel(lon=lon, method="nearest").lon.item()
            logger.warning(f"Exact coordinates not found. Using nearest values: Lat {lat}, Lon {lon}")
        return self.prediction.sel(lat=lat, lon=lon, channel=channel).isel(time=n_step).item()

    def point_wind_uv(self, lat: float, lon: float, pressure_level: int = 1000, n_step=1):
        u = self.point(lat=lat, lon=lon, channel=f"u{pressure_level}", n_step=n_step)
        v = self.point(lat=lat, lon=lon, channel=f"v{pressure_level}", n_step=n_step)
        return u, v

    def wind_speed(self, lat: float, lon: float, pressure_level: int, n_step=1):
        u, v = self.point_wind_uv(lat, lon, pressure_level, n_step)
        return (u ** 2 + v ** 2) ** 0.5

    def surface_wind_speed(self, lat: float, lon: float, n_step=1):
        return self.wind_speed(lat, lon, pressure_level=1000, n_step=n_step)


class GlobalPredictionRollout:
    def __init__(self, rollout: list):
        self.rollout = [GlobalPrediction(source) for source in rollout]
        self.time_steps = [r.prediction.time.values[-1] for r in self.rollout]

    def __repr__(self):
        return f"<GlobalPredictionRollout with {len(self.rollout)} predictions, last times: {self.time_steps}>"

    def wind_speed(self, lat: float, lon: float, pressure_level: int, n_step=1):
        return [pred.wind_speed(lat, lon, pressure_level, n_step) for pred in self.rollout]

    def surface_wind_speed(self, lat: float, lon: float, n_step=1):
        return self.wind_speed(lat, lon, pressure_level=1000, n_step=n_step)
