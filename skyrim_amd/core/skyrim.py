"""``Skyrim`` facade -- /root/reference/skyrim/core/skyrim.py:12-95, same constructor, ``predict``,
``forecast`` and ``list_available_models``."""
from __future__ import annotations

import datetime
import logging

from .models import MODELS
from .models.base import GlobalModel, GlobalPrediction, adjust_lead_time
from .models.ensemble import GlobalEnsemble

logger = logging.getLogger("skyrim_amd")


class Skyrim:
    def __init__(self, *model_names: str, ic_source: str = "cds"):
        missing_names = [name for name in model_names if name not in MODELS]
        if missing_names:
            raise ValueError(f"Invalid model name(s): {missing_names}")
        self.model_names = model_names
        self.ic_source = ic_source
        self.model: GlobalEnsemble | GlobalModel
        if len(model_names) > 1:
            self.model = GlobalEnsemble(model_names, ic_source=ic_source)
        else:
            self.model = MODELS[model_names[0]](ic_source=ic_source)

    def __repr__(self) -> str:
        return f"Skyrim(models={self.model_names},ic={self.ic_source})"

    @staticmethod
    def list_available_models():
        return list(MODELS.keys())

    def forecast(self, start_time: datetime.datetime, n_steps: int = 4, channels: list | None = None):
        """Full concatenated forecast (all steps from the IC on) for the channels of interest."""
        start_time = start_time.replace(second=0, microsecond=0)
        return self.model.forecast(start_time=start_time, n_steps=n_steps, channels=channels or [])

    def predict(self, date: str, time: str, lead_time: int = 6, save: bool = False, save_config: dict | None = None):
        """Predict a single lead-time snapshot, optionally saving every intermediate step.
        date: YYYYMMDD, time: HHMM, lead_time in hours (clipped down to a multiple of 6, at least 6)."""
        start_time = datetime.datetime(int(date[:4]), int(date[4:6]), int(date[6:8]), int(time[:2]), int(time[2:4]))
        lead_time = adjust_lead_time(lead_time, step_size=6)
        n_steps = int(lead_time // (self.model.time_step.total_seconds() / 3600))
        pred, output_paths = self.model.rollout(start_time=start_time, n_steps=n_steps, save=save, save_config=save_config)
        return GlobalPrediction(pred, model_name=self.model_names), output_paths
