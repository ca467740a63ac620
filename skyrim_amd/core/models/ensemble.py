"""Ensembles.  ``GlobalEnsemble`` keeps the reference's semantics -- mean over a new leading dimension on
the channels every member shares (/root/reference/skyrim/core/models/ensemble.py:51-67) -- and actually
runs (the reference's ``rollout`` passes an unknown ``output_dir=`` kwarg and cannot complete)."""
from __future__ import annotations

import datetime

from ...labeled import concat


class GlobalEnsemble:
    def __init__(self, model_names, ic_source: str = "cds"):
        from . import MODELS
        missing = [n for n in model_names if n not in MODELS]
        if missing:
            raise ValueError(f"Models {missing} are not available in MODELS.")
        self.model_names = model_names
        self.ic_source = ic_source
        self.common_channels = None

    @property
    def time_step(self):
        return datetime.timedelta(hours=6)

    def __repr__(self) -> str:
        return f"GlobalEnsemble({self.model_names})"

    def _ensemble_predictions(self, predictions):
        """Average predictions along shared channels."""
        if not predictions:
            raise ValueError("No predictions to average or no common channels available.")
        common = [c for c in predictions[0].channel.values.tolist() if all(c in p.channel.values for p in predictions)]
        if not common:
            raise ValueError("No predictions to average or no common channels available.")
        self.common_channels = common
        return concat([p.sel(channel=common) for p in predictions], dim="model").mean(dim="model")

    def rollout(self, start_time: datetime.datetime, n_steps: int = 3, save: bool = True, save_config: dict | None = None):
        from . import MODELS
        predictions, output_paths = [], []
        for name in self.model_names:
            model = MODELS[name](ic_source=self.ic_source)
            pred, paths = model.rollout(start_time=start_time, n_steps=n_steps, save=save, save_config=dict(save_config or {}))
            predictions.append(pred)
            output_paths.extend(paths)
            del model
        return self._ensemble_predictions(predictions), output_paths
