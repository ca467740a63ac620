"""Ensembles of models.  ``GlobalEnsemble`` follows /root/reference/skyrim/core/models/ensemble.py: the members run ONE AT A TIME on the GPU
(:86-98 -- load, rollout, release in a ``finally``), the final predictions are averaged over the channels every member shares (:51-67), and with
``save=True`` the per-step ENSEMBLE-MEAN files are written under ``{output_dir}/{sorted names joined by _}/`` as
``{a_b}__{source}__{start}__{end}.nc`` and THEIR paths are what ``rollout`` returns (:103-128).  The reference's own ``rollout`` cannot complete
(it hands ``output_dir=`` to ``GlobalModel.rollout``, which has no such argument, and joins ``OUTPUT_DIR / prefix`` on a ``str``: SURVEY.md 3.5);
the contract it evidently intends is what is implemented here."""
from __future__ import annotations

import datetime
import logging
from pathlib import Path

from ...common import OUTPUT_DIR
from ...labeled import concat, open_dataarray
from .base import GlobalPrediction

logger = logging.getLogger("skyrim_amd")


def _on_grid_of(first, p):
    """``p`` on ``first``'s latitude / longitude order.  ``xr.concat`` aligns members by coordinate LABEL (reference :64); the arrays here are
    positional, and GraphCast delivers its latitudes ascending where the others descend (core/models/graphcast.py), so a reversed axis is
    turned round and anything else is refused instead of averaged cell against the wrong cell."""
    import numpy as np
    for dim in ("lat", "lon"):
        if dim not in p.dims or dim not in first.dims:
            continue
        a, b = np.asarray(first._coords[dim]), np.asarray(p._coords[dim])
        if a.shape == b.shape and np.array_equal(a, b):
            continue
        if a.shape == b.shape and np.array_equal(a, b[::-1]):
            p = p.isel(**{dim: slice(None, None, -1)})
        else:
            raise ValueError(f"ensemble members are on different {dim} axes")
    return p


class GlobalEnsemble:
    def __init__(self, model_names, ic_source: str = "cds", model_kwargs: "dict | None" = None):
        """``model_kwargs`` (beyond the reference's signature): ``{model name: constructor keywords}`` for members that are not built with
        their defaults (grid geometry, parameters, device ...)."""
        from . import MODELS
        missing = [n for n in model_names if n not in MODELS]
        if missing:
            raise ValueError(f"Models {missing} are not available in MODELS.")
        self.model_names = model_names
        self.ic_source = ic_source
        self.model_kwargs = dict(model_kwargs or {})
        self.common_channels = None
        self._model = None

    @property
    def time_step(self):
        return datetime.timedelta(hours=6)

    def __repr__(self) -> str:
        return f"GlobalEnsemble({self.model_names})"

    def _load_model(self, model_name):
        """Build the member on the GPU (reference :30-38: ``MODELS[name]()`` + ``.to("cuda")``; the HIP engines are built on their device)."""
        from . import MODELS
        logger.debug(f"Loading {model_name} model.")
        self._model = MODELS[model_name](ic_source=self.ic_source, **self.model_kwargs.get(model_name, {}))

    def _release_model(self):
        """Release the current member's device memory (reference :39-48): ``GlobalModel.release_model`` -- engine contexts destroyed,
        arenas dropped, caching allocator emptied -- before the next member is built."""
        if self._model is not None:
            logger.debug(f"Releasing {self._model.__class__.__name__} model.")
            self._model.release_model()
        self._model = None

    def _ensemble_predictions(self, predictions):
        """Average predictions along shared channels."""
        if not predictions:
            raise ValueError("No predictions to average or no common channels available.")
        common = [c for c in predictions[0].channel.values.tolist() if all(c in p.channel.values for p in predictions)]
        if not common:
            raise ValueError("No predictions to average or no common channels available.")
        self.common_channels = common
        return concat([_on_grid_of(predictions[0], p.sel(channel=common)) for p in predictions], dim="model").mean(dim="model")

    def predict_one_step(self, start_time: datetime.datetime, save: bool = False):
        """Subclasses should implement this method."""
        raise NotImplementedError

    def rollout(self, start_time: datetime.datetime, n_steps: int = 3, save: bool = True, save_config: dict | None = None):
        """Roll every member out, one at a time; returns (mean of the members' final predictions over the shared channels, paths of the per-step
        ensemble-mean files -- ``[]`` with ``save=False``).  The members' own per-step files stay where ``GlobalModel.rollout`` wrote them
        (``self.member_paths``)."""
        cfg = dict(save_config or {})
        if save and (cfg.get("file_type") or "netcdf") != "netcdf":
            raise ValueError("GlobalEnsemble.rollout(save=True) averages the members' per-step netCDF files; file_type must be 'netcdf'")
        predictions, output_paths = [], []
        for name in self.model_names:
            self._load_model(name)
            try:
                pred, paths = self._model.rollout(start_time=start_time, n_steps=n_steps, save=save, save_config=dict(cfg))
                output_paths.extend(paths)
                predictions.append(pred)
            finally:
                self._release_model()
        self.member_paths = list(output_paths)
        averaged = self._ensemble_predictions(predictions)
        ens_output_paths = []
        if save:
            logger.debug("Calculating and saving ensemble predictions.")
            ens_output_paths = self._save_ensembled_outputs(output_paths, n_steps, Path(cfg.get("output_dir") or OUTPUT_DIR))
        return averaged, ens_output_paths

    def _save_ensembled_outputs(self, output_paths, n_steps, output_dir):
        """Per step: the members' files of that step (every member wrote ``n_steps`` files, in step order), averaged over the shared channels,
        as ``{prefix}/{prefix}__{source}__{start}__{end}.nc`` (reference :103-128).  netCDF member files only (the naming is parsed back)."""
        output_dir = Path(output_dir)
        ens_prefix = "_".join(sorted(self.model_names))
        ens_directory = output_dir / ens_prefix
        ens_directory.mkdir(parents=True, exist_ok=True)
        if len(output_paths) != n_steps * len(self.model_names):
            raise ValueError(f"expected {n_steps} per-step files from each of {len(self.model_names)} members, got {len(output_paths)} paths")
        ens_output_paths = []
        for s in range(n_steps):
            step_paths = [Path(p) for p in output_paths[s::n_steps]]
            _, source, start_time, end_time = step_paths[0].stem.split("__")
            ens_pred = self._ensemble_predictions([open_dataarray(p) for p in step_paths])
            file_path = ens_directory / f"{ens_prefix}__{source}__{start_time}__{end_time}.nc"
            ens_pred.to_netcdf(file_path)
            ens_output_paths.append(file_path)
        return ens_output_paths


class GlobalEnsemblePrediction(GlobalPrediction):
    """A saved / in-memory ensemble-mean prediction with the point and wind accessors of ``GlobalPrediction`` (reference :131-133)."""

    def __init__(self, source):
        super().__init__(source)
