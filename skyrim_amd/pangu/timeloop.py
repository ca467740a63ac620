"""``PanguTimeLoop``: the earth2mip ``TimeLoop`` protocol on top of the HIP engine.

This is the drop-in seam of the reference: the object ``PanguModel.build_model()``
returns (/root/reference/skyrim/core/models/pangu.py:45-46) and ``run_basic_inference`` iterates
(/root/reference/skyrim/core/models/utils.py:34-40):

    for k, (time, output, restart) in enumerate(model(time, x)):   # k = 0 echoes the initial state

  * ``x``: float32 ``(B=1, n_history_levels=1, 69, n_lat, n_lon)`` on ``.device``
  * yields ``(time, (B, 69, n_lat, n_lon) tensor, restart)``; infinite, the caller breaks
  * every step is the 6-h network (the reference's rollout re-creates the loop each step, base.py:131-132)

The state stays in HBM between steps; the yielded tensor is a fresh buffer the caller owns.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass

import torch

from .. import weights
from .engine import DEFAULT_PRECISION, PanguEngine
from .spec import CHANNELS, PanguGeometry, init_synthetic


@dataclass
class Grid:
    lat: list
    lon: list

    @property
    def shape(self):
        return (len(self.lat), len(self.lon))


def _load_weights(path: str, geom: PanguGeometry) -> dict:
    """A torch state dict (``.pt``) keyed by ``spec.param_spec`` names, or the reference's own weight format: an ONNX
    file (``pangu_weather_6.onnx`` / ``_24.onnx``) read by ``onnx_weights.convert``; ``<path>.map.json`` next to it may
    hold an explicit {"mapping": ..., "extra": ...} when the automatic slot mapping reports unresolved slots."""
    if str(path).endswith(".onnx"):
        import json
        from . import onnx_weights
        side = str(path) + ".map.json"
        cfg = json.load(open(side)) if os.path.exists(side) else {}
        arrays = onnx_weights.convert(path, geom, cfg.get("mapping"), cfg.get("extra"))
        return {k: torch.from_numpy(v) for k, v in arrays.items()}
    return torch.load(path, map_location="cpu")


def _load_state(path: str, geom: PanguGeometry) -> torch.Tensor:
    """A (69, n_lat, n_lon) calibration state from a file: torch ``.pt``, numpy ``.npy`` or a netCDF forecast file (its last time level)."""
    if not os.path.exists(path):
        raise ValueError(f"calibration = {path!r}: 'synthetic', 'first', 'off' or the path of a state file")
    if path.endswith(".npy"):
        import numpy as np
        t = torch.from_numpy(np.load(path))
    elif path.endswith(".nc"):
        from ..labeled import open_dataarray
        t = torch.from_numpy(open_dataarray(path).values)
    else:
        t = torch.load(path, map_location="cpu")
    t = t.float()
    while t.dim() > 3:
        t = t[-1]
    if tuple(t.shape) != (geom.n_channels, geom.n_lat, geom.n_lon):
        raise ValueError(f"calibration state {path!r}: expected {(geom.n_channels, geom.n_lat, geom.n_lon)}, got {tuple(t.shape)}")
    return t.contiguous()


class PanguTimeLoop:
    # sigma: 512 x 2^-22 = 1.2e-4 of an O(1) signal.  An API divergence: the reference accepts any state; here an initial condition beyond the
    # limit raises FloatingPointError (one blocking .item() per forecast).  SKYRIM_PANGU_RANGE_LIMIT=<sigma> (or `inf`) overrides it.
    RANGE_LIMIT = float(os.environ.get("SKYRIM_PANGU_RANGE_LIMIT", "512"))
    n_history_levels = 1
    time_step = datetime.timedelta(hours=6)
    in_channel_names = list(CHANNELS)
    out_channel_names = list(CHANNELS)

    def __init__(self, params: dict | None = None, geom: PanguGeometry | None = None, precision: str = DEFAULT_PRECISION,
                 device: str | torch.device = "cuda:0", seed: int = 0, params24: dict | None = None, conventions: dict | None = None,
                 calibration: "torch.Tensor | str | None" = None, rounding: str = "default", guard: "bool | None" = None):
        """``params``: 6-h network (default: ``SKYRIM_PANGU_WEIGHTS`` state dict or seeded random init).
        ``params24`` (optional, or ``SKYRIM_PANGU_WEIGHTS_24``): the 24-h network; when present a multi-step
        generator interleaves the two like earth2mip's Pangu loop does (every 4th step is a 24-h step from the state
        24 h earlier, the 6-h network fills in between) -- what ``GlobalModel.forecast`` sees in the reference
        (base.py:105-107); ``rollout`` re-creates the loop every step and therefore only ever uses the 6-h network."""
        # ``conventions``: the points the public pseudocode leaves open and a real pangu_weather_6.onnx settles -- roll_sign, mask_value,
        # surface, qkv_order, bias_index (PanguEngine); padding placement is ``geom.pad``
        # ``calibration``: the state a term plan's biases are calibrated on (PanguEngine.load_params / calibrate): "synthetic" (the
        # built-in state from the weights' own normalisation constants: the DEFAULT, whatever the weights' source -- a set of weights is
        # always prepared the same way, so the same initial condition gives the same bits in every process), a (69, n_lat, n_lon) state,
        # a PATH to one (``.pt`` / ``.npy`` tensor or a saved forecast ``.nc``: its last time level -- e.g. a climatological analysis),
        # "off", or "first" (opt-in: the first initial condition this loop is called with; forecasts then depend on what the process saw
        # first).  SKYRIM_PANGU_CALIBRATION overrides the default with any of these.
        self.geom = geom or PanguGeometry()
        if calibration is None:
            calibration = os.environ.get("SKYRIM_PANGU_CALIBRATION", "synthetic")
        self._calibrate_on_first = isinstance(calibration, str) and calibration == "first"
        if self._calibrate_on_first:
            calibration = "off"
        elif isinstance(calibration, str) and calibration not in ("synthetic", "off"):
            calibration = _load_state(calibration, self.geom)
        conventions = dict(conventions or {})
        self.engine = PanguEngine(self.geom, precision, device, **conventions)
        if params is None:
            params = weights.resolve("SKYRIM_PANGU_WEIGHTS", lambda p: _load_weights(p, self.geom), lambda: init_synthetic(self.geom, seed), "pangu")
        # ``guard``: the engine's load-time precision guard (PanguEngine.load_params): on by default -- the plan in effect is ``self.term_plan``
        # (calibration "first": the plan is fitted -- and judged -- on the first initial condition, not at load time)
        load_guard = False if self._calibrate_on_first else guard
        self.engine.load_params(params, calibration=calibration, rounding=rounding, guard=load_guard)
        if params24 is None and os.environ.get("SKYRIM_PANGU_WEIGHTS_24"):
            params24 = _load_weights(os.environ["SKYRIM_PANGU_WEIGHTS_24"], self.geom)
        self._guard = guard
        self.engine24 = None
        if params24 is not None:
            self.engine24 = PanguEngine(self.geom, precision, device, **conventions)
            self.engine24.load_params(params24, calibration=calibration, rounding=rounding, guard=load_guard)
        self.grid = Grid(self.geom.lat, self.geom.lon)
        self._mean = params["norm.mean"].to(self.engine.device, torch.float32).reshape(-1, 1, 1)
        self._std = params["norm.std"].to(self.engine.device, torch.float32).reshape(-1, 1, 1)

    @property
    def device(self):
        return self.engine.device

    @property
    def term_plan(self) -> int:
        """The MFMA term plan the 6-h engine runs with (the default unless its load-time guard fell back; engine.guard_report has the figures)."""
        return self.engine.term_plan_in_effect

    def to(self, device):
        if torch.device(device) != self.engine.device:
            raise NotImplementedError("the engine's arenas are bound to one GPU; build a new PanguTimeLoop for another device")
        return self

    def take_pending_check(self):
        """(flag tensor, step, hint) of the last yielded state's non-finite check, taken OUT of the running loop -- the caller promises to read
        it before anybody reads that state (core/models/utils.py: with the state's copy to the host) -- or None."""
        g = self.__dict__.get("_active_guard")
        p = g.take() if g is not None else None
        return None if p is None else (p[0], p[1], g.hint)

    def release(self):
        """``skpangu_destroy`` + arenas dropped, for both networks (GlobalModel.release_model)."""
        for e in (self.engine, self.engine24):
            if e is not None:
                e.release()
        self._mean = self._std = None

    def __call__(self, time: datetime.datetime, x: torch.Tensor, restart=None):
        own = self.__dict__.pop("_state_is_own_output", False)       # run_basic_inference: ``x`` is this loop's last output, still in HBM
        if x.dim() != 5 or x.shape[0] != 1 or x.shape[1] != self.n_history_levels or tuple(x.shape[2:]) != self.engine.state_shape:
            raise ValueError(f"expected x of shape (1, 1, {', '.join(map(str, self.engine.state_shape))}), got {tuple(x.shape)}")
        state = x[0, 0].to(self.device, torch.float32).contiguous()
        # range check of the initial condition, once per forecast (not of a state the loop produced itself: that one passed the finite check): the engine carries activations as fp16 hi/lo planes -- 22 significant
        # bits per element -- so a channel that sits N sigma from its mean costs the O(1) signals it is mixed with N x 2^-22 (measured: two
        # channels at 1e4 sigma -> 2.8e-3 per-channel error, outside the 1e-3 bar, with finite output).  Analyses stay within tens of sigma;
        # beyond RANGE_LIMIT the forecast is refused rather than delivered degraded.
        z = 0.0 if own else ((state - self._mean) / self._std).abs().amax().item()
        if not (z <= self.RANGE_LIMIT):
            raise FloatingPointError(f"initial condition reaches {z:.3g} sigma from the channel means (limit {self.RANGE_LIMIT:g}): beyond what the "
                                     "engine's fp16 hi/lo operand planes resolve inside the 1e-3 bar -- check the units / channel order of the state")
        if self._calibrate_on_first:                       # once per loop object: rollout / forecast calls after it reuse the biases
            self._calibrate_on_first = False
            for e in (self.engine, self.engine24):
                if e is not None:
                    e._guard_on = self._guard
                    e.calibrate(state)
                    e.calibrated_on = "first"
        yield time, state.unsqueeze(0).clone(), restart
        state24, k = state, 0
        guard = weights.FiniteGuard(f"precision {self.engine.precision!r} keeps activations as fp16 planes (|x| < 65504); "
                                    "use PanguModel(precision='bf16x3') for the wide-range mode")
        self._active_guard = guard
        try:
            while True:
                k += 1
                if self.engine24 is not None and k % 4 == 0:
                    state = state24 = self.engine24.step(state24)      # 24-h step from the state 24 h earlier
                else:
                    state = self.engine.step(state)                    # new buffer each step: the caller keeps the yielded one
                time = time + self.time_step
                guard.push(state, k)
                yield time, state.unsqueeze(0), restart
        except GeneratorExit:
            # the consumer stopped (run_basic_inference breaks at k == n and closes the generator): the flag of the LAST yielded
            # step is still pending -- with n = 1 (predict_one_step / rollout) it is the only one there ever is.  Only on a clean
            # close: when step() itself raised, a device sync here could mask that error.
            guard.check()
            raise
