"""PanguModel wrapper -- /root/reference/skyrim/core/models/pangu.py, with ``build_model`` returning the
HIP-engine TimeLoop instead of ``earth2mip.networks.pangu.load(registry.get_model("e2mip://pangu"))``."""
from __future__ import annotations

from ...pangu.spec import CHANNELS  # noqa: F401  (same list as the reference's pangu.py:6-13)
from ...pangu.engine import DEFAULT_PRECISION
from .base import GlobalModel


class PanguModel(GlobalModel):
    """
    n_history_levels: int = 1
    grid.lat: list of length 721, [90, 89.75, 89.50, ..., -89.75, -90]
    grid.lon: list of length 1440, [0.0, 0.25, ..., 359.75]
    in_channel_names / out_channel_names: list of length 69, ["z1000", "z925", ..., "t2m"]
    """

    model_name = "pangu"

    def __init__(self, *args, geom=None, precision: str = DEFAULT_PRECISION, device="cuda:0", params=None, conventions=None, **kwargs):
        # extras beyond the reference's signature (all optional): grid geometry (default 721x1440), MFMA precision mode, device, a
        # parameter dict (default: SKYRIM_PANGU_WEIGHTS or seeded random init) and the open conventions (PanguTimeLoop)
        self._engine_kw = dict(geom=geom, precision=precision, device=device, params=params, conventions=conventions)
        super().__init__(self.model_name, *args, **kwargs)

    def build_model(self):
        from ...pangu.timeloop import PanguTimeLoop
        return PanguTimeLoop(**self._engine_kw)

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
