"""FourcastnetV2Model wrapper -- /root/reference/skyrim/core/models/fourcastnet_v2.py, with ``build_model`` returning
the HIP SFNO TimeLoop instead of ``fcnv2_sm.load(registry.get_model("e2mip://fcnv2_sm"))``."""
from __future__ import annotations

from ...sfno.spec import CHANNELS  # noqa: F401  (same list as the reference's fourcastnet_v2.py:12-21)
from .base import GlobalModel


class FourcastnetV2Model(GlobalModel):
    """
    n_history_levels: int = 1
    grid.lat: list of length 721, [90, 89.75, 89.50, ..., -89.75, -90]
    grid.lon: list of length 1440, [0.0, 0.25, ..., 359.75]
    in_channel_names / out_channel_names: list of length 73, ['u10m', 'v10m', 'u100m', 'v100m', ..., 'r1000']
    """

    model_name = "fourcastnet_v2"

    def __init__(self, *args, cfg=None, device="cuda:0", params=None, **kwargs):
        # extras beyond the reference's signature (all optional): network configuration, device, parameter dict
        self._engine_kw = dict(cfg=cfg, device=device, params=params)
        super().__init__(self.model_name, *args, **kwargs)

    def build_model(self):
        from ...sfno.timeloop import SfnoTimeLoop
        return SfnoTimeLoop(**self._engine_kw)

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
