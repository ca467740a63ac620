"""GraphcastModel wrapper -- /root/reference/skyrim/core/models/graphcast.py, with ``build_model`` returning the HIP GraphCast
TimeLoop instead of ``graphcast.load_time_loop_operational(registry.get_model("e2mip://graphcast"))``.

The reference drives GraphCast through its own loop (``stepper.initialize / stepper.step`` on xarray Datasets, then
``_to_global_da``, graphcast.py:68-118); the state it ends up with is the same ``(time, channel, lat, lon)`` array as the
other models, channel order CHANNELS (graphcast.py:17-26).  Here the engine works on that array directly, so the generic
``GlobalModel.forecast / rollout`` (two history levels) replace the Dataset round trip."""
from __future__ import annotations

from ...graphcast.spec import CHANNELS  # noqa: F401  (same list as the reference's graphcast.py:17-26)
from .base import GlobalModel


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
