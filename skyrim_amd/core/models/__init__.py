from .fourcastnet_v2 import FourcastnetV2Model
from .graphcast import GraphcastModel
from .pangu import PanguModel

# The reference registers pangu, fourcastnet, fourcastnet_v2, dlwp, graphcast, fuxi, fengwu
# (/root/reference/skyrim/core/models/__init__.py:9-17).  This build ships the hot paths of
# three of them; the others are not named by the north star and are absent rather than stubbed.
MODELS = {
    "pangu": PanguModel,
    "fourcastnet_v2": FourcastnetV2Model,
    "graphcast": GraphcastModel,
}
