from .models.base import GlobalPredictionRollout  # noqa: F401
from .skyrim import Skyrim  # noqa: F401
