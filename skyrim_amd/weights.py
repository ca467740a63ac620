"""Where a TimeLoop's parameters come from when the caller passes none.

The reference downloads checkpoints through ``earth2mip.registry.get_model("e2mip://...")`` (/root/reference/skyrim/core/models/
pangu.py:46, fourcastnet_v2.py:37, graphcast.py:52-54).  There is no network here: a weight file is named by an environment
variable, and the seeded random-init stand-in is used ONLY when ``SKYRIM_SYNTHETIC_WEIGHTS=1`` opts in -- a forecast through a
random network must never happen silently."""
from __future__ import annotations

import logging
import os
from typing import Callable

logger = logging.getLogger("skyrim_amd")


def resolve(env_var: str, load: Callable[[str], dict], synthetic: Callable[[], dict], model: str) -> dict:
    path = os.environ.get(env_var)
    if path:
        return load(path)
    if os.environ.get("SKYRIM_SYNTHETIC_WEIGHTS") == "1":
        logger.warning(f"{model}: no weight file ({env_var} unset); SKYRIM_SYNTHETIC_WEIGHTS=1 -> SEEDED RANDOM parameters")
        return synthetic()
    raise RuntimeError(f"{model}: no parameters given and {env_var} is unset.  Pass ``params=``, point {env_var} at a weight file, or set "
                       "SKYRIM_SYNTHETIC_WEIGHTS=1 to run on seeded random-init parameters (benchmarks / tests)")


class FiniteGuard:
    """Deferred non-finite check of the yielded states: the flag of step k is computed on the stream and read when step k + 1
    is about to be yielded, so it never stalls the rollout.  fp16-plane modes overflow above 65504."""

    def __init__(self, hint: str):
        self.hint, self.pending = hint, None

    def push(self, state, step: int):
        import torch
        self.check()
        self.pending = (torch.isfinite(state).all(), step)

    def take(self):
        """The pending (flag tensor, step) -- or None -- handed to a caller that reads it later, together with the state's copy to the host
        (core/models/utils.py: a rollout never waits for a step it does not look at); nothing is left for ``check``."""
        pending, self.pending = self.pending, None
        return pending

    def check(self):
        if self.pending is not None:
            flag, step = self.pending
            self.pending = None
            if not bool(flag.item()):
                raise FloatingPointError(f"non-finite values in the state after step {step}: {self.hint}")
