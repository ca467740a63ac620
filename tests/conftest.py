import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


# Tests run on seeded synthetic initial conditions and random-init parameters; both are explicit opt-ins of the product
# (skyrim_amd/datasource.py, skyrim_amd/weights.py), which refuses to substitute them silently.
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def toy():
    """13 levels x 49 lat x 192 lon: every padding / crop / roll path of the full grid at toy size."""
    import torch
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    torch.manual_seed(0)
    g = PanguGeometry(49, 192)
    return g, init_synthetic(g, 0), synthetic_state(g, 0)
