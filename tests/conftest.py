import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(1, str(ROOT / "tests"))          # tests/_oracle_jobs.py


# Tests run on seeded synthetic initial conditions and random-init parameters; both are explicit opt-ins of the product
# (skyrim_amd/datasource.py, skyrim_amd/weights.py), which refuses to substitute them silently.
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The full-size oracle comparisons run last: their oracle results come from host processes started when collection finishes
    (tests/_oracle_jobs.py, four minutes of host time), and every other GPU test runs while those compute."""
    last = [i for i in items if "_gpu.py::test_full_size" in i.nodeid]
    if last:
        items[:] = [i for i in items if "_gpu.py::test_full_size" not in i.nodeid] + last


def pytest_collection_finish(session):
    """The full-size oracle runs the selected GPU tests will ask for start now, in host processes of their own (tests/_oracle_jobs.py)."""
    ids = [item.nodeid for item in session.items]
    if not any("_gpu.py::test_full_size" in i for i in ids):
        return
    import torch
    if not torch.cuda.is_available() or os.environ.get("SKYRIM_TEST_ORACLE_JOBS", "1") == "0":
        return
    import _oracle_jobs
    _oracle_jobs.start([k for k, frags in _oracle_jobs.WANTED_BY.items() if any(f in i for f in frags for i in ids)])


def pytest_sessionfinish(session, exitstatus):
    mod = sys.modules.get("_oracle_jobs")
    if mod is not None:
        mod.stop()


@pytest.fixture(scope="session")
def toy():
    """13 levels x 49 lat x 192 lon: every padding / crop / roll path of the full grid at toy size."""
    import torch
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    torch.manual_seed(0)
    g = PanguGeometry(49, 192)
    return g, init_synthetic(g, 0), synthetic_state(g, 0)
