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
    # The oracles the tests call in-process work on SMALL grids (49x192, 97x192, 33x64 ...): on the GPU box's 128 host threads their PyTorch-CPU
    # ops are slower than on 8 (measured: the SFNO 40-step rollout test 72 s with 128 threads, 9 s with 32, 4.5 s with 8 -- thread fan-out per tiny
    # tensor op).  The full-size oracle runs are separate processes with their own thread counts (tests/_oracle_jobs.py, tests/golden/make_full_size.py).
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    # a GPU run: the full-size oracle jobs start NOW (tests/_oracle_jobs.py; Pangu's rollout, the longest, first) -- they compute in host
    # processes while pytest collects and the small-grid parity tests run
    expr = getattr(config.option, "markexpr", "") or ""
    # (only with SKYRIM_TEST_LIVE_ORACLE=1: by default the full-size comparisons read the committed golden vectors of the same oracles,
    # tests/golden/full_*.npz -- tests/_golden_full.py)
    if "gpu" in expr and "not gpu" not in expr and os.environ.get("SKYRIM_TEST_LIVE_ORACLE") == "1" and not config.option.collectonly:
        import torch
        if torch.cuda.is_available():
            import _oracle_jobs
            _oracle_jobs.start(list(_oracle_jobs.WANTED_BY))


# Collection order of the GPU suite (the driver runs `pytest -x`: whatever fails first hides the rest, so the parity evidence goes first):
#   0  small-grid oracle parity of the three default modes (the native library test first)
#   1  the full-size comparisons with the oracles (BASELINE.json's sizes; each waits for its host job -- SFNO's is ready first)
#   2  everything else, in file order
_TIER0 = ("test_pangu_gpu.py::test_native_library_is_the_path_that_runs[{d}]", "test_pangu_gpu.py::test_full_step_per_channel[{d}]",
          "test_pangu_gpu.py::test_step_matches_golden_fixture[{d}]", "test_pangu_gpu.py::test_earth_specific_block[{d}-",
          "test_sfno_gpu.py::test_step_vs_oracle_per_channel[", "test_sfno_gpu.py::test_step_matches_golden_fixture",
          "test_graphcast_gpu.py::test_step_vs_oracle[", "test_graphcast_gpu.py::test_matches_golden_fixture")
_TIER1 = ("test_pangu_gpu.py::test_full_size_step_vs_oracle", "test_sfno_gpu.py::test_full_size", "test_graphcast_gpu.py::test_full_size",
          "test_pangu_gpu.py::test_full_size_24h_rollout", "test_pangu_gpu.py::test_full_size_term_plans", "test_pangu_gpu.py::test_full_size")


def _tier(nodeid, default_mode):
    for t, frags in enumerate((_TIER0, _TIER1)):
        for rank, f in enumerate(frags):
            if f.format(d=default_mode) in nodeid:
                return t, rank
    return 2, 0


def pytest_collection_modifyitems(config, items):
    if not any("_gpu.py::" in i.nodeid for i in items):
        return
    from skyrim_amd.pangu.engine import DEFAULT_PRECISION
    order = {id(i): (*_tier(i.nodeid, DEFAULT_PRECISION), n) for n, i in enumerate(items)}
    items.sort(key=lambda i: order[id(i)])


def pytest_collection_finish(session):
    """Oracle jobs that no selected test asks for stop here."""
    mod = sys.modules.get("_oracle_jobs")
    if mod is None:
        return
    ids = [item.nodeid for item in session.items]
    mod.stop([k for k, frags in mod.WANTED_BY.items() if not any(f in i for f in frags for i in ids)])


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
