"""tests/_oracle_jobs.py: the background host processes the full-size GPU tests take their oracle results from."""
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import _oracle_jobs as J


def test_a_job_runs_in_its_own_process_and_fetch_returns_its_tensors():
    try:
        J.start(["selftest"])
        proc, path = J._started["selftest"]
        out = J.fetch("selftest")
        assert proc.returncode == 0 and Path(path).exists()
        assert torch.equal(out["x"], torch.arange(4.0)) and out["has_oracle"]
    finally:
        J.stop()
    assert not J._started and not Path(path).exists()


def test_fetch_computes_in_process_when_no_job_was_started():
    assert "selftest" not in J._started
    assert torch.equal(J.fetch("selftest")["x"], torch.arange(4.0))


def test_a_failing_job_raises_with_its_log(monkeypatch):
    monkeypatch.setitem(J.JOBS, "nope", lambda: None)            # known here, unknown to the child process
    try:
        J.start(["nope"])
        with pytest.raises(RuntimeError, match="oracle job nope failed"):
            J.fetch("nope")
    finally:
        J.stop()


def test_every_job_is_wanted_by_a_collected_gpu_test():
    for key, frags in J.WANTED_BY.items():
        assert key in J.JOBS
        for f in frags:
            mod, name = f.split("::")
            assert f"def {name}" in (Path(__file__).resolve().parent / mod).read_text()


def test_full_size_comparisons_are_collected_last_and_their_jobs_first():
    """conftest.pytest_collection_modifyitems: the full-size oracle comparisons go to the end of the run in their collected order, everything else
    keeps its order; the job list starts the longest (Pangu's 4-step rollout) first."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_conftest_under_test", Path(__file__).resolve().parent / "conftest.py")
    conf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conf)
    ids = ["tests/test_graphcast_gpu.py::test_a", "tests/test_graphcast_gpu.py::test_full_size_step_vs_oracle_per_channel", "tests/test_pangu_gpu.py::test_b",
           "tests/test_pangu_gpu.py::test_full_size_step_vs_oracle", "tests/test_pangu_gpu.py::test_c", "tests/test_sfno_gpu.py::test_full_size_step_vs_oracle_per_channel",
           "tests/test_sfno_gpu.py::test_d", "tests/test_host_api.py::test_full_size_is_not_a_gpu_file"]
    items = [type("Item", (), {"nodeid": i})() for i in ids]
    conf.pytest_collection_modifyitems(None, items)
    got = [i.nodeid for i in items]
    assert got == [ids[0], ids[2], ids[4], ids[6], ids[7], ids[1], ids[3], ids[5]]
    assert list(J.JOBS)[1] == "pangu_full_rollout4"                 # (index 0 is the self-test job)
    for key, frags in J.WANTED_BY.items():
        assert any(f in i for f in frags for i in ids), key
