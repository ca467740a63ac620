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


def test_parity_evidence_is_collected_first_and_the_longest_job_starts_first():
    """conftest.pytest_collection_modifyitems: small-grid oracle parity of the default modes, then the full-size comparisons (Pangu's single step
    first: its oracle step is ready a minute into the run), then everything else in its collected order -- under `pytest -x` a failing
    host-API test must not hide the headline comparisons.  The job list starts the longest (Pangu's 4-step rollout) first."""
    import importlib.util
    from skyrim_amd.pangu.engine import DEFAULT_PRECISION as D
    spec = importlib.util.spec_from_file_location("_conftest_under_test", Path(__file__).resolve().parent / "conftest.py")
    conf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conf)
    ids = ["tests/test_graphcast_gpu.py::test_a", "tests/test_graphcast_gpu.py::test_full_size_step_vs_oracle_per_channel", "tests/test_pangu_gpu.py::test_b",
           "tests/test_pangu_gpu.py::test_full_size_24h_rollout_vs_oracle", "tests/test_pangu_gpu.py::test_full_size_step_vs_oracle",
           f"tests/test_pangu_gpu.py::test_full_step_per_channel[{D}]", "tests/test_pangu_gpu.py::test_full_step_per_channel[bf16x3]",
           "tests/test_sfno_gpu.py::test_full_size_step_vs_oracle_per_channel", "tests/test_sfno_gpu.py::test_step_vs_oracle_per_channel[tiny]",
           "tests/test_host_api.py::test_full_size_is_not_a_gpu_file", f"tests/test_pangu_gpu.py::test_native_library_is_the_path_that_runs[{D}]"]
    items = [type("Item", (), {"nodeid": i})() for i in ids]
    conf.pytest_collection_modifyitems(None, items)
    got = [i.nodeid for i in items]
    assert got == [ids[10], ids[5], ids[8], ids[4], ids[7], ids[1], ids[3], ids[0], ids[2], ids[6], ids[9]]
    only_cpu = [type("Item", (), {"nodeid": i})() for i in ("tests/test_host_api.py::b", "tests/test_abi.py::a")]
    conf.pytest_collection_modifyitems(None, only_cpu)
    assert [i.nodeid for i in only_cpu] == ["tests/test_host_api.py::b", "tests/test_abi.py::a"]          # a CPU run keeps its order
    assert list(J.JOBS)[1] == "pangu_full_rollout4"                 # (index 0 is the self-test job)
    for key, frags in J.WANTED_BY.items():
        assert any(f in i for f in frags for i in ids), key


def test_pangu_rollout_steps_arrive_one_by_one(monkeypatch, tmp_path):
    """PanguRollout[k] waits for step k's file of the running job only (the one-step comparison starts a minute into the run)."""
    import subprocess
    import time
    path = str(tmp_path / "pangu_full_rollout4.pt")
    code = ("import time, torch, os; p = %r\n"
            "for k in range(2):\n"
            "    torch.save(torch.full((2,), float(k)), p + f'.step{k}.part'); os.replace(p + f'.step{k}.part', p + f'.step{k}'); time.sleep(1.0)\n") % path
    proc = subprocess.Popen([sys.executable, "-c", code])
    monkeypatch.setitem(J._started, "pangu_full_rollout4", (proc, path))
    try:
        r = J.PanguRollout()
        t0 = time.time()
        assert r[0].tolist() == [0.0, 0.0] and r[1].tolist() == [1.0, 1.0]
        assert r[0] is r[0] and time.time() - t0 < 30
        proc.wait()
        with pytest.raises(RuntimeError, match="oracle job pangu_full_rollout4 failed"):
            (tmp_path / "pangu_full_rollout4.log").write_text("no such step")
            r[2]
    finally:
        proc.kill()
        proc.wait()
