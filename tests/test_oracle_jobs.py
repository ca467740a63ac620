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
