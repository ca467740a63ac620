"""Profile hygiene: a committed counter summary (profiles/<round>_<model>_pmc.json) describes the library that is in the tree, or bench.py
refuses to use it.  The stamp is written by tools/final_profiles.sh on the GPU box (sha256 of the .so it profiled, first 16 hex digits)."""
import importlib.util
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_tests", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("model", ["pangu", "sfno", "graphcast"])
def test_counter_summary_is_of_the_library_in_the_tree(model):
    b = _bench()
    f = ROOT / "profiles" / f"{b.PROFILE_ROUND}_{model}_pmc.json"
    have = b.lib_sha16(model)
    if have is None:
        pytest.skip("library not built")
    if not f.exists():
        assert b.pmc_summary(model) is None and b.pmc_stale(model) is None      # nothing committed: nothing claimed
        return
    stamp = json.loads(f.read_text())["stamp"]
    if not b._stamp_is_current(stamp, model):
        # kernels changed since the counters were taken (mid-round state): the summary must be DROPPED by bench.py, never paired with this
        # build's timings -- and the round's last GPU call (tools/final_profiles.sh) replaces it
        assert b.pmc_summary(model) is None and b.pmc_stale(model)["stale"] is True
        pytest.skip(f"{f.name} was taken on {stamp.split()[0]} / other sources; bench.py reports it as stale until tools/final_profiles.sh is re-run")
    assert b.pmc_summary(model) is not None and b.pmc_stale(model) is None


def test_a_summary_of_another_build_is_dropped_not_reported(tmp_path, monkeypatch):
    b = _bench()
    if b.lib_sha16("graphcast") is None:
        pytest.skip("library not built")
    fake = tmp_path / "profiles"
    fake.mkdir()
    (fake / f"{b.PROFILE_ROUND}_graphcast_pmc.json").write_text(json.dumps({"stamp": "0123456789abcdef libskyrim_graphcast.so, 2026-01-01T00:00Z",
                                                                           "total": {"steps": 1, "hbm_GB_per_step": 1.0}, "kernels": {}}))
    (tmp_path / "skyrim_amd").symlink_to(ROOT / "skyrim_amd")
    monkeypatch.setattr(b, "ROOT", tmp_path)
    assert b.pmc_summary("graphcast") is None
    st = b.pmc_stale("graphcast")
    assert st and st["stale"] is True and b.pmc_kernels("graphcast", "edge_update_kernel")["stale"] is True
