"""Full-size oracle runs in background host processes (test infrastructure, like oracle/ itself).

One 721x1440 step of a CPU oracle costs one to two minutes of host time and the GPU tests need six of them (Pangu's 24-h rollout = 4,
GraphCast 1, SFNO 1); run one after another inside the tests they were 470 s of an 890 s suite during which the GPU sat idle.  conftest.py
starts them in pytest_configure of a `-m gpu` run on a GPU box (Pangu's first: it is the longest), and stops those no selected test asks for
when collection finishes; each is `python tests/_oracle_jobs.py <key> <file>`, computes
from the same seeds the test uses and saves its tensors; `fetch(key)` waits for the file (or computes in-process when no job was
started, e.g. when a test module is imported by hand).  The checker is unchanged -- same oracle functions, same inputs -- only where and
when it runs."""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
_started = {}          # key -> (Popen, path)
_dir = None


def _pangu_full_rollout4(path=None):
    """The oracle's 24-h rollout; with a path every step is saved as soon as it exists (`<path>.step<k>`), so that the one-step comparison
    does not wait for the other three."""
    import torch
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    g = PanguGeometry(721, 1440)
    params, x, outs = init_synthetic(g, 0), synthetic_state(g, 0), []
    for k in range(4):
        x = O.forward(params, x)                  # = O.rollout(params, x, 4), step by step
        outs.append(x)
        if path is not None:
            torch.save(x, f"{path}.step{k}.part")
            os.replace(f"{path}.step{k}.part", f"{path}.step{k}")
    return {"rollout": outs}


def _graphcast_full_step():
    import torch
    from oracle import graphcast_graph as OG
    from oracle import graphcast_oracle as O
    from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states
    cfg = GraphcastConfig()
    x0, x1 = synthetic_states(cfg, 0)
    with torch.no_grad():
        return {"ref": O.forward(init_synthetic(cfg, 0), OG.build(cfg.n_lat, cfg.n_lon, cfg.splits), x0, x1, forcings(cfg, 1000.0))}


def _sfno_full_step():
    import torch
    from oracle import sfno_oracle as O
    from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic, synthetic_state
    cfg = SfnoConfig()
    with torch.no_grad():
        return {"ref": O.forward(init_synthetic(cfg, 0), synthetic_state(cfg, 0), cfg)}


def _selftest():
    import torch
    from oracle import pangu_oracle as O          # the job's process sees the repository root
    return {"x": torch.arange(4.0), "has_oracle": hasattr(O, "rollout")}


JOBS = {"selftest": _selftest, "pangu_full_rollout4": _pangu_full_rollout4, "graphcast_full_step": _graphcast_full_step, "sfno_full_step": _sfno_full_step}
# which collected tests need which job (node id fragments)
WANTED_BY = {"pangu_full_rollout4": ("test_pangu_gpu.py::test_full_size",), "graphcast_full_step": ("test_graphcast_gpu.py::test_full_size_step",),
             "sfno_full_step": ("test_sfno_gpu.py::test_full_size_step",)}


def start(keys):
    """Start one host process per key (the longest first).  Each gets a share of the host's threads: the oracles stop scaling well
    before 128 threads, and the tests that run meanwhile need a few themselves."""
    global _dir
    if not keys:
        return
    _dir = tempfile.mkdtemp(prefix="skyrim_oracle_")
    cores = os.cpu_count() or 8
    # Pangu's four steps are the critical path (about a minute each on 128 threads, not much less on 64): half the host; the two single steps share a third
    share = {k: max(8, cores // 2 if k == "pangu_full_rollout4" else cores // 6) for k in keys}
    for key in sorted(keys, key=lambda k: list(JOBS).index(k)):
        if key in _started:
            continue
        path = os.path.join(_dir, key + ".pt")
        log = open(os.path.join(_dir, key + ".log"), "w")
        env = dict(os.environ, OMP_NUM_THREADS=str(share[key]), MKL_NUM_THREADS=str(share[key]), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        _started[key] = (subprocess.Popen([sys.executable, str(Path(__file__).resolve()), key, path], cwd=str(ROOT), env=env, stdout=log, stderr=log), path)


def _failed(key, rc, path):
    return RuntimeError(f"oracle job {key} failed (rc {rc}): " + Path(path[:-3] + ".log").read_text()[-2000:])


def fetch(key):
    import torch
    if key in _started:
        proc, path = _started[key]
        rc = proc.wait()
        if rc != 0:
            raise _failed(key, rc, path)
        return torch.load(path)
    return JOBS[key]()


class PanguRollout:
    """`rollout[k]` = step k of the oracle's full-size 24-h rollout, waiting only for that step of the running job."""

    def __init__(self):
        self._have = {}

    def __getitem__(self, k):
        import time
        import torch
        if k not in self._have:
            if "pangu_full_rollout4" in _started:
                proc, path = _started["pangu_full_rollout4"]
                part = f"{path}.step{k}"
                while not os.path.exists(part):
                    rc = proc.poll()
                    if rc is not None and not os.path.exists(part):
                        raise _failed("pangu_full_rollout4", rc, path)
                    time.sleep(0.5)
                self._have[k] = torch.load(part)
            else:
                for i, x in enumerate(JOBS["pangu_full_rollout4"]()["rollout"]):
                    self._have[i] = x
        return self._have[k]


def stop(keys=None):
    """Session end (or: jobs no selected test asks for): our own children only, by the handles we hold."""
    for key in list(_started) if keys is None else [k for k in keys if k in _started]:
        proc, _ = _started.pop(key)
        if proc.poll() is None:
            proc.kill()
            proc.wait()
    if keys is not None:
        return
    if _dir is not None:
        import shutil
        shutil.rmtree(_dir, ignore_errors=True)


if __name__ == "__main__":
    import torch
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
    os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")
    out = JOBS[sys.argv[1]](sys.argv[2]) if sys.argv[1] == "pangu_full_rollout4" else JOBS[sys.argv[1]]()
    torch.save(out, sys.argv[2] + ".part")
    os.replace(sys.argv[2] + ".part", sys.argv[2])
