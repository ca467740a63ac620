"""GPU box: where the time of a saving rollout goes -- start / end of every predict_one_step on the main thread and of every file's write
(whole call, wait for the device-to-host copy, payload part) on the save threads, for one warmed full-size Pangu rollout.
    SKYRIM_SAVE_WORKERS=6 SKYRIM_NC_THREADS=4 python tools/save_timeline.py [n_steps]"""
import datetime
import json
import os
import shutil
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")
os.environ["SKYRIM_PANGU_CALIBRATION"] = "off"
os.environ["SKYRIM_PANGU_GUARD"] = "off"


def main():
    import torch
    from skyrim_amd import labeled, ncio
    from skyrim_amd.core.models import base
    from skyrim_amd.core.models.pangu import PanguModel
    from skyrim_amd.pangu.engine import DEFAULT_PRECISION
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    g = PanguGeometry(721, 1440)
    m = PanguModel(ic_source="synthetic", geom=g, params=init_synthetic(g, 0), precision=DEFAULT_PRECISION, device=torch.device("cuda", 0))
    ev, lock = [], threading.Lock()

    def wrap(obj, name, tag):
        fn = getattr(obj, name)

        def inner(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                with lock:
                    ev.append((tag, threading.current_thread().name, t, time.perf_counter()))
        setattr(obj, name, inner)
    wrap(ncio, "write_dataarray_netcdf3", "file")
    wrap(ncio, "_parallel_payload_write", "payload")
    wrap(base.GlobalModel, "_step_delivering", "step")
    values = labeled.DataArray.values.fget

    def timed_values(self):
        if self.__dict__.get("_ready") is None:
            return values(self)
        t = time.perf_counter()
        try:
            return values(self)
        finally:
            with lock:
                ev.append(("copy_wait", threading.current_thread().name, t, time.perf_counter()))
    labeled.DataArray.values = property(timed_values, labeled.DataArray.values.fset)

    real_empty = torch.empty

    def timed_empty(*a, **k):
        if not k.get("pin_memory"):
            return real_empty(*a, **k)
        t = time.perf_counter()
        try:
            return real_empty(*a, **k)
        finally:
            with lock:
                ev.append(("pin_alloc", threading.current_thread().name, t, time.perf_counter()))
    torch.empty = timed_empty
    if os.environ.get("PRIME_PINNED"):
        for entries in (1, 2):             # an intermediate step's image holds one state (the other is borrowed), the last step's arrays two
            blocks = [real_empty((entries, 69, 721, 1440), dtype=torch.float32, pin_memory=True) for _ in range(int(os.environ["PRIME_PINNED"]))]
            del blocks
    t0 = datetime.datetime(2024, 1, 1)
    d = tempfile.mkdtemp(prefix="skyrim_tl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        pred, _ = m.rollout(t0, n_steps=3, save=False)
        pred, paths = m.rollout(t0, n_steps=n_steps, save=True, save_config={"output_dir": d}, initial_condition=pred)
        for p in paths:
            os.unlink(p)
        ev.clear()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        pred, paths = m.rollout(t0, n_steps=n_steps, save=True, save_config={"output_dir": d}, initial_condition=pred)
        pred.values
        total = time.perf_counter() - t_begin
    finally:
        shutil.rmtree(d, ignore_errors=True)
    rows = sorted(ev, key=lambda e: e[2])
    for tag, th, a, b in rows:
        print(f"{tag:10s} {th:18s} {1e3 * (a - t_begin):8.1f} -> {1e3 * (b - t_begin):8.1f}   ({1e3 * (b - a):6.1f} ms)")
    by = {}
    for tag, _, a, b in rows:
        by.setdefault(tag, []).append(1e3 * (b - a))
    print(json.dumps({"ms_per_step": round(1e3 * total / n_steps, 2), "workers": os.environ.get("SKYRIM_SAVE_WORKERS"), "nc_threads": os.environ.get("SKYRIM_NC_THREADS"),
                      "mean_ms": {k: round(sum(v) / len(v), 1) for k, v in by.items()}, "cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
