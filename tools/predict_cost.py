"""GPU box: the reference-shaped host path of one Pangu forecast step (bench.py's predict_inclusive) with one / several save workers and with the pwrite / mapped payload writers, and the two writers alone on a 573 MB payload in tmpfs over a range of worker counts.
    python tools/predict_cost.py [--writers-only]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("SKYRIM_SYNTHETIC_IC", "1")
os.environ.setdefault("SKYRIM_SYNTHETIC_WEIGHTS", "1")


def writers():
    from skyrim_amd import ncio
    n = 2 * 69 * 721 * 1440
    a = np.random.default_rng(0).standard_normal(n, dtype=np.float32)
    path = "/dev/shm/skyrim_writer_bench.bin" if os.path.isdir("/dev/shm") else "/tmp/skyrim_writer_bench.bin"
    out = {}
    for mapped in (True, False):
        for th in (8, 16, 32, 64):
            best = 1e9
            for _ in range(3):
                if os.path.exists(path):
                    os.unlink(path)
                with open(path, "wb") as f:
                    f.write(b"\0" * 4096)
                t = time.perf_counter()
                ncio._parallel_payload_write(path, 4096, a, threads=th, use_mmap=mapped)
                best = min(best, time.perf_counter() - t)
            out[f"{'mmap' if mapped else 'pwrite'}_{th}"] = round(1e3 * best, 1)
    ok = np.array_equal(np.fromfile(path, dtype=">f4", count=n, offset=4096), a)
    os.unlink(path)
    print(json.dumps({"writer_ms_573MB": out, "round_trip": bool(ok), "cpus": os.cpu_count()}))


def first_touch():
    """573 MB per file into tmpfs with plain pwrite, N files at once, three rounds each (files deleted in between): round 0 of a fresh
    box allocates guest memory nobody touched before, the later rounds reuse it."""
    from concurrent.futures import ThreadPoolExecutor
    n = 2 * 69 * 721 * 1440 * 4
    mv = memoryview(np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8))
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"

    def wr(i):
        fd = os.open(f"{base}/skyrim_touch_{i}.bin", os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        off = 0
        while off < n:
            off += os.pwrite(fd, mv[off:off + (64 << 20)], off)
        os.close(fd)
    out = {}
    for files in (1, 2, 4, 8, 16):
        for r in range(3):
            t = time.perf_counter()
            with ThreadPoolExecutor(files) as ex:
                list(ex.map(wr, range(files)))
            dt = time.perf_counter() - t
            out[f"files{files}_round{r}"] = {"ms_per_file": round(1e3 * dt / files, 1), "GBps": round(files * n / dt / 1e9, 1)}
            for i in range(files):
                os.unlink(f"{base}/skyrim_touch_{i}.bin")
    print(json.dumps({"first_touch_vs_reuse": out}))


def main():
    if "--touch" in sys.argv:
        first_touch()
    if "--no-writers" not in sys.argv:
        writers()
    if "--writers-only" in sys.argv:
        return
    import torch
    import bench
    from skyrim_amd.pangu.engine import DEFAULT_PRECISION
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic
    g = PanguGeometry(721, 1440)
    params = init_synthetic(g, 0)
    matrix = [tuple(c.split(":")) for c in os.environ.get("PREDICT_COST_MATRIX", "3:8 1:8 1:32 4:8").split()]
    for workers, threads in matrix:
        os.environ["SKYRIM_NC_THREADS"], os.environ["SKYRIM_SAVE_WORKERS"] = threads, workers
        r = bench.predict_inclusive(DEFAULT_PRECISION, g, params, torch.device("cuda", 0))
        print(json.dumps({"SKYRIM_NC_THREADS": threads, "SKYRIM_SAVE_WORKERS": workers, "no_save_ms": round(r["no_save"]["ms_per_step"], 2), "save_ms": round(r["save"]["ms_per_step"], 2), "save_first_rollout_ms": round(r["save_first_rollout"]["ms_per_step"], 2),
                          "io_counters": r["io_counters"]}))


if __name__ == "__main__":
    main()
