#!/usr/bin/env python
"""Headline benchmark: Pangu 6-h forecast steps/s on the 721x1440x69 state (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one skpangu_step() = one 6-h forward of the Pangu network on a synthetic state that is
already resident in HBM (autoregressive, in place).  N > 1: one process per GPU, one ensemble member
per rank (weak scaling, members are independent); the timed region ends with the RCCL reduction that
forms the ensemble mean/spread of the final step.  Prints ONE JSON line on rank 0.

At N = 1 the same line also carries, under "models", a short driver-timed run of the other two rows of the hot path (FourCastNet
v2-small and GraphCast at 721x1440: ms per step, the roofline of the dominant stage, the committed profile it is read against);
``--model sfno|graphcast`` makes one of them the headline line instead.  Counter-derived fields (HBM traffic, MFMA-busy) are never
measured in this run: they are read from the committed ``profiles/r03_<model>_pmc.json`` and are dropped -- with a note -- when that
summary holds no kernel of the name the live run's dominant stage launches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic work of one 721x1440 step
F_ALG_STEP = 8.42e12
PEAK_MFMA_BF16 = 2.5e15      # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


# precision modes: (arithmetic, note on the 1e-3 parity bar with the per-channel error observed on the toy / full grid)
_ATT = "; window attention (Q, K, V, P) single-term fp16; fp32 accumulate, LayerNorm, softmax, GELU"
MODE_NOTES = {
    "f16x1m": ("fp16 MFMA; activations as hi/lo fp16 planes; proj / fc1 / fc2 of every block with the weights as ONE fp16 plane; layers 2 / 3 (C = 384: "
               "12 of 16 blocks) read the activation operands' hi plane only -- ONE term, A_hi W -- and their QKV is one term; layers 1 / 4: two terms "
               "(A_hi W + A_lo W), QKV with hi/lo weights; weights rounded with error feedback against the operands (term plan 0x66F)" + _ATT,
               "default since round 5: 2.5 .. 2.8e-4 over the full-size 24-h rollout"),
    "f16x2m": ("f16x1m with two terms (hi/lo activation operands) in layers 2 / 3 as well (term plan 0x6F)" + _ATT,
               "round 4's default: 1.4 .. 1.7e-4 over the full-size 24-h rollout"),
    "f16x2c": ("f16x2m with layers 1 / 4 at three terms (hi/lo weights; term plan 0x66, calibrated)" + _ATT, "meets the bar with 3x margin"),
    "f16x3q": ("fp16 MFMA on hi/lo fp16 planes (22-bit operands): 3 terms per GEMM, QKV 2 terms (stream hi plane only)" + _ATT,
               "meets the bar (~1e-4)"),
    "f16x3": ("fp16 MFMA on hi/lo fp16 planes, 3 terms per GEMM" + _ATT, "meets the bar (~8e-5)"),
    "bf16x3": ("bf16 MFMA on hi/lo bf16 planes (16-bit operands, fp32 range), 3 terms per GEMM" + _ATT,
               "wide-range alternative; meets the bar (~8e-5)"),
}


def cpu_baseline(params, geom, x, keep=None):
    """CPU restatement (oracle/, kind 'port') timed on this box's host cores: ONE full step of the same workload, no scaling.
    ``keep`` (a dict): receives the step's output under "y" -- the full-size parity figure of the line is read against it."""
    from oracle import pangu_oracle as O
    # the thread count the restatement is FASTEST with, not the largest available: measured on the 128-thread GPU box (tools/cpu_scale.py) one
    # full-size step takes 62.5 s on 128 threads, 44.6 on 64, 42.8 on 32, 43.3 on 16 -- the baseline runs on 32 (SKYRIM_BENCH_CPU_THREADS overrides)
    have = torch.get_num_threads()
    cores = int(os.environ.get("SKYRIM_BENCH_CPU_THREADS", 0)) or min(32, have)
    torch.set_num_threads(cores)
    try:
        t0 = time.time()
        with torch.no_grad():
            y = O.forward(params, x)
        dt = time.time() - t0
    finally:
        torch.set_num_threads(have)
    if keep is not None:
        keep["y"] = y
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"ONE full {geom.n_lat}x{geom.n_lon} 6-h step of the PyTorch-CPU fp32 restatement (oracle/pangu_oracle.py) in {dt:.1f} s on {cores} "
                      f"threads (of {have}: more are slower); no scaling; not the reference's ONNX graph (onnxruntime and the weights are not obtainable here); finite={bool(torch.isfinite(y).all())}",
            "s_per_step": dt}


class small_oracle_threads:
    """A small-grid oracle call on a many-core host: PyTorch-CPU fans every tiny tensor op out over all threads and gets SLOWER (measured on the
    128-thread GPU box: a 97x192 SFNO oracle step 1.8 s with 128 threads, 0.1 s with 8).  The full-size CPU baseline keeps every thread."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(min(8, self.n))

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def toy_parity(precision, rounding="default"):
    """Engine vs oracle on the 13x49x192 toy grid (seconds)."""
    from oracle import pangu_oracle as O
    from skyrim_amd.pangu.engine import PanguEngine
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state
    g = PanguGeometry(49, 192)
    p = init_synthetic(g, 0)
    x = synthetic_state(g, 0)
    eng = PanguEngine(g, precision)
    eng.load_params(p, rounding=rounding)
    with small_oracle_threads():
        y, ref = eng.step(x.to(eng.device)).cpu(), O.forward(p, x)
    return {"grid": "49x192", "max_rel_err": O.per_channel_rel_err(y, ref).max().item(), "bar": 1e-3, "rounding": eng.rounding,
            "max_sigma_err": O.per_channel_sigma_err(y, ref, p["norm.std"]).max().item()}     # the same difference in units of the channel's sigma


PROFILE_ROUND = "r06"


def lib_sha16(model: str) -> str | None:
    """First 16 hex digits of sha256(skyrim_amd/lib/libskyrim_<model>.so) -- what tools/final_profiles.sh stamps its counter summaries with."""
    import hashlib
    f = ROOT / "skyrim_amd" / "lib" / f"libskyrim_{model}.so"
    try:
        return hashlib.sha256(f.read_bytes()).hexdigest()[:16]
    except OSError:
        return None


def src_sha16() -> str:
    """First 16 hex digits of sha256 over the kernel sources (skyrim_amd/csrc/*, include/*.h, names + contents).  hipcc does not reproduce a
    shared object bit for bit (a from-scratch rebuild of identical sources hashes differently), so a counter summary is also accepted when
    it names THESE sources: same kernels, another link."""
    import hashlib
    h = hashlib.sha256()
    files = sorted((ROOT / "skyrim_amd" / "csrc").glob("*")) + sorted((ROOT / "include").glob("*.h"))
    for f in files:
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _stamp_is_current(stamp: str, model: str) -> bool:
    """A counter summary belongs to this run when it names the library that is loaded -- or the kernel sources, and then only if the loaded
    library is provably a build OF those sources: the default in-tree path (no SKYRIM_<MODEL>_LIB override) and not older than any source
    file.  Otherwise counters of one binary could be paired with timings of another (a stale or variant build)."""
    have = lib_sha16(model)
    if have and stamp.startswith(have):
        return True
    if f"src {src_sha16()}" not in stamp or os.environ.get(f"SKYRIM_{model.upper()}_LIB"):
        return False
    lib = ROOT / "skyrim_amd" / "lib" / f"libskyrim_{model}.so"
    srcs = [f for f in list((ROOT / "skyrim_amd" / "csrc").glob("*")) + list((ROOT / "include").glob("*.h")) if f.is_file()]
    try:
        return lib.stat().st_mtime >= max(f.stat().st_mtime for f in srcs)
    except OSError:
        return False


def pmc_summary(model: str):
    """profiles/<round>_<model>_pmc.json (tools/pmc_collect.sh + tools/pmc_summary.py: rocprofv3 --pmc passes, FETCH_SIZE doubled per
    MI355X_MICROARCH.md 'HBM'), or None.  A COMMITTED profile, not a measurement of this run: it is only used when its stamp names the
    library this process loads (or the kernel sources it was built from: ``src_sha16``) -- counters of another build are dropped (``pmc_stale`` says so), never paired with fresh timings."""
    f = ROOT / "profiles" / f"{PROFILE_ROUND}_{model}_pmc.json"
    try:
        d = json.loads(f.read_text())
    except Exception:
        return None
    if not _stamp_is_current(str(d.get("stamp", "")), model):
        return None
    return d


def pmc_stale(model: str):
    """{'stale': True, ...} when a counter summary exists but was taken on another build of the library; None otherwise."""
    f = ROOT / "profiles" / f"{PROFILE_ROUND}_{model}_pmc.json"
    try:
        stamp = str(json.loads(f.read_text()).get("stamp", ""))
    except Exception:
        return None
    have = lib_sha16(model)
    if _stamp_is_current(stamp, model):
        return None
    return {"stale": True, "profiled_at": stamp, "library_sha16": have, "note": "counter summary dropped: it describes another build of the library"}


def pmc_kernels(model: str, *needles: str):
    """Counters of the kernels whose (demangled) name contains every needle, summed per STEP: HBM bytes, kernel time under the
    counters, MFMA-busy share (time-weighted).  When the committed summary has no such kernel (renamed or retuned since the profile
    was taken) the counters are NOT reported: the entry says so instead of pairing stale counters with fresh timings."""
    d = pmc_summary(model)
    src = f"profiles/{PROFILE_ROUND}_{model}_pmc.json"
    if not d:
        return pmc_stale(model) or {"note": f"no committed counter summary ({src})"}
    steps = d["total"]["steps"]
    rows = [e for k, e in d["kernels"].items() if all(n in k for n in needles)]
    if not rows:
        return {"note": f"{src} holds no kernel matching {list(needles)}: counters dropped (profile older than the kernel)"}
    t = sum(e["avg_us"] * e["calls"] for e in rows)
    return {"hbm_bytes_per_step": 1e9 * sum(e["hbm_GB"] * e["calls"] for e in rows) / steps,
            "hbm_bytes_per_launch": 1e9 * sum(e["hbm_GB"] * e["calls"] for e in rows) / sum(e["calls"] for e in rows),
            "ms_per_step_under_pmc": t / steps / 1e3,
            "mfma_busy_pct": sum((e["mfma_busy_pct"] or 0.0) * e["avg_us"] * e["calls"] for e in rows) / t if t else None,
            "source": src, "kind": "committed profile, not measured in this run",
            "profiled_at": d.get("stamp")}


# bench stage -> what identifies its kernel in the counter summary
PANGU_STAGE_KERNEL = {"mlp_r0": ("fused_mlp_kernel", "MlpShape<192"), "mlp_r1": ("fused_mlp_kernel", "MlpShape<384"),
                      "proj_mlp_r0": ("proj_mlp", "Shape<192"), "proj_mlp_r1": ("proj_mlp", "Shape<384"),
                      "qkv_r0": ("rt_qkv_kernel", "QkvShape<192"), "qkv_r1": ("rt_qkv_kernel", "QkvShape<384"),
                      "attn_r0": ("attention", "QaShape<192"), "attn_r1": ("attention", "QaShape<384"),      # qkv_attention_kernel: QKV + attention in one launch
                      "proj_r0": ("gemm_dma_kernel", "256x192", "EpLayerNorm", "RowMapIndexed"), "proj_r1": ("gemm_dma_kernel", "128x384", "EpLayerNorm")}


def quick_mode(precision, geom, params, x_host, dev, steps=3, mlp="fused", rounding="default"):
    """Short run of another precision mode (same workload) for the 'modes' table: TIMING only, so the load-time calibration (which changes
    weights' last bits and biases, never the kernels or their time; 13 s per engine at full size with compensated rounding) is off."""
    from skyrim_amd.pangu.engine import PanguEngine
    eng = PanguEngine(geom, precision, dev, mlp=mlp)
    eng.load_params(params, calibration="off", rounding=rounding, guard=False)
    x = x_host.to(dev)
    eng.step(x, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(x, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del eng
    return {"ms_per_step": 1e3 * dt, "steps_per_s": 1.0 / dt}


def api_rollout_rate(precision, geom, params, x_host, dev, n=6):
    """PCIe-inclusive rate through the reference-shaped API: run_basic_inference (core/models/utils.py) yields every
    step to the host as the reference does (286 MB per step), here through a pinned buffer on a copy stream."""
    import datetime
    from skyrim_amd.core.models.utils import run_basic_inference
    from skyrim_amd.labeled import DataArray
    from skyrim_amd.pangu.timeloop import PanguTimeLoop
    loop = PanguTimeLoop(params, geom, precision, dev, calibration="off", guard=False)      # a rate, not a forecast: no load-time calibration / guard
    t0 = datetime.datetime(2024, 1, 1)
    x = DataArray(x_host.numpy()[None], dims=["time", "channel", "lat", "lon"],
                  coords=dict(time=[t0], channel=loop.in_channel_names, lat=loop.grid.lat, lon=loop.grid.lon))
    run_basic_inference(loop, n, None, t0, x=x)            # warm: page-locks the result buffer once (torch caches it)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = run_basic_inference(loop, n, None, t0, x=x)
    dt = (time.perf_counter() - t) / n
    del loop
    return {"steps_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "steps": n,
            "note": "run_basic_inference: every 6-h state copied to the host (pinned buffer, copy stream overlapped with the next "
                    "step); includes the H2D of the initial state; the page-locked result buffer is warm; finite="
                    + str(bool(torch.isfinite(torch.from_numpy(out.values[-1])).all()))}


def members_on_streams(precision, geom, params, x_host, dev, n_members=2, steps=6):
    """Ensemble throughput on ONE GPU (BASELINE configs[4] runs 6-7 members per GPU): ``n_members`` members, each with its own engine
    workspace, advanced on separate HIP streams so that one member's bandwidth-bound kernels (QKV, attention, row gathers / stores) can
    run beside the other's MFMA-bound ones -- against the same members advanced back to back on one stream."""
    from skyrim_amd.pangu.engine import PanguEngine
    engs, xs = [], []
    for m in range(n_members):
        e = PanguEngine(geom, precision, dev)
        e.load_params(params, calibration="off", guard=False)           # throughput only
        engs.append(e)
        xs.append(x_host.to(dev) + 1e-3 * m)
    streams = [torch.cuda.Stream(dev) for _ in range(n_members)]

    def run(concurrent):
        for e, x in zip(engs, xs):
            e.step(x, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for m in range(n_members):
                if concurrent:
                    with torch.cuda.stream(streams[m]):
                        engs[m].step(xs[m], xs[m])
                else:
                    engs[m].step(xs[m], xs[m])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (steps * n_members)
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    serial, conc = run(False), run(True)
    return {"members": n_members, "member_steps_per_s_one_stream": 1.0 / serial, "member_steps_per_s_streams": 1.0 / conc,
            "ms_per_member_step_one_stream": 1e3 * serial, "ms_per_member_step_streams": 1e3 * conc, "gain": serial / conc,
            "finite": bool(all(torch.isfinite(x).all() for x in xs))}


def predict_inclusive(precision, geom, params, dev, n_steps=8):
    """What ``Skyrim('pangu').predict(..., save=...)`` costs per step end to end: ``GlobalModel.rollout`` (the call ``predict`` makes,
    core/skyrim.py) through ``predict_one_step`` / ``run_basic_inference`` -- every step delivered to the host as a (2, 69, 721, 1440) array
    like the reference's (573 MB), the state itself staying in HBM between steps -- once without saving and once with the per-step netCDF
    files written to tmpfs by the save thread.  The rollouts continue from a delivered prediction (``initial_condition=``), so the fetch
    of the initial condition (a data-source cost: 3-11 s in the reference's own notebook) is not in the figure."""
    import datetime
    import shutil
    import tempfile
    from skyrim_amd.core.models.pangu import PanguModel
    keep = {k: os.environ.get(k) for k in ("SKYRIM_PANGU_CALIBRATION", "SKYRIM_PANGU_GUARD")}
    os.environ["SKYRIM_PANGU_CALIBRATION"] = "off"         # a cost figure, not a forecast: skip the 13 s of load-time calibration
    os.environ["SKYRIM_PANGU_GUARD"] = "off"               # ... and the guard that would judge the uncalibrated plan
    try:
        m = PanguModel(ic_source="synthetic", geom=geom, params=params, precision=precision, device=dev)
    finally:
        for k, v in keep.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    t0 = datetime.datetime(2024, 1, 1)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    out = {}
    pred, _ = m.rollout(t0, n_steps=3, save=False)                                                   # warm: IC, pinned buffers
    from skyrim_amd.core.models.base import SAVE_WORKERS
    in_flight = int(os.environ.get("SKYRIM_SAVE_WORKERS", SAVE_WORKERS)) + 3
    for save in (False, True):
        d = tempfile.mkdtemp(prefix="skyrim_bench_", dir=base)
        try:
            # Warm-up = the timed rollout itself, run once before: the pinned delivery buffers (one 573 MB block per prediction in flight,
            # up to SAVE_WORKERS + 2 of them) and the memory the page cache takes for the files have then been touched once.  On a fresh
            # VM the FIRST touch of guest memory is what a save rollout measures otherwise (~3 - 7 GB/s however many threads write,
            # docs/experiments.md A6.5); its figure is kept as ``save_first_rollout``.  The warm-up's files are deleted before the timed run.
            torch.cuda.synchronize()
            t = time.perf_counter()
            pred, paths = m.rollout(t0, n_steps=n_steps if save else 2, save=save, save_config={"output_dir": d}, initial_condition=pred)
            pred.values
            if save:
                out["save_first_rollout"] = {"ms_per_step": 1e3 * (time.perf_counter() - t) / n_steps, "files": len(paths)}
                for p in paths:
                    os.unlink(p)
                # ... and torch's pinned-block cache holds one block per prediction that can be in flight (the first rollout stalls on its
                # own allocations, so it never has that many at once: tools/save_timeline.py shows a 45 - 90 ms hipHostMalloc inside a
                # later rollout's step otherwise)
                for entries in (1, pred.shape[0]):       # an intermediate step's image holds the new state only, the last step's arrays the pair
                    blocks = [torch.empty((entries,) + tuple(pred.shape[1:]), dtype=torch.float32, pin_memory=True) for _ in range(in_flight)]
                    del blocks
            torch.cuda.synchronize()
            t = time.perf_counter()
            pred, paths = m.rollout(t0, n_steps=n_steps, save=save, save_config={"output_dir": d}, initial_condition=pred)
            pred.values                                  # the last prediction's copy to the host is part of the rollout (DataArray.values waits for it)
            dt = (time.perf_counter() - t) / n_steps
            out["save" if save else "no_save"] = {"ms_per_step": 1e3 * dt, "steps_per_s": 1.0 / dt, "files": len(paths),
                                                   "bytes_per_file": os.path.getsize(paths[0]) if paths else 0}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out["io_counters"] = dict(m.model.io_counters)
    from skyrim_amd import deliver
    out["big_endian_image"] = deliver.enabled()
    out["note"] = (f"GlobalModel.rollout(n_steps={n_steps}, initial_condition=<the previous prediction>) through the reference-shaped API: the state stays in "
                   "HBM (io_counters: one upload, the initial condition of the warm-up) and the host never waits for the GPU between two steps (a step's "
                   "non-finite flag is read with its copy); no_save: only the last prediction's (t, t + 6 h) pair is copied to the host -- what the "
                   "call returns; "
                   f"save: one netCDF-3 file of 573 MB per step ({'tmpfs' if base else 'tmp dir'}); the pair is byte-swapped in HBM (skio_bswap32) and it is that "
                   "big-endian image that crosses to pinned memory (intermediate steps: instead of the native copy; last step: both), so the save threads only "
                   "pwrite while the next steps run (SKYRIM_SAVE_BE=0: native copy, swapped by the save threads). save = the second saving rollout of the "
                   "process with torch's pinned-block cache holding a block per prediction in flight; save_first_rollout = the first one (it allocates them)")
    return out


def run_sfno(args, rank, local_rank, world, dist):
    """The same contract for the SFNO row (BASELINE.json configs[2]): one step = one 6-h forward of FourCastNet v2-small on a
    synthetic 73-channel 721x1440 state resident in HBM; N > 1 = one member per rank + the closing ensemble reduction."""
    from skyrim_amd.pangu.ensemble import ensemble_mean_spread
    from skyrim_amd.sfno.engine import SfnoEngine
    from skyrim_amd.sfno.spec import SfnoConfig, flops_per_step, init_synthetic, synthetic_state
    cfg = SfnoConfig(n_lat=args.n_lat, n_lon=args.n_lon)
    params = init_synthetic(cfg, 0)
    dev = torch.device("cuda", local_rank)
    eng = SfnoEngine(cfg, dev)
    eng.load_params(params)
    x_host = synthetic_state(cfg, rank if world > 1 else 0)
    x = x_host.to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step(x, x)
    if world > 1:
        ensemble_mean_spread([x], world)
    eng.profiling = True
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step(x, x)
    if world > 1:
        ensemble_mean_spread([x], world)
    sync()
    elapsed = time.perf_counter() - t0
    stats = eng.profile_read()
    eng.profiling = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    finite = bool(torch.isfinite(x).all().item())
    if rank != 0:
        return
    f_step = flops_per_step(cfg)
    dom = max(stats, key=lambda s: s["total_ms"])
    dom_ms = dom["total_ms"] / dom["launches"]
    achieved = dom["flops"] / dom["launches"] / (dom_ms * 1e-3)
    gpu_ms = sum(s["total_ms"] for s in stats) / args.steps
    dom_kernel = "sfno_chain_kernel" if eng.chain is not None and dom["name"] in ("encoder", "mlp", "mlp_outer", "mlp_decoder") else "gemm_strided_kernel"
    out = {
        "metric": "6-h forecast steps/sec on 721x1440 state, 1/2/4/8 MI355X; per-channel max rel-err vs ref",
        "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"FourCastNet v2-small (SFNO: embed {cfg.embed_dim}, {cfg.num_layers} layers, scale factor {cfg.scale_factor}) 6-h "
                               f"autoregressive rollout, {cfg.n_lat}x{cfg.n_lon}x{cfg.in_chans} state, random-init weights, state resident in "
                               "HBM, 1 ensemble member per GPU",
                   "precision": "every linear map (1x1 convs, DFT, Legendre, dhconv) as a GEMM with fp16 hi/lo operands, 3 MFMA terms, fp32 "
                                "accumulate; fp32 activations" + ("; encoder, block MLPs (+ norm1) and MLP + decoder of the last block as one "
                                                                  "pixel-wise chain kernel each" if eng.chain is not None else ""),
                   "parallelism": f"member-parallel x{world}" if world > 1 else "single GPU", "finite": finite},
        # SFNO is a bandwidth-bound network (1.9 TFLOP against ~30 GB of fp32 activations per step): the roofline is the HBM one.
        # achieved = algorithmic bytes (A operand read once + output written once + residuals, fp32) of the dominant stage / its time
        "roofline": {"bound": "hbm", "kernel": dom["name"] + f" ({dom_kernel})", "achieved": dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9,
                     "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": dom["bytes"] / (dom["total_ms"] * 1e-3) / PEAK_HBM,
                     "traffic": pmc_kernels("sfno", dom_kernel).get("hbm_bytes_per_launch"),
                     "counters_all_gemm_launches": pmc_kernels("sfno", "gemm_strided_kernel"),
                     "avg_launch_ms": dom_ms, "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
                     "mfma": {"achieved_tflops": achieved / 1e12, "frac": achieved / PEAK_MFMA_BF16,
                              "note": "dense GEMM FLOPs of the dominant stage (the Legendre / dhconv GEMMs also multiply the l < m zeros)"},
                     "step": {"alg_tflop": f_step / 1e12, "alg_GB": sum(s["bytes"] for s in stats) / args.steps / 1e9, "gpu_ms": gpu_ms,
                              "hbm_frac": sum(s["bytes"] for s in stats) / args.steps / (gpu_ms * 1e-3) / PEAK_HBM,
                              "mfma_frac": f_step / (gpu_ms * 1e-3) / PEAK_MFMA_BF16,
                              "hbm_GB_measured": ((pmc_summary("sfno") or {}).get("total") or {}).get("hbm_GB_per_step")},
                     "stages": {s["name"]: {"ms_per_step": round(s["total_ms"] / args.steps, 4), "launches_per_step": s["launches"] // args.steps,
                                            "dense_tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 1),
                                            "alg_GBps": round(s["bytes"] / (s["total_ms"] * 1e-3) / 1e9, 1)} for s in stats}},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import sfno_oracle as O
        tr = O.Transforms(cfg)                           # quadrature / Legendre tables: setup, not timed (the engine prepares its own once, too)
        t0 = time.time()
        with torch.no_grad():
            O.forward(params, x_host, cfg, tr=tr)
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"ONE full {cfg.n_lat}x{cfg.n_lon} step of the PyTorch-CPU restatement (fp32 network, fp64 transforms) in {dt:.1f} s; no scaling"}
    if world == 1 and not args.no_parity:
        from oracle import sfno_oracle as O
        tiny = SfnoConfig(n_lat=97, n_lon=192, in_chans=11, out_chans=11, embed_dim=40, num_layers=4, scale_factor=3)
        tp, tx = init_synthetic(tiny, 0), synthetic_state(tiny, 0)
        te = SfnoEngine(tiny, dev)
        te.load_params(tp)
        with small_oracle_threads():
            out["parity"] = {"grid": "97x192", "max_rel_err": O.per_channel_rel_err(te.step(tx.to(dev)).cpu(), O.forward(tp, tx, tiny)).max().item(), "bar": 1e-3}
    return out


def run_graphcast(args, rank, local_rank, world, dist):
    """The same contract for the GraphCast row (BASELINE.json configs[3]): one step = one 6-h forward of the 0.25-degree, 13-level
    GraphCast (M6 multi-mesh, 16 processor layers) on synthetic 83-channel states resident in HBM; N > 1 = one member per rank (the
    2-GPU mesh split of configs[3] is not built: a step fits one GPU) + the closing ensemble reduction."""
    from skyrim_amd.graphcast.engine import GraphcastEngine
    from skyrim_amd.graphcast.spec import GraphcastConfig, alg_bytes_per_step, flops_per_stage, flops_per_step, flops_per_step_executed, forcings, init_synthetic, synthetic_states
    from skyrim_amd.pangu.ensemble import ensemble_mean_spread
    cfg = GraphcastConfig(n_lat=args.n_lat, n_lon=args.n_lon)
    dev = torch.device("cuda", local_rank)
    sharded = args.shard and world > 1
    comm = {}
    if sharded and args.backend != "nccl":           # gloo moves host memory: stage the two collectives (control-flow runs on a 1-GPU box only)
        def _reduce(t):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

        def _gather(out, mine):
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h.view(-1, h.shape[-1]), mine.cpu())
            out.copy_(h)
        comm = dict(reduce_fn=_reduce, gather_fn=_gather)
    eng = GraphcastEngine(cfg, dev, shard=(rank, world) if sharded else (0, 1), **comm)
    g = eng.graph
    params = init_synthetic(cfg, 0)
    eng.load_params(params)
    band = slice(eng.lat0, eng.lat1)
    x0h, x1h = synthetic_states(cfg, 0 if (sharded or world == 1) else rank)
    a, b = x0h[:, band].contiguous().to(dev), x1h[:, band].contiguous().to(dev)
    fcs = [forcings(cfg, 1000.0 + 6.0 * k)[:, band].contiguous().to(dev) for k in range(args.warmup + args.steps)]   # host-side preparation

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(k):
        nonlocal a, b
        nxt = eng.step(a, b, fcs[k], out=a)          # x(t-6h) is dead after the step: its buffer takes x(t+6h)
        a, b = b, nxt

    for k in range(args.warmup):
        step(k)
    if world > 1 and not sharded:
        ensemble_mean_spread([b], world)
    eng.profiling = True
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    if world > 1 and not sharded:
        ensemble_mean_spread([b], world)
    sync()
    elapsed = time.perf_counter() - t0
    stats = eng.profile_read()
    eng.profiling = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    finite = bool(torch.isfinite(b).all().item())
    if rank != 0:
        return
    f_step = flops_per_step(cfg, cfg.n_lat * cfg.n_lon, g.n_mesh, len(g.mesh_edges), len(g.g2m_edges) * (world if sharded else 1), 3 * cfg.n_lat * cfg.n_lon)
    f_exec = flops_per_step_executed(cfg, cfg.n_lat * cfg.n_lon, g.n_mesh, len(g.mesh_edges), len(g.g2m_edges) * (world if sharded else 1), 3 * cfg.n_lat * cfg.n_lon)
    counts = (cfg, cfg.n_lat * cfg.n_lon, g.n_mesh, len(g.mesh_edges), len(g.g2m_edges) * (world if sharded else 1), 3 * cfg.n_lat * cfg.n_lon)
    f_stage_pub, f_stage_exe = flops_per_stage(*counts), flops_per_stage(*counts, executed=True)
    dom = max(stats, key=lambda s: s["total_ms"])
    achieved = dom["flops"] / (dom["total_ms"] * 1e-3)
    gpu_ms = sum(s["total_ms"] for s in stats) / args.steps
    alg = alg_bytes_per_step(cfg, cfg.n_lat * cfg.n_lon, g.n_mesh, len(g.mesh_edges), len(g.g2m_edges) * (world if sharded else 1), 3 * cfg.n_lat * cfg.n_lon,
                             edge_bytes=2 if eng.fused else None)
    dom_alg = alg.get(dom["name"], 0.0)
    dom_step_ms = dom["total_ms"] / args.steps
    out = {
        "metric": "6-h forecast steps/sec on 721x1440 state, 1/2/4/8 MI355X; per-channel max rel-err vs ref",
        "value": (1 if sharded else world) * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"GraphCast (M{cfg.splits} multi-mesh: {g.n_mesh} nodes, {len(g.mesh_edges)} edges; {len(g.g2m_edges)} grid->mesh and "
                               f"{len(g.m2g_edges)} mesh->grid edges; latent {cfg.latent}, {cfg.steps} processor layers) 6-h autoregressive rollout, "
                               f"{cfg.n_lat}x{cfg.n_lon}x{cfg.n_vars} state, random-init weights (35.4 M), states resident in HBM, 1 member per GPU",
                   "precision": ("node MLPs and node-term GEMMs: fp16 hi/lo operands, 3 MFMA terms, fp32 accumulate, fp32 node latents; edge MLPs: "
                                 f"edge latents / hidden activations as ONE fp16 plane, second Linear with hi/lo weights (2 terms), W_e with {eng.w1_planes} "
                                 "plane(s)") if eng.fused else "every Linear as a GEMM with fp16 hi/lo operands, 3 MFMA terms, fp32 accumulate; fp32 latents",
                   "parallelism": (f"one forecast over {world} GPUs: latitude bands of the grid + mesh-node ranges (owner computes); per step one "
                                   f"all-reduce of the ({g.n_mesh} x {cfg.latent}) grid->mesh aggregate and {cfg.steps} all-gathers of the node latents") if sharded else
                                  (f"member-parallel x{world}" if world > 1 else "single GPU"), "finite": finite},
        # The roofline of the line is the one the dominant stage sits closer to: HBM on ALGORITHMIC bytes (spec.alg_bytes_per_step: every tensor of
        # the data flow once in, once out, at the engine's storage types) or the dense fp16 MFMA peak on the MFMA FLOPs the stage executes
        # (dense FLOPs x terms).  The round-3 kernel sequence was bandwidth-limited (fp32 latents, 235-278 GB measured per step); the fused
        # kernels keep the edge data on chip and are matrix-pipe / LDS-DMA limited
        # `frac` is SURVEY 8(d)'s figure: ALGORITHMIC FLOPs of the dominant stage (the network as published: every edge MLP on its concatenated
        # 3L-wide row) / its time / the dense fp16 MFMA peak -- recomputable as FLOPs / time / 2.5 PF.  `frac_executed`: the same with the dense
        # FLOPs the kernels run after the distributive rewrite of the edge MLPs' first Linear.  `mfma_busy_equiv`: those FLOPs weighted by
        # the MFMA terms per product (hi/lo operand planes) -- matrix-pipe occupancy, NOT a roofline fraction.
        "roofline": dict(
            ({"bound": "mfma", "achieved": f_stage_pub.get(dom["name"], dom["flops"] / args.steps) / (dom_step_ms * 1e-3) / 1e12, "peak": PEAK_MFMA_BF16 / 1e12, "unit": "TFLOP/s",
              "frac": f_stage_pub.get(dom["name"], dom["flops"] / args.steps) / (dom_step_ms * 1e-3) / PEAK_MFMA_BF16,
              "frac_executed": f_stage_exe.get(dom["name"], dom["flops"] / args.steps) / (dom_step_ms * 1e-3) / PEAK_MFMA_BF16,
              "mfma_busy_equiv": dom["mfma_flops"] / (dom["total_ms"] * 1e-3) / PEAK_MFMA_BF16,
              "alg_flops_per_step_of_stage": f_stage_pub.get(dom["name"]), "executed_flops_per_step_of_stage": f_stage_exe.get(dom["name"])}
             if dom["mfma_flops"] / PEAK_MFMA_BF16 > dom_alg * args.steps / PEAK_HBM else
             {"bound": "hbm", "achieved": dom_alg / (dom_step_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": dom_alg / (dom_step_ms * 1e-3) / PEAK_HBM}),
            **{"kernel": dom["name"] + (" (edge_update_kernel + node_mlp_kernel + gemm_strided_kernel_s)" if eng.fused else
                                         " (gemm_strided_kernel + sum_linear_ln_kernel + gather_gemm_kernel + linear_ln_kernel)"),
               "hbm": {"achieved_GBps": dom_alg / (dom_step_ms * 1e-3) / 1e9, "frac": dom_alg / (dom_step_ms * 1e-3) / PEAK_HBM},
               "alg_bytes_per_step_of_stage": dom_alg, "stage_ms_per_step": dom_step_ms,
               "traffic": pmc_kernels("graphcast", "edge_update_kernel" if eng.fused else "ln_kernel").get("hbm_bytes_per_step"),
               "mfma": {"achieved_tflops": dom["mfma_flops"] / (dom["total_ms"] * 1e-3) / 1e12, "frac": dom["mfma_flops"] / (dom["total_ms"] * 1e-3) / PEAK_MFMA_BF16,
                        "dense_tflops": achieved / 1e12, "note": "MFMA FLOPs the dominant stage executes (dense FLOPs x MFMA terms per product): pipe occupancy (= mfma_busy_equiv), not the roofline fraction"},
               "counters_edge_kernels": pmc_kernels("graphcast", "edge_update_kernel" if eng.fused else "ln_kernel"),
               "hbm_GB_per_step_all_kernels": ((pmc_summary("graphcast") or {}).get("total") or {}).get("hbm_GB_per_step"),
               "avg_launch_ms": dom["total_ms"] / dom["launches"],
               "step": {"alg_tflop": f_step / 1e12, "gpu_ms": gpu_ms, "mfma_frac": f_step / (gpu_ms * 1e-3) / PEAK_MFMA_BF16,
                        "executed_tflop": f_exec / 1e12, "alg_GB": alg["total"] / 1e9, "hbm_frac": alg["total"] / (gpu_ms * 1e-3) / PEAK_HBM,
                        "alg_GB_per_stage": {k: round(v / 1e9, 2) for k, v in alg.items() if k != "total"},
                        "note": "alg_tflop: the network as published (every edge MLP on the concatenated 1536-wide row); executed_tflop: the same "
                                "result with the first Linear of the edge MLPs taken apart by distributivity (what the kernels run)"},
               "stages": {s["name"]: {"ms_per_step": round(s["total_ms"] / args.steps, 3), "launches_per_step": s["launches"] // args.steps,
                                            "dense_tflops": round(s["flops"] / (s["total_ms"] * 1e-3) / 1e12, 1),
                                            "alg_GBps": round(alg.get(s["name"], 0.0) / (s["total_ms"] / args.steps * 1e-3) / 1e9, 1)} for s in stats}}),
    }
    if world == 1 and not args.no_cpu_baseline:
        # bounded sample with one piece per cost class, each timed on the host cores and scaled by its own count (not by FLOPs alone:
        # the edge updates are gather / scatter-bound on a CPU): (1) the grid-node embedder on 1/8 of the grid (dense, per grid node);
        # (2) ONE full processor layer on the multi-mesh (gathers over 327 660 edges + receiver sum + node update) x 16;
        # (3) the mesh->grid edge update + receiver sum on 1/16 of the grid's edges x 16; the remaining stages are priced from these
        from oracle import graphcast_graph as OG
        from oracle import graphcast_oracle as O
        og = OG.build(cfg.n_lat, cfg.n_lon, cfg.splits)
        L, gen = cfg.latent, torch.Generator().manual_seed(0)
        with torch.no_grad():
            mean, std = params["norm.mean"][:, None, None], params["norm.std"][:, None, None]
            n8 = og.n_grid // 8
            feats = torch.cat([(x0h - mean) / std, (x1h - mean) / std, fcs[0].cpu(), params["static"]], dim=0).flatten(1).T[:n8]
            t0 = time.time()
            O.mlp(params, "embed.grid", torch.cat([feats, torch.from_numpy(og.grid_node_feat[:n8])], dim=1))
            t_embed = (time.time() - t0) * 8
            vm, em = torch.randn(og.n_mesh, L, generator=gen), torch.randn(len(og.mesh_edges), L, generator=gen)
            me = torch.from_numpy(og.mesh_edges)
            t0 = time.time()
            de = O.edge_update(params, "proc.0.edge", em, vm, vm, me)
            vm2 = vm + O.mlp(params, "proc.0.node", torch.cat([vm, O.aggregate(de, me[:, 1], og.n_mesh)], dim=1))
            t_proc = (time.time() - t0) * cfg.steps
            n16 = og.n_grid // 16
            m2g = torch.from_numpy(og.m2g_edges[:3 * n16])
            e2, vg = torch.randn(3 * n16, L, generator=gen), torch.randn(n16, L, generator=gen)
            t0 = time.time()
            ee = O.edge_update(params, "m2g.edge", e2, vm2, vg, m2g)
            O.aggregate(ee, m2g[:, 1], n16)
            t_m2g = (time.time() - t0) * 16
        # priced from the samples: grid->mesh edges like mesh->grid edges per edge; the three other grid-node MLPs like the embedder per FLOP
        t_g2m = t_m2g * len(og.g2m_edges) / len(og.m2g_edges)
        f_embed = cfg.grid_in * L + L * L
        t_grid_mlps = t_embed * ((L * L + L * L) + (2 * L * L + L * L) + (L * L + L * cfg.n_vars)) / f_embed
        t_step = t_embed + t_proc + t_m2g + t_g2m + t_grid_mlps
        out["cpu_baseline"] = {"value": 1.0 / t_step, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"grid-node embedder on 1/8 of the grid ({t_embed / 8:.1f} s) + one full processor layer ({t_proc / cfg.steps:.1f} s) + mesh->grid "
                                         f"edge update and receiver sum on 1/16 of the grid ({t_m2g / 16:.1f} s); each scaled by its own count, the grid->mesh edges and the "
                                         "other grid-node MLPs priced from these; PyTorch-CPU fp32 restatement on the oracle's own graph",
                               "s_per_step_est": t_step}
    if world == 1 and not args.no_parity:
        from oracle import graphcast_oracle as O
        small = GraphcastConfig(n_lat=61, n_lon=120, splits=3, latent=512, steps=4)      # the production latent: the fused edge / node kernels
        se = GraphcastEngine(small, dev)
        sp = init_synthetic(small, 0)
        se.load_params(sp)
        s0, s1 = synthetic_states(small, 0)
        sf = forcings(small, 1000.0)
        y = se.step(s0.to(dev), s1.to(dev), sf.to(dev)).cpu()
        from oracle import graphcast_graph as OG
        with small_oracle_threads():
            ref = O.forward(sp, OG.build(small.n_lat, small.n_lon, small.splits), s0, s1, sf)
        out["parity"] = {"grid": "61x120, M3 mesh, latent 512, 4 processor layers", "fused_kernels": bool(se.fused), "max_rel_err": O.per_channel_rel_err(y, ref).max().item(),
                         "max_rel_err_of_increment": O.increment_rel_err(y, ref, s1).max().item(), "bar": 1e-3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default=None, help="engine precision mode (default: skyrim_amd.pangu.engine.DEFAULT_PRECISION)")
    ap.add_argument("--n-lat", type=int, default=721)
    ap.add_argument("--n-lon", type=int, default=1440)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-alt-modes", action="store_true", help="skip the short runs of the other precision modes")
    ap.add_argument("--members-per-gpu", type=int, default=2, help="pangu, N = 1: members advanced on separate streams in the members_per_gpu entry")
    ap.add_argument("--no-models", action="store_true", help="pangu, N = 1: skip the short SFNO / GraphCast runs reported under \"models\"")
    ap.add_argument("--members", type=int, default=0, help="pangu, --gpus > 1: ensemble members in total, sharded round-robin over the ranks "
                    "(BASELINE configs[4]: --gpus 8 --members 50 --save-every 1 --gather); default: one member per rank")
    ap.add_argument("--save-every", type=int, default=0, help="ensemble mean / spread (and --gather) every this many steps; default: once, after the last step")
    ap.add_argument("--gather", action="store_true", help="also all-gather the member states at every saved step")
    ap.add_argument("--reduce", default="allgather", choices=["allgather", "allreduce"], help="ensemble reduction: all-gather of per-rank partial "
                    "sums + local reduce (direct xGMI exchange, default) or ring all-reduce")
    ap.add_argument("--graph", action="store_true", help="pangu: replay one captured HIP graph per step instead of 72 eager launches (per-stage timing off)")
    ap.add_argument("--mlp", default="fused", choices=["fused", "split"], help="pangu: one-kernel MLP (default) or the two tiled GEMMs of round 1")
    ap.add_argument("--model", default="pangu", choices=["pangu", "sfno", "graphcast"],
                    help="pangu (default; BASELINE.json's headline configuration), sfno (FourCastNet v2-small, configs[2]) or graphcast (configs[3])")
    ap.add_argument("--shard", action="store_true", help="graphcast only: the N ranks share ONE forecast (latitude bands of the grid, mesh "
                    "replicated, one all-reduce of the mesh aggregate per step) instead of running one member each; strong scaling")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-rank control flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: the Pangu hot path has no CPU fallback")
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"{world} ranks but {n_dev} GPUs visible (RCCL needs one GPU per rank)")
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    if args.model in ("sfno", "graphcast"):
        line = (run_sfno if args.model == "sfno" else run_graphcast)(args, rank, local_rank, world, dist)
        if line is not None:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from skyrim_amd.pangu.engine import DEFAULT_PRECISION, PRECISIONS, PanguEngine
    from skyrim_amd.pangu.ensemble import ensemble_mean_spread
    args.precision = args.precision or DEFAULT_PRECISION
    if args.precision not in PRECISIONS:
        raise SystemExit(f"--precision must be one of {sorted(PRECISIONS)}")
    from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state

    geom = PanguGeometry(args.n_lat, args.n_lon)
    params = init_synthetic(geom, 0)
    dev = torch.device("cuda", local_rank)
    eng = PanguEngine(geom, args.precision, dev, mlp=args.mlp)
    eng.load_params(params)
    from skyrim_amd.pangu.ensemble import gather_members, member_shard
    n_members = args.members or world
    if n_members < world:
        raise SystemExit("--members must be >= --gpus")
    mine = member_shard(n_members, rank, world)              # round-robin: 50 members on 8 ranks -> 7,7,6,6,6,6,6,6
    x_host = synthetic_state(geom, 0, member=mine[0] if world > 1 else None)
    xs = [synthetic_state(geom, 0, member=m if world > 1 else None).to(dev) for m in mine]
    x = xs[0]
    save_every = args.save_every or args.steps

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def saved_step():
        out = ensemble_mean_spread(xs, n_members, args.reduce)
        return out + ((gather_members(xs, n_members),) if args.gather else ())

    for _ in range(args.warmup):
        for xm in xs:
            eng.step(xm, xm)
    graphs = [eng.capture(xm) for xm in xs] if args.graph else None

    def advance(i):
        if graphs is not None:
            graphs[i].replay()
        else:
            eng.step(xs[i], xs[i])

    if world > 1:   # warm the communicator outside the timed region
        saved_step()
    eng.profile(not args.graph)
    sync()
    t0 = time.perf_counter()
    for k in range(1, args.steps + 1):
        for i in range(len(xs)):
            advance(i)
        if world > 1 and (k % save_every == 0 or k == args.steps):
            saved = saved_step()
    sync()
    elapsed = time.perf_counter() - t0
    stats = eng.profile_read()
    eng.profile(False)
    if args.graph:        # per-stage events are not recorded inside a captured graph: time the stages in one eager step afterwards
        eng.profile(True)
        scratch = xs[0].clone()
        eng.step(scratch, scratch)
        stats = eng.profile_read()
        eng.profile(False)
        for s_ in stats:
            s_["total_ms"] *= args.steps * len(mine)
            s_["launches"] *= args.steps * len(mine)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    finite = bool(torch.isfinite(x).all().item())

    if rank == 0:
        timed = [s for s in stats if s["launches"] > 0]
        dom = max(timed, key=lambda s: s["total_ms"])
        dom_ms = dom["total_ms"] / dom["launches"]
        achieved = dom["flops"] / (dom_ms * 1e-3)
        gpu_ms = sum(s["total_ms"] for s in timed) / (args.steps * len(mine))
        out = {
            "metric": "6-h forecast steps/sec on 721x1440 state, 1/2/4/8 MI355X; per-channel max rel-err vs ref",
            "value": n_members * args.steps / elapsed,          # 6-h steps of all members per second (whole job)
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if args.precision.startswith("bf16") else "f16",
            "data": "synthetic",
            "config": {
                "workload": f"Pangu 6-h autoregressive rollout, {args.n_lat}x{args.n_lon}x69 state "
                            "(13 levels x 5 vars + 4 surface), random-init weights (64 M params), "
                            "state resident in HBM, 1 ensemble member per GPU",
                "precision": MODE_NOTES[args.precision][0],
                "parallelism": (f"member-parallel: {n_members} members round-robin over {world} GPUs ({len(mine)} on rank 0), ensemble mean / spread "
                                f"every {save_every} step(s) by {args.reduce}" + (", member states all-gathered at every saved step" if args.gather else ""))
                               if world > 1 else "single GPU",
                "members": n_members, "finite": finite,
                "calibration": eng.calibrated_on,       # the term plan's biases: "synthetic" = fitted on the engine's built-in state, not on this run's
                "rounding": eng.rounding,               # of the one-plane weights: "nearest" | "compensated" (pangu/calibration.py)
                # the MFMA term plan the timed steps ran with, and what the load-time guard measured for it (sigma-unit error of one step
                # against the three-term engine on the calibration state; PanguEngine._guard)
                "term_plan": f"{eng.term_plan_in_effect:#05x}",
                "guard": None if not eng.guard_report else [[f"{pl:#05x}", float(f"{e:.3e}")] for pl, e in eng.guard_report],
            },
            "roofline": {
                "bound": "mfma", "kernel": dom["name"], "achieved": achieved / 1e12, "peak": PEAK_MFMA_BF16 / 1e12,
                "unit": "TFLOP/s", "frac": achieved / PEAK_MFMA_BF16,
                "traffic": pmc_kernels("pangu", *PANGU_STAGE_KERNEL.get(dom["name"], ("?",))).get("hbm_bytes_per_launch"),
                "counters": pmc_kernels("pangu", *PANGU_STAGE_KERNEL.get(dom["name"], ("?",))),
                "hbm_bytes_per_step_all_kernels": ((pmc_summary("pangu") or {}).get("total") or {}).get("hbm_GB_per_step"),
                "alg_bytes_per_launch": dom["bytes"],
                "avg_launch_ms": dom_ms, "alg_flops_per_launch": dom["flops"],
                "step": {"alg_tflop": F_ALG_STEP / 1e12, "gpu_ms": gpu_ms,
                         "mfma_frac": F_ALG_STEP / (gpu_ms * 1e-3) / PEAK_MFMA_BF16,
                         "t_roof_ms": 1e3 * F_ALG_STEP / PEAK_MFMA_BF16},
                "stages": {s["name"]: {"ms_per_launch": round(s["total_ms"] / s["launches"], 4),
                                       "launches_per_step": s["launches"] // (args.steps * len(mine)),
                                       "tflops": round(s["flops"] / (s["total_ms"] / s["launches"] * 1e-3) / 1e12, 1),
                                       "alg_GBps": round(s["bytes"] / (s["total_ms"] / s["launches"] * 1e-3) / 1e9, 1)}
                           for s in timed},
            },
        }
        kept = {}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, geom, x_host, kept)
        if world == 1 and not args.no_parity:
            out["parity"] = toy_parity(args.precision)
            if "y" in kept and args.members <= 1:
                # the timed workload itself: this engine's first step from the synthetic state against the oracle step the CPU baseline just ran
                from oracle import pangu_oracle as O
                y_gpu = eng.step(x_host.to(dev)).cpu()
                out["parity"]["full_size"] = {"grid": f"{geom.n_lat}x{geom.n_lon}", "steps": 1, "max_rel_err": O.per_channel_rel_err(y_gpu, kept["y"]).max().item(),
                                              "max_sigma_err": O.per_channel_sigma_err(y_gpu, kept["y"], params["norm.std"]).max().item(),
                                              "note": "per channel, max|y - ref| / max|ref|; the 24-h rollout (4 steps, each asserted) is tests/test_pangu_gpu.py"}
                del y_gpu
        kept.clear()
        if world == 1 and not args.no_alt_modes:
            out["pcie_inclusive"] = api_rollout_rate(args.precision, geom, params, x_host, dev)
            eng = None
            torch.cuda.empty_cache()
            out["predict_inclusive"] = predict_inclusive(args.precision, geom, params, dev)
            out["members_per_gpu"] = members_on_streams(args.precision, geom, params, x_host, dev, args.members_per_gpu)
            torch.cuda.empty_cache()
            out["modes"] = {m: dict(quick_mode(m, geom, params, x_host, dev), note=MODE_NOTES[m][1])
                            for m in ("f16x2m", "f16x3q", "bf16x3") if m != args.precision}
            if not args.no_parity:
                # the load-time rounding of the one-plane weights (pangu/calibration.py): same kernels, same speed -- what the default gives away without it
                for m, rounding in ((args.precision, "nearest"),):
                    try:
                        out["modes"][m + "/" + rounding] = dict(quick_mode(m, geom, params, x_host, dev, rounding=rounding), parity=toy_parity(m, rounding),
                                                                note="one-plane weights rounded to the nearest fp16 + bias fold (no error feedback)")
                    except Exception as exc:                # an optional table entry must not take the headline down with it
                        out["modes"][m + "/" + rounding] = {"error": f"{type(exc).__name__}: {exc}"}
                    torch.cuda.empty_cache()
            out["modes"][args.precision + "/split-mlp"] = dict(quick_mode(args.precision, geom, params, x_host, dev, mlp="split"),
                                                               note="same arithmetic with the MLP as two tiled GEMMs (hidden through HBM): the round-1 path")
        if world == 1 and not args.no_models:
            # the other two rows of the hot path, driver-timed in the same invocation: a short run each (5 steps after 2 of warm-up)
            import copy
            eng = x = xs = graphs = None             # release the Pangu engine's arenas and states before GraphCast's 47 GB
            torch.cuda.empty_cache()
            sub = copy.copy(args)
            sub.steps, sub.warmup, sub.no_cpu_baseline, sub.no_parity, sub.shard = 5, 2, True, True, False
            out["models"] = {}
            for name, fn in (("sfno", run_sfno), ("graphcast", run_graphcast)):
                try:
                    line = fn(sub, 0, local_rank, 1, dist)
                    out["models"][name] = {"ms_per_step": line["ms_per_step"], "steps_per_s": line["value"], "steps": line["steps"], "warmup": line["warmup"],
                                           "workload": line["config"]["workload"], "precision": line["config"]["precision"], "finite": line["config"]["finite"],
                                           "roofline": {k: line["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if k in line["roofline"]},
                                           "step": line["roofline"].get("step"), "stages_ms_per_step": {k: v["ms_per_step"] for k, v in line["roofline"]["stages"].items()},
                                           "profile": f"profiles/{PROFILE_ROUND}_{name}_kernel_stats.csv, profiles/{PROFILE_ROUND}_{name}_pmc.json "
                                                      f"(python bench.py --model {name}, same kernels)"}
                except Exception as e:      # a failure of a secondary row must not take the headline line with it -- but it must be visible
                    out["models"][name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        emit(out)
    if world > 1:
        dist.destroy_process_group()


def compact_line(out: dict) -> dict:
    """The line the driver records: everything BENCH_rNN.json must hold, under 3 KB.  The whole detail goes to bench_detail.json."""
    keep = lambda d, keys: {k: d[k] for k in keys if d is not None and k in d}  # noqa: E731
    line = keep(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = out.get("config", {})
    line["config"] = {"workload": cfg.get("workload", "")[:160], "precision": cfg.get("precision", "")[:120], **keep(cfg, ("parallelism", "members", "finite", "calibration", "rounding", "term_plan", "guard"))}
    line["config"]["parallelism"] = str(line["config"].get("parallelism", ""))[:120]
    roof = out.get("roofline", {})
    line["roofline"] = keep(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "alg_flops_per_launch", "avg_launch_ms",
                                   "hbm_bytes_per_step_all_kernels"))
    if "step" in roof:
        line["roofline"]["step"] = keep(roof["step"], ("alg_tflop", "gpu_ms", "mfma_frac", "t_roof_ms"))
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {**keep(cb, ("value", "unit", "cores", "kind")), "sample": str(cb.get("sample", ""))[:120]}
    if "parity" in out:
        par = out["parity"]
        line["parity"] = {**keep(par, ("max_rel_err", "max_sigma_err", "bar", "grid")), **({"full_size": keep(par["full_size"], ("grid", "steps", "max_rel_err", "max_sigma_err"))} if "full_size" in par else {})}
    if "models" in out:
        line["models"] = {}
        for name, m in out["models"].items():
            if "error" in m:
                line["models"][name] = {"error": m["error"][:200]}
                continue
            r = m.get("roofline", {})
            line["models"][name] = {**keep(m, ("ms_per_step", "steps_per_s", "steps", "warmup", "finite")),
                                    "roofline": keep(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_executed", "mfma_busy_equiv"))}
            line["models"][name]["roofline"]["kernel"] = str(line["models"][name]["roofline"].get("kernel", ""))[:60]
    for k in ("pcie_inclusive", "predict_inclusive"):
        if k in out and isinstance(out[k], dict):
            line[k] = {kk: vv for kk, vv in out[k].items() if isinstance(vv, (int, float)) or (isinstance(vv, dict) and kk in ("save", "no_save", "save_first_rollout"))}
            line[k] = {kk: (keep(vv, ("ms_per_step",)) if isinstance(vv, dict) else vv) for kk, vv in line[k].items()}
    line["detail"] = "bench_detail.json"
    return line


def emit(out: dict):
    """The full record -> bench_detail.json (repo root, and gpurun_out/ when it exists: the file that travels back from a GPU box); the
    LAST stdout line is the compact JSON the driver parses."""
    here = os.path.dirname(os.path.abspath(__file__))
    for d in (here, os.path.join(here, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
    line = compact_line(out) if os.environ.get("SKYRIM_BENCH_FULL_LINE") != "1" else out
    text = json.dumps(line)
    if len(text) > 3000 and line is not out:            # a field grew: drop the optional entries, never the contract
        for k in ("predict_inclusive", "pcie_inclusive"):
            line.pop(k, None)
        text = json.dumps(line)
    print(text, flush=True)


if __name__ == "__main__":
    main()
