"""Committed golden vectors of the CPU oracles at BASELINE.json's full sizes (tests/golden/full_<model>.npz, written by
tests/golden/make_full_size.py on a many-core host), and the comparison of an engine state with them.

Why.  One 721x1440 oracle step costs one to two minutes on 128 host threads; rounds 3-5 ran six of them inside `pytest -m gpu` (background host
jobs, tests/_oracle_jobs.py) and the suite sat at 792 s of the driver's 1200 s limit with its parity record one slow box away from a timeout.
The oracle's full-size trajectories are deterministic functions of seeds, so they are computed ONCE, reduced to what a comparison needs and
committed; the GPU tests compare against the fixture (SKYRIM_TEST_LIVE_ORACLE=1 brings the live host jobs back, all grid points).

What a fixture holds, per step k of the oracle's own autoregressive rollout and per channel c (SELF-ORACLE, parity unpinned: DESIGN.md 2):
  samples[k][c]   the oracle's values on a lattice of grid points -- every LAT_STRIDE-th latitude row (both poles included) x every
                  LON_STRIDE-th longitude: 49 x 45 = 2205 points per channel, strides that are not multiples of any window / patch size
  absmax[k][c]    max |ref| over the WHOLE field (the denominator of SURVEY 8(d)'s per-channel error, exact)
  incmax[k][c]    max |ref_k - ref_(k-1)| over the whole field (GraphCast: the size of the predicted increment)
  cells[k][c]     the mean of the whole field over each cell of a 24 x 24 partition (30 x 60 points): a wrong window / tile / token anywhere
                  moves its cell's mean far outside the tolerance even when no lattice point falls in it
The per-channel error of a state y is then  max over lattice points |y - ref| / absmax  (a lower bound of the all-points figure, reported as
such) together with  max over cells |mean(y) - mean(ref)| / absmax.  Test infrastructure like oracle/: nothing under skyrim_amd/ imports it."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
LAT_STRIDE, LON_STRIDE, CELLS = 15, 32, 24


def lattice(n_lat: int, n_lon: int):
    ii = np.unique(np.concatenate([np.arange(0, n_lat, LAT_STRIDE), [n_lat - 1]]))
    return ii, np.arange(0, n_lon, LON_STRIDE)


def cell_means(y: torch.Tensor) -> torch.Tensor:
    """(C, H, W) -> (C, CELLS, CELLS) float64 means over the cells of an (almost) even partition; on the tensor's own device."""
    C, H, W = y.shape
    he = torch.linspace(0, H, CELLS + 1).round().long().tolist()
    we = torch.linspace(0, W, CELLS + 1).round().long().tolist()
    out = torch.empty(C, CELLS, CELLS, dtype=torch.float64, device=y.device)
    for a in range(CELLS):
        rows = y[:, he[a]:he[a + 1]].double().sum(1)                      # (C, W)
        cs = torch.cat([rows.new_zeros(C, 1), rows.cumsum(1)], 1)
        for b in range(CELLS):
            out[:, a, b] = (cs[:, we[b + 1]] - cs[:, we[b]]) / ((he[a + 1] - he[a]) * (we[b + 1] - we[b]))
    return out


def reduce_field(ref: torch.Tensor, prev: "torch.Tensor | None") -> dict:
    """What the fixture keeps of one oracle state (C, H, W); ``prev``: the state before it (the increment's reference) or None."""
    ii, jj = lattice(ref.shape[-2], ref.shape[-1])
    r = ref.float()
    out = {"samples": r[:, ii][:, :, jj].numpy().astype(np.float32), "absmax": r.abs().amax(dim=(1, 2)).numpy().astype(np.float32),
           "cells": cell_means(r).numpy().astype(np.float32)}
    out["incmax"] = ((r - prev.float()).abs().amax(dim=(1, 2)).numpy().astype(np.float32) if prev is not None else np.zeros(r.shape[0], np.float32))
    return out


class FullSizeGolden:
    def __init__(self, model: str):
        self.path = GOLDEN / f"full_{model}.npz"
        if not self.path.exists():
            raise FileNotFoundError(f"{self.path}: written by `python tests/golden/make_full_size.py {model}` on a many-core host")
        z = np.load(self.path, allow_pickle=False)
        self.samples, self.absmax, self.incmax, self.cells = z["samples"], z["absmax"], z["incmax"], z["cells"]
        self.ii, self.jj = z["ii"], z["jj"]
        self.meta = str(z["meta"])
        self.steps = self.samples.shape[0]

    def errors(self, k: int, y: torch.Tensor) -> dict:
        """Per-channel figures of engine state ``y`` (C, H, W; any device) against step ``k`` of the oracle's rollout: ``rel`` (lattice points,
        relative to max|ref| of the whole field), ``cell`` (cell means, same denominator), ``inc`` (lattice points relative to the predicted
        increment's size)."""
        ii, jj = torch.as_tensor(self.ii, device=y.device), torch.as_tensor(self.jj, device=y.device)
        ys = y.index_select(1, ii).index_select(2, jj).double().cpu().numpy()
        d = np.abs(ys - self.samples[k].astype(np.float64)).max(axis=(1, 2))
        den = np.maximum(self.absmax[k].astype(np.float64), 1e-30)
        cm = cell_means(y).cpu().numpy()
        dc = np.abs(cm - self.cells[k].astype(np.float64)).max(axis=(1, 2))
        return {"rel": d / den, "cell": dc / den, "inc": d / np.maximum(self.incmax[k].astype(np.float64), 1e-30)}
