"""The arithmetic behind ``skpangu_calibrate`` (include/skyrim_pangu.h), on the CPU restatement: a Linear run with its weights rounded
to ONE fp16 plane drops x @ (W - fp16(W)).T; the mean of that term over the tokens of a DIFFERENT state, folded into the bias, removes
most of the error the rounding adds to one 6-h step.  The GPU engine's own calibration is measured against the oracle in
tests/test_pangu_gpu.py; this is the statement it implements."""
import pytest
import torch
import torch.nn.functional as F

from oracle import pangu_oracle as O
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state

SHORT = ("attn.proj", "mlp.fc1", "mlp.fc2")


def _fp16(t):
    return t.to(torch.float16).to(t.dtype)


def test_bias_fold_removes_most_of_the_weight_rounding_error(monkeypatch):
    g = PanguGeometry(49, 96)
    p = init_synthetic(g, 3)
    x, x_cal = synthetic_state(g, 3), synthetic_state(g, 11)
    names = {id(v): k for k, v in p.items()}
    ref = O.forward(p, x)
    mode = {"round": False, "fold": None, "collect": None}
    plain = O._linear

    def linear(xx, w, b=None, *a, **kw):
        key = names.get(id(w), "")
        short = any(c in key for c in SHORT)
        if mode["collect"] is not None and short:
            mode["collect"][key] = xx.reshape(-1, xx.shape[-1]).mean(0)
        if mode["round"] and short:
            w16 = _fp16(w)
            if mode["fold"] is not None:
                c = (w - w16) @ mode["fold"][key]
                b = c if b is None else b + c
            w = w16
        return F.linear(xx, w, b)

    monkeypatch.setattr(O, "_linear", linear)
    mode["collect"] = {}
    O.forward(p, x_cal)                                   # operand means of the three-term network on the calibration state
    means, mode["collect"] = mode["collect"], None
    assert len(means) == 3 * 16
    mode["round"] = True
    e_plain = O.per_channel_rel_err(O.forward(p, x), ref).max().item()
    mode["fold"] = means
    e_fold = O.per_channel_rel_err(O.forward(p, x), ref).max().item()
    monkeypatch.setattr(O, "_linear", plain)
    assert 1e-5 < e_plain < 2e-3                          # the rounding is visible ...
    assert e_fold < 0.7 * e_plain                         # ... and its mean was most of it (measured 0.4-0.5x at 49x192 and 73x288)


def test_builtin_calibration_state_is_fixed_and_is_not_a_test_state():
    """``calibration_state``: built from the weights' own normalisation constants and a fixed seed -- the same for every load of a set of
    weights, and none of the synthetic states the parity tests or bench.py forecast from (seeds 0..11)."""
    from skyrim_amd.pangu.engine import CALIBRATION_SEED, calibration_state
    g = PanguGeometry(25, 96)
    p = init_synthetic(g, 0)
    a = calibration_state(g, p["norm.mean"], p["norm.std"])
    b = calibration_state(g, p["norm.mean"], p["norm.std"])
    assert a.shape == (69, 25, 96) and a.dtype == torch.float32 and torch.equal(a, b)
    assert CALIBRATION_SEED not in range(0, 100)
    for seed in range(12):
        assert not torch.allclose(a, synthetic_state(g, seed))
    # its statistics are the normalisation constants' (what makes it a stand-in for an analysis of that climate)
    z = (a - p["norm.mean"].reshape(-1, 1, 1)) / p["norm.std"].reshape(-1, 1, 1)
    assert z.mean().abs() < 0.2 and abs(z.flatten(1).std(1).mean().item() - 1.0) < 0.05


def _oracle_taps(monkeypatch, p, x_cal):
    """(layer, block, {kind: operand rows}) of the oracle's three-term network on ``x_cal`` -- what calibration.engine_taps reads from the
    GPU engine's buffers (window-ordered rows here, padding rows included: statistics do not care about the order)."""
    names = {id(v): k for k, v in p.items()}
    got = {}
    plain = O._linear

    def linear(xx, w, b=None, *a, **kw):
        key = names.get(id(w), "")
        for kind in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            if key.endswith(kind + ".weight"):
                layer, blk = int(key[5]), int(key.split(".")[1][5:])
                rows = xx.reshape(-1, xx.shape[-1])
                if kind == "attn.qkv":
                    rows = _fp16(rows[rows.abs().sum(1) > 0])                 # the stream's hi plane, padding rows out
                got.setdefault((layer, blk), {})[kind] = rows
        return plain(xx, w, b, *a, **kw)

    monkeypatch.setattr(O, "_linear", linear)
    O.forward(p, x_cal)
    monkeypatch.setattr(O, "_linear", plain)
    return [(layer, blk, ops) for (layer, blk), ops in sorted(got.items())]


def test_compensated_rounding_beats_nearest_rounding(monkeypatch):
    """skyrim_amd/pangu/calibration.py: the weights of the full-resolution layers 1 / 4 (the sensitive ones) on the fp16 grid
    -- nearest, nearest + mean fold, error-compensated + mean fold -- against the unrounded network, statistics from another state."""
    from skyrim_amd.pangu.calibration import calibrated_params, compensated_round, short_kinds
    assert short_kinds(0x6F, 1) == ("attn.proj", "mlp.fc1", "mlp.fc2") and short_kinds(0x6F, 2)[-1] == "attn.qkv" and short_kinds(0x00, 3) == ()
    g = PanguGeometry(49, 192)
    p = init_synthetic(g, 3)
    x, x_cal = synthetic_state(g, 3), synthetic_state(g, 11)
    plan = 0x99                                                   # layers 1 and 4: proj / fc1 / fc2 AND the QKV on one plane
    taps = [t for t in _oracle_taps(monkeypatch, p, x_cal) if t[0] in (1, 4)]
    ref = O.forward(p, x)
    errs = {}
    nearest = calibrated_params(p, plan, taps, rounding="nearest")
    plain = dict(nearest)
    for k in p:
        if k.endswith(".bias"):
            plain[k] = p[k]                                        # nearest rounding with the ORIGINAL biases
    for tag, q in (("nearest", plain), ("nearest + fold", nearest), ("compensated + fold", calibrated_params(p, plan, taps))):
        for k, v in q.items():
            if k.endswith("weight") and k.startswith(("layer1", "layer4")) and any(s in k for s in SHORT + ("attn.qkv",)):
                assert torch.equal(v, _fp16(v)) and not torch.equal(v, p[k])          # on the fp16 grid, and it is a different matrix
        errs[tag] = O.per_channel_rel_err(O.forward(q, x), ref).max().item()
    print(errs)
    assert errs["nearest + fold"] < 0.8 * errs["nearest"]
    assert errs["compensated + fold"] < 0.5 * errs["nearest + fold"]
    # the routine itself: with an isotropic covariance there is nothing to exploit and it IS nearest rounding
    w = torch.randn(8, 64)
    assert torch.equal(compensated_round(w, torch.eye(64, dtype=torch.float64)), _fp16(w))
    with pytest.raises(RuntimeError):
        calibrated_params(p, 0x0F, taps)                              # the plan names layers the taps did not cover


def test_plane_readback_helpers_invert_the_blocked_layout():
    """calibration._planes / _unblock against csrc/common.h's blk_off: element (row, col) of a [rows][K] operand sits at
    ((row >> 4) * (K >> 5) + (col >> 5)) * 512 + (row & 15) * 32 + (col & 31) of its plane; hi at the start of the buffer, lo one
    plane (half the buffer) further."""
    from skyrim_amd.pangu.calibration import _planes, _unblock
    rows, k, plane = 48, 96, 48 * 96 + 512            # the plane is larger than the operand (sized for the other resolution)
    x = torch.randn(rows, k) * 3.0
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    r, c = torch.meshgrid(torch.arange(rows), torch.arange(k), indexing="ij")
    off = ((r >> 4) * (k >> 5) + (c >> 5)) * 512 + (r & 15) * 32 + (c & 31)
    buf = torch.zeros(2 * plane, dtype=torch.float16)
    buf[off.reshape(-1)] = hi.reshape(-1)
    buf[plane + off.reshape(-1)] = lo.reshape(-1)
    got = _unblock(_planes(buf.view(torch.uint8), rows * k), rows, k)
    assert torch.equal(got, hi.float() + lo.float()) and (got - x).abs().max() < 1e-5
