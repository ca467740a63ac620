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


class _OracleTapEngine:
    """Stands in for the tiled three-term GPU engine in ``calibration.engine_taps``: the same stage-level calls and debug buffers
    (window tables, attention output and hidden activation as hi/lo fp16 planes in the blocked layout), filled from the oracle."""
    mlp, term_plan = "split", 0

    def __init__(self, g, params):
        import importlib.util
        from pathlib import Path
        spec = importlib.util.spec_from_file_location("gpu_diag", Path(__file__).resolve().parent.parent / "tools" / "gpu_diag.py")
        self.diag = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(self.diag)                         # block_reference: an oracle block with its intermediates
        self.geom, self.p, self.device = g, params, torch.device("cpu")
        self.og = O.Geometry(g.n_lat, g.n_lon)
        self.bufs = {}
        self.plane = {"ao": max(self._mwin(1) * 192, self._mwin(2) * 384), "hid": max(g.tokens(1) * 768, g.tokens(2) * 1536)}

    def _mwin(self, layer):
        z, h, w = self.geom.res(layer)
        return z * self.geom.padded_lat(layer) * w

    def tokens(self, layer):
        return self.geom.tokens(layer), self.geom.dim(layer)

    def patch_embed(self, state):
        xn = (state - self.p["norm.mean"][:, None, None]) / self.p["norm.std"][:, None, None]
        return O.patch_embed(self.p, self.og, *O.split_state(xn))

    def downsample(self, x):
        return O.downsample(self.p, self.og, x)

    def upsample(self, x):
        return O.upsample(self.p, self.og, x)

    def step(self, x):
        return O.forward(self.p, x)

    def _store(self, name, rows):
        hi = rows.to(torch.float16)
        lo = (rows - hi.float()).to(torch.float16)
        r, k = rows.shape
        buf = torch.zeros(2 * self.plane[name], dtype=torch.float16)
        for j, t in enumerate((hi, lo)):
            buf[j * self.plane[name]: j * self.plane[name] + r * k] = t.reshape(r // 16, 16, k // 32, 32).permute(0, 2, 1, 3).reshape(-1)
        self.bufs[name] = buf

    def block(self, layer, i, x):
        ref = self.diag.block_reference(O._block_params(self.p, layer, i), x, self.og.res(layer), O.HEADS[layer - 1], i % 2 == 1)
        self._store("ao", ref["ao"])
        self._store("hid", ref["hid"])
        return ref["y"]

    def debug_buffer(self, name, dtype):
        if name.startswith("widx"):
            layer, roll = (1, 2)[int(name[4])], int(name[5])
            z, h, w = self.geom.res(layer)
            hp, top = self.geom.padded_lat(layer), self.geom.pad_top(layer)
            t = F.pad(torch.arange(z * h * w).reshape(z, h, w) + 1, (0, 0, top, hp - h - top)) - 1
            if roll:
                t = torch.roll(t, shifts=(-1, -3, -6), dims=(0, 1, 2))
            return t.reshape(z // 2, 2, hp // 6, 6, w // 12, 12).permute(0, 2, 4, 1, 3, 5).reshape(-1).to(dtype)
        return self.bufs[name].view(dtype)


def test_engine_taps_reads_back_the_operands_of_every_block(monkeypatch):
    """calibration.engine_taps end to end on the CPU, the GPU engine replaced by an oracle-backed stand-in with the same buffers: the
    rows it delivers for the calibration state and its forecast ARE the operands of the oracle's Linears (pooled: second half = the
    forecast's), and the parameters fitted to them hold their error over a rollout where a single-state fit does not."""
    from skyrim_amd.pangu.calibration import calibrated_params, engine_taps
    g = PanguGeometry(49, 192)
    p = init_synthetic(g, 3)
    x, x_cal = synthetic_state(g, 3), synthetic_state(g, 11)
    eng = _OracleTapEngine(g, p)
    states = [x_cal, eng.step(x_cal)]
    taps = list(engine_taps(eng, p, states))
    assert [(layer, i) for layer, i, _ in taps] == [(layer, i) for layer in (1, 2, 3, 4) for i in range((2, 6, 6, 2)[layer - 1])]
    want = {(layer, i): ops for layer, i, ops in _oracle_taps(monkeypatch, p, states[1])}      # the oracle's own operands on the forecast
    for layer, i, ops in taps:
        n = g.tokens(layer)
        for kind in ("mlp.fc1", "mlp.fc2"):
            assert ops[kind].shape[0] == 2 * n
            ref = want[(layer, i)][kind]
            assert ((ops[kind][n:] - ref).abs().max() / ref.abs().max()).item() < 1e-5, (layer, i, kind)
        q_in = want[(layer, i)]["attn.qkv"]                           # window order without the padding rows: compare as multisets of rows
        assert ops["attn.qkv"][n:].shape == q_in.shape and torch.allclose(ops["attn.qkv"][n:].sum(0), q_in.sum(0), rtol=1e-3, atol=1e-2)
    plan = 0x99
    sub = [t for t in taps if t[0] in (1, 4)]
    single = [(layer, i, {k: v[: g.tokens(layer)] for k, v in ops.items()}) for layer, i, ops in sub]
    refs, s = [], x
    for _ in range(3):
        s = O.forward(p, s)
        refs.append(s)
    errs = {}
    for tag, t in (("single", single), ("pooled", sub)):
        q = calibrated_params(p, plan, t)
        s, e = x, []
        for k in range(3):
            s = O.forward(q, s)
            e.append(O.per_channel_rel_err(s, refs[k]).max().item())
        errs[tag] = e
    print(errs)
    assert max(errs["pooled"]) < 1e-4 and errs["pooled"][2] < 0.7 * errs["single"][2]
