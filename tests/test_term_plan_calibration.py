"""The arithmetic behind ``skpangu_calibrate`` (include/skyrim_pangu.h), on the CPU restatement: a Linear run with its weights rounded
to ONE fp16 plane drops x @ (W - fp16(W)).T; the mean of that term over the tokens of a DIFFERENT state, folded into the bias, removes
most of the error the rounding adds to one 6-h step.  The GPU engine's own calibration is measured against the oracle in
tests/test_pangu_gpu.py; this is the statement it implements."""
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
