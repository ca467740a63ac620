"""Load-time preparation of the Linears a term plan runs with ONE fp16 weight plane (include/skyrim_pangu.h: skpangu_config.term_plan).

Such a Linear computes ``x @ Q.T + b'`` with ``Q`` on the fp16 grid instead of ``x @ W.T + b``.  Two things can be chosen freely at load
time, and neither costs anything per step (the kernels see fp16 weights and fp32 biases either way):

* the bias: ``b' = b + (W - Q) @ mean(x)`` removes the mean of the dropped term over the tokens of a calibration state (the C ABI does
  this on its own for nearest rounding: ``skpangu_calibrate``);
* the rounding: instead of rounding every weight to its NEAREST fp16 neighbour, round column k, then push the error it made onto the
  not-yet-rounded columns k + 1 .. K - 1 in the direction the operand's covariance ``H = cov(x)`` says will cancel it in the OUTPUT
  (``compensated_round``: the column-by-column error feedback of optimal-brain / GPTQ quantisation, on the fp16 grid).  What remains is the
  part of ``x @ (W - Q).T`` that no choice of later columns could have cancelled.

Measured on the CPU restatement (tests/test_term_plan_calibration.py; statistics from a state that is NOT the forecast's): per-channel error
of one 6-h step with proj / fc1 / fc2 of all 16 blocks on one plane -- nearest 4.2e-4, nearest + mean fold 2.0e-4, compensated + mean fold
7.5e-5 (three terms everywhere: 6.6e-5).

The statistics are pooled over the calibration state AND its own 6-h forecast (``PanguEngine.calibrate``): fitted to the first alone, the
error of a rollout climbs back towards nearest rounding from its second step on -- the second step's input is a model output, a
different distribution -- (oracle emulation, plan 0xFF, four steps: 6.3e-5 / 2.6e-4 / 2.4e-4 / 2.5e-4); fitted to both it stays where it
started (6.6e-5 / 5.5e-5 / 6.8e-5 / 7.1e-5).

``engine_taps`` collects the operands on the GPU: one step of the THREE-term engine in its tiled form (mlp="split": the attention output
and the hidden activation reach HBM there), stage by stage through the C ABI's stage-level entry points, reading the engine's own buffers
back.  Nothing here runs per step, and nothing here is an alternative compute path: the forecasts run on the HIP kernels only.
"""
from __future__ import annotations

from typing import Iterable, Iterator

import torch
import torch.nn.functional as F

from .spec import DEPTHS

KINDS = ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")


def short_kinds(term_plan: int, layer: int) -> tuple[str, ...]:
    """The Linears of ``layer`` (1..4) that ``term_plan`` runs with one weight plane."""
    kinds = ()
    if (term_plan >> (layer - 1)) & 1:
        kinds += ("attn.proj", "mlp.fc1", "mlp.fc2")
    if (term_plan >> (3 + layer)) & 1:
        kinds += ("attn.qkv",)
    return kinds


def operand_statistics(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(mean [K], covariance [K, K]) of the rows of ``x`` [rows, K]; float64, centred before the products (the second moment the
    rounding is fitted to is ``cov + outer(mean, mean)``, see ``calibrated_params``)."""
    x = x.reshape(-1, x.shape[-1])
    mu = x.double().mean(0)
    xc = (x - mu.to(x.dtype)).float()
    cov = torch.zeros(x.shape[1], x.shape[1], dtype=torch.float64, device=x.device)
    step = 1 << 16
    for r in range(0, xc.shape[0], step):                       # fp32 products of at most 65536 rows, summed in float64
        c = xc[r:r + step]
        cov += (c.T @ c).double()
    return mu, cov / x.shape[0]


def compensated_round(w: torch.Tensor, cov: torch.Tensor, damp: float = 0.01, block: int = 128) -> torch.Tensor:
    """``w`` [N, K] -> ``q`` [N, K] with every entry on the fp16 grid (returned as float32), minimising ``(w - q) cov (w - q).T`` row by
    row with the greedy column order 0 .. K - 1: after column k is rounded, ``(w_k - q_k) / U[k, k] * U[k, k+1:]`` is subtracted from
    the columns still to come, ``U`` the upper Cholesky factor of ``inv(cov + damp * mean(diag cov) * I)``.  Rows are independent and
    share ``cov``; ``block`` columns are updated eagerly, the rest once per block."""
    n, k = w.shape
    h = cov.double().clone()
    scale = h.diagonal().mean().clamp_min(1e-30)
    h += torch.eye(k, dtype=torch.float64, device=h.device) * (damp * scale)
    u = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(h)), upper=True)
    work = w.double().clone()
    q = torch.empty_like(work)
    for b0 in range(0, k, block):
        b1 = min(b0 + block, k)
        wb, ub = work[:, b0:b1].clone(), u[b0:b1, b0:b1]
        eb = torch.empty_like(wb)
        for j in range(b1 - b0):
            qj = wb[:, j].to(torch.float16).double()
            q[:, b0 + j] = qj
            e = (wb[:, j] - qj) / ub[j, j]
            wb[:, j:] -= e[:, None] * ub[j, j:][None, :]
            eb[:, j] = e
        work[:, b1:] -= eb @ u[b0:b1, b1:]
    if not torch.isfinite(q).all():
        raise FloatingPointError("compensated_round: a weight left the fp16 range")
    return q.float()


def calibrated_params(params: dict, term_plan: int, taps: Iterable, rounding: str = "compensated", damp: float = 0.01) -> dict:
    """A copy of ``params`` in which every Linear ``term_plan`` runs short has fp16-grid weights (``rounding``: "compensated" |
    "nearest") and the bias that takes the mean of what the rounding dropped.  ``taps`` yields ``(layer, block, {kind: operand rows})``
    with the operands of the three-term network on the calibration state (``engine_taps``; the tests drive it from the oracle)."""
    if rounding not in ("compensated", "nearest"):
        raise ValueError(f"rounding = {rounding!r}: 'compensated' or 'nearest'")
    out = dict(params)
    seen = set()
    for layer, i, operands in taps:
        for kind in short_kinds(term_plan, layer):
            pre = f"layer{layer}.block{i}.{kind}."
            x = operands[kind]
            if kind != "attn.qkv" and (term_plan >> (7 + layer)) & 1:
                x = x.to(torch.float16).float()                   # a one-term layer: this GEMM reads the operand's fp16 hi plane
            w = params[pre + "weight"].to(x.device, torch.float32)
            b = params[pre + "bias"].to(x.device, torch.float32)
            mu, cov = operand_statistics(x)
            # fitted to the UNCENTRED second moment: with the bias fold the mean of THIS state drops out exactly, so the centred covariance
            # would be the exact objective -- for this state.  It leaves the rounding free to put its error along the mean direction, and
            # the residual stream's channel means are large: on another state, whose mean differs a little, a one-term QKV fitted that
            # way is no better than nearest rounding (3.1e-4 against 2.1e-5 with the mean direction penalised; proj / fc1 / fc2 do not
            # care: 3.8e-5 / 4.9e-5).
            q = compensated_round(w, cov + torch.outer(mu, mu), damp) if rounding == "compensated" else w.to(torch.float16).float()
            out[pre + "weight"] = q.cpu()
            out[pre + "bias"] = (b.double() + (w.double() - q.double()) @ mu).float().cpu()
            seen.add((layer, i))
    want = {(layer, i) for layer in range(1, 5) for i in range(DEPTHS[layer - 1]) if short_kinds(term_plan, layer)}
    if seen != want:
        raise RuntimeError(f"calibration taps covered blocks {sorted(seen)}, the plan needs {sorted(want)}")
    return out


# --------------------------------------------------------------------------------------------- #
#  the operands, from the engine's own buffers (GPU)
# --------------------------------------------------------------------------------------------- #
def _planes(buf: torch.Tensor, n: int) -> torch.Tensor:
    """hi + lo of a two-plane fp16 buffer (plane stride = half of the buffer) -> fp32 [n]"""
    t = buf.view(torch.float16)
    per = t.numel() // 2
    return t[:n].float() + t[per:per + n].float()


def _unblock(flat: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """[rows / 16][cols / 32][16][32] blocked layout (csrc/common.h blk_off) -> [rows][cols]"""
    return flat[:rows * cols].reshape(rows // 16, cols // 32, 16, 32).permute(0, 2, 1, 3).reshape(rows, cols)


def engine_taps(eng, params: dict, states) -> Iterator:
    """One step of a three-term, tiled-form engine (``PanguEngine(geom, "f16x3q", mlp="split")`` with ``params`` loaded) on every state
    of ``states`` (a (69, n_lat, n_lon) tensor or a list of them), stage by stage and in lock-step; yields ``(layer, block, operands)``
    where ``operands[kind]`` holds the fp32 [tokens, K] operand rows of that Linear for all states, one after the other: the stream
    rounded to fp16 for the QKV (what a one-term QKV reads), the attention output in token order, the mid-block stream
    ``x + LayerNorm(proj(attention))``, the hidden activation."""
    if eng.mlp != "split" or eng.term_plan:
        raise ValueError("engine_taps needs the tiled three-term engine (mlp='split', no term plan)")
    if isinstance(states, torch.Tensor):
        states = [states]
    dev = eng.device
    xs = [eng.patch_embed(s.to(dev, torch.float32).contiguous()) for s in states]
    for layer in range(1, 5):
        if layer == 2:
            xs = [eng.downsample(x) for x in xs]
        elif layer == 4:
            xs = [eng.upsample(x) for x in xs]
        ntok, c = eng.tokens(layer)
        res = 0 if layer in (1, 4) else 1
        for i in range(DEPTHS[layer - 1]):
            pre = f"layer{layer}.block{i}."
            p = {k: params[pre + k].to(dev, torch.float32) for k in ("attn.proj.weight", "attn.proj.bias", "norm1.weight", "norm1.bias")}
            widx = eng.debug_buffer(f"widx{res}{i & 1}", torch.int32).long()     # window row -> stream token, -1 on padding rows
            mwin = widx.numel()
            valid = widx >= 0
            per_state = []
            for k, x in enumerate(xs):
                y = eng.block(layer, i, x)
                ao_win = _unblock(_planes(eng.debug_buffer("ao", torch.uint8), mwin * c), mwin, c)
                ao = torch.empty(ntok, c, dtype=torch.float32, device=dev)
                ao[widx[valid]] = ao_win[valid]
                mid = x + F.layer_norm(F.linear(ao, p["attn.proj.weight"], p["attn.proj.bias"]), (c,), p["norm1.weight"], p["norm1.bias"], 1e-5)
                hid = _unblock(_planes(eng.debug_buffer("hid", torch.uint8), ntok * 4 * c), ntok, 4 * c)
                per_state.append({"attn.qkv": x.to(torch.float16).float(), "attn.proj": ao, "mlp.fc1": mid, "mlp.fc2": hid})
                xs[k] = y
            yield layer, i, {kind: torch.cat([o[kind] for o in per_state]) for kind in KINDS}
