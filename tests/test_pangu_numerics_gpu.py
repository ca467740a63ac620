"""The default precision mode away from home: its one-plane weights are rounded with error feedback
against the operand statistics of ONE built-in state (mean + sigma x 9x9-box-smoothed noise), and until round 5 every test state was drawn
from that same family, on a synthetic network that damps perturbations.  Here, on the toy grid against the CPU oracle:
  * states the calibration never saw -- a k^-3 power-law spectrum, 3x3 and 31x31 smoothing, a zonal-mean meridional profile;
  * a calibration state with deliberately different per-channel statistics than the evaluated one;
  * a parameter set scaled to perturbation gain ~1 (tests/_states.py), 16 steps fed and free-running (20 measured: docs/experiments.md A.2).
Every error is printed in both units: SURVEY 8(d)'s max|d| / max|ref| per channel (asserted) and max|d| / sigma_c (oracle:
per_channel_sigma_err).  Tolerance: 5e-4 per step for the default mode (half the north star's 1e-3 bar)."""
import pytest
import torch

import _states
from oracle import pangu_oracle as O
from skyrim_amd.pangu.engine import DEFAULT_PRECISION

pytestmark = pytest.mark.gpu
OOD_TOL = 5e-4
MODES = sorted({DEFAULT_PRECISION, "f16x2m", "f16x1m"})
N_GAIN_STEPS = 16            # of the gain-1 rollout (20 measured in round 5: the figures settle after step 5)


def _errs(y, ref, params):
    return O.per_channel_rel_err(y, ref).max().item(), O.per_channel_sigma_err(y, ref, params["norm.std"]).max().item()


@pytest.fixture(scope="module")
def engines(toy):
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, _ = toy
    out = {}
    for m in MODES:
        out[m] = PanguEngine(g, m, "cuda:0")
        out[m].load_params(params)                    # default rounding, calibrated on the built-in state
    return out


@pytest.mark.parametrize("kind", _states.KINDS)
def test_states_the_calibration_never_saw(toy, engines, kind):
    g, params, _ = toy
    x = _states.state(g, kind)
    ref1 = O.forward(params, x)
    ref2 = O.forward(params, ref1)
    for m, e in engines.items():
        y1 = e.step(x.cuda()).cpu()
        y2 = e.step(ref1.cuda()).cpu()               # fed the oracle's forecast: the second step of a rollout
        (r1, s1), (r2, s2) = _errs(y1, ref1, params), _errs(y2, ref2, params)
        print(f"{kind:10s} {m}: step 1 rel {r1:.2e} sigma {s1:.2e} | step 2 rel {r2:.2e} sigma {s2:.2e}")
        assert torch.isfinite(y1).all() and max(r1, r2) < (OOD_TOL if m == DEFAULT_PRECISION else 1e-3), (kind, m, r1, r2)


def test_calibration_state_with_other_channel_statistics(toy):
    """calibrate on a state whose channels are rescaled and shifted (and meridionally structured), evaluate on the usual one --
    a fitted plan must not be worse off than the bar on a distribution it was not fitted to."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    mean, std = params["norm.mean"][:, None, None], params["norm.std"][:, None, None]
    c = torch.arange(g.n_channels, dtype=torch.float32)[:, None, None]
    other = mean + std * ((0.4 + 0.1 * (c % 7)) * _states.unit_field(g, "meridional", 3) + 1.5 * torch.sin(c))   # sigma x0.4 .. x1.0, means shifted by up to 1.5 sigma
    ref = O.forward(params, x)
    for m in MODES:
        e = PanguEngine(g, m, "cuda:0")
        e.load_params(params, calibration=other.contiguous())
        r, s = _errs(e.step(x.cuda()).cpu(), ref, params)
        print(f"calibrated on other statistics, {m}: rel {r:.2e} sigma {s:.2e}")
        assert r < (OOD_TOL if m == DEFAULT_PRECISION else 1e-3), (m, r)
        del e


def test_gain_one_network_rollout(toy):
    """A network whose 6-h map carries a perturbation at gain ~1 (the random-init one damps it 12x per step, so 'no growth over 20 steps'
    said nothing): (a) fed the oracle's state at every step, (b) free-running."""
    from skyrim_amd.pangu.engine import PanguEngine
    g, params, x = toy
    p1 = _states.gain_one_params(params)
    std = p1["norm.std"][:, None, None]
    d = 1e-6 * std * torch.randn(x.shape, generator=torch.Generator().manual_seed(2))
    a, b, gains = x, x + d, []
    for _ in range(4):                               # the oracle's own response to a 1e-6 sigma perturbation
        a, b = O.forward(p1, a), O.forward(p1, b)
        gains.append((((b - a) / std).norm() / (d / std).norm()).item())
    print("gain-1 network: perturbation norm after 1..4 steps / initial " + " ".join(f"{v:.2f}" for v in gains))
    assert 0.5 < gains[0] < 2.0 and 0.2 < gains[-1] < 8.0, gains
    oracle_states = [x]
    for k in range(N_GAIN_STEPS):
        oracle_states.append(O.forward(p1, oracle_states[-1]))
    import os
    for m in (MODES if os.environ.get("SKYRIM_TEST_ALL_MODES") else [DEFAULT_PRECISION]):
        e = PanguEngine(g, m, "cuda:0")
        e.load_params(p1)                             # with the load-time guard: on this network it judges the plan (in sigma units) and may fall back
        print(f"gain-1 {m}: load-time guard {[(hex(a), float(f'{b:.2e}')) for a, b in (e.guard_report or [])]} -> plan {e.term_plan_in_effect:#05x}")
        assert not e.guard_report or e.guard_report[-1][1] < 5e-4
        e.load_params(p1, guard=False)                # the plan AS NAMED is what the figures below are of
        free, fed_err, free_err = x.cuda().clone(), [], []
        for k in range(N_GAIN_STEPS):
            fed_err.append(_errs(e.step(oracle_states[k].cuda()).cpu(), oracle_states[k + 1], p1))
            e.step(free, free)
            free_err.append(_errs(free.cpu(), oracle_states[k + 1], p1))
        print(f"gain-1 {m} fed  rel " + " ".join(f"{r:.1e}" for r, _ in fed_err))
        print(f"gain-1 {m} fed  sig " + " ".join(f"{s:.1e}" for _, s in fed_err))
        print(f"gain-1 {m} free rel " + " ".join(f"{r:.1e}" for r, _ in free_err))
        assert torch.isfinite(free).all()
        assert max(r for r, _ in fed_err) < (OOD_TOL if m == DEFAULT_PRECISION else 1e-3), (m, fed_err)
        assert max(r for r, _ in free_err) < 1e-3, (m, free_err)     # free-running differences accumulate at gain 1: held to the north star's bar
        del e
