"""Binding of the C-ABI Pangu engine (include/skyrim_pangu.h) + torch plumbing.

Context management (sizes, create, prepare, profiling) is plain ctypes; every call that launches kernels on the hot path goes
through the ``torch.ops.skyrim_hip.pangu_*`` custom ops (skyrim_amd/ops.py), which hand raw device pointers and torch's current
stream to the C ABI.  PyTorch is device memory, streams and the dispatcher only.  There is no CPU fallback: constructing an
engine without the built HIP library (or without a GPU) raises, and the ops have no CPU kernel.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

from .. import ops
from .spec import PanguGeometry

PREC_BF16X3 = 0
PREC_F16X3 = 3
PREC_F16X3_Q = 4
# per-layer MFMA term plan (skpangu_config.term_plan): bit l = layer l + 1 runs proj / fc1 / fc2 with two terms (weights as ONE fp16 plane);
# bits 4-7: the layer's QKV with ONE term (stream hi plane x weight hi plane); bits 8-11: the layer's proj / fc1 / fc2 with ONE term (the
# activation operands as their fp16 hi plane too).  Any other plan: PanguEngine(geom, "f16x3q", term_plan=...).
TERM_PLANS = {"f16x1m": 0x66F, "f16x2m": 0x6F, "f16x2c": 0x66}
PRECISIONS = {"bf16x3": PREC_BF16X3, "f16x3": PREC_F16X3, "f16x3q": PREC_F16X3_Q, **{m: PREC_F16X3_Q for m in TERM_PLANS}}
# Modes (all: fp16 hi/lo ACTIVATION planes in the residual stream, fp32 accumulation / LayerNorm / softmax / GELU, fp16 attention core):
#   "f16x1m"  DEFAULT since round 5 (plan 0x66F).  Every block's proj / fc1 / fc2 weights are ONE fp16 plane; in the coarse layers 2 / 3 (C = 384:
#             12 of the 16 blocks, 72 % of the FLOPs) those GEMMs read the activations' hi plane only -- ONE MFMA term, A_hi W -- and the QKV
#             is one term too; the full-resolution layers 1 / 4 keep two terms (A_hi W + A_lo W) and hi/lo QKV weights.  The one-plane
#             weights are rounded with error feedback against the (rounded) operands' statistics (rounding="compensated", below), which is
#             what pays for the activation rounding.  16.2 ms per step at 721x1440 against 18.4 for "f16x2m"; per-channel error over the
#             full-size 24-h rollout 1.95 / 2.7 / 2.8 / 2.8e-4 (2.0 - 3.05e-4 across builds; f16x2m: 1.2 / 1.6 / 1.5 / 1.6e-4), on states the calibration never saw
#             (power-law spectrum, other smoothing scales, meridional structure) 1.6 - 2.0e-4, on a gain-1 network over 20 steps <= 4.2e-4
#             (tests/test_pangu_numerics_gpu.py) -- three times inside the 1e-3 bar everywhere it was looked at.
#   "f16x2m"  round 3 / 4's default (0x6F): two terms in every block, one-term QKV in layers 2 / 3.
#   "f16x2c"  0x66: layers 1 / 4 at three terms.   "f16x3q": three terms everywhere (~1e-4), QKV from the stream's hi plane.
#   "f16x3"   three terms, hi/lo QKV operand.      "bf16x3": the wide-range alternative (activations beyond fp16's 65504).
# The term a one-plane Linear drops, A x (W - fp16(W)), has its mean over a calibration state folded into the bias at load time
# (``PanguEngine.calibrate``).
DEFAULT_PRECISION = "f16x1m"
# Rounding of the one-plane weights (PanguEngine.load_params).  "compensated" (pangu/calibration.py: error feedback against the operand
# statistics of the calibration state and of its own forecast) is the default since round 4: a load-time choice, the kernels and the step time
# are the same.  With 1 % of the weight rows scaled x5 the two-term plan holds 9e-5 (nearest: 2.8e-4); x30 on every Linear class at once
# breaks EVERY mode, three-term and bf16x3 included, at ~1e-2 (tools/pangu_outlier_scan.py: the attention's one-plane fp16 operands).
DEFAULT_ROUNDING = "compensated"
# Load-time precision guard (PanguEngine._guard): a term plan is never applied blind to weights it was not tested on.  After the plan is prepared,
# ONE step of it and ONE step of the three-term engine on the calibration state are compared, per channel, in units of the channel's sigma
# (norm.std); at or above GUARD_TOL -- half the 1e-3 bar -- the engine falls back  plan -> plan without one-term block GEMMs -> three terms
# everywhere (0x66F -> 0x6F -> 0x00), re-fitting each time, and says so.  SKYRIM_PANGU_GUARD=off skips it (timing-only sections of bench.py).
# When even three terms everywhere disagree with the tiled three-term reference by GUARD_TOL the weights themselves are the problem: FloatingPointError.
GUARD_TOL = float(os.environ.get("SKYRIM_PANGU_GUARD_TOL", "5e-4"))

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libskyrim_pangu.so"


class SkConfig(ctypes.Structure):
    _fields_ = [("n_lat", ctypes.c_int), ("n_lon", ctypes.c_int), ("precision", ctypes.c_int),
                ("roll_sign", ctypes.c_int), ("pad_mode", ctypes.c_int), ("mask_value", ctypes.c_float), ("mlp_mode", ctypes.c_int), ("term_plan", ctypes.c_int),
                ("surface_last", ctypes.c_int), ("qkv_order", ctypes.c_int), ("bias_transposed", ctypes.c_int)]


class SkSizes(ctypes.Structure):
    _fields_ = [("master_floats", ctypes.c_longlong), ("prepared_bytes", ctypes.c_size_t),
                ("workspace_bytes", ctypes.c_size_t), ("state_floats", ctypes.c_longlong),
                ("n_params", ctypes.c_int)]


class SkStageStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 24), ("launches", ctypes.c_int), ("total_ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


EXPORTS = [
    "skpangu_abi_version", "skpangu_error_string", "skpangu_query_sizes", "skpangu_param_info",
    "skpangu_create", "skpangu_destroy", "skpangu_prepare", "skpangu_calibrate", "skpangu_step", "skpangu_patch_embed",
    "skpangu_block", "skpangu_downsample", "skpangu_upsample", "skpangu_patch_recover",
    "skpangu_debug_buffer", "skpangu_profile", "skpangu_profile_read",
]

_lib = None
ABI_VERSION = 6            # include/skyrim_pangu.h SKPANGU_ABI_VERSION


def load_library() -> ctypes.CDLL:
    """Load libskyrim_pangu.so (built in-tree by ``__graft_entry__.build()`` / ``make -C skyrim_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SKYRIM_PANGU_LIB", str(_LIB_PATH))
    if not os.path.exists(path):
        raise RuntimeError(
            f"HIP engine library not found at {path}; build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " -- there is no CPU fallback for the Pangu hot path")
    lib = ctypes.CDLL(path)
    vp, ci, cll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    lib.skpangu_abi_version.restype = ci
    lib.skpangu_error_string.restype = ctypes.c_char_p
    lib.skpangu_error_string.argtypes = [ci]
    lib.skpangu_query_sizes.argtypes = [ctypes.POINTER(SkConfig), ctypes.POINTER(SkSizes)]
    lib.skpangu_param_info.argtypes = [ctypes.POINTER(SkConfig), ci, ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.POINTER(cll), ctypes.POINTER(ci), ctypes.POINTER(cll * 6)]
    lib.skpangu_create.argtypes = [ctypes.POINTER(SkConfig), vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.skpangu_destroy.argtypes = [vp]
    lib.skpangu_destroy.restype = None
    lib.skpangu_prepare.argtypes = [vp, vp, vp]
    lib.skpangu_calibrate.argtypes = [vp, vp, vp, vp]
    lib.skpangu_step.argtypes = [vp, vp, vp, vp]
    lib.skpangu_patch_embed.argtypes = [vp, vp, vp, vp]
    lib.skpangu_block.argtypes = [vp, ci, ci, vp, vp]
    lib.skpangu_downsample.argtypes = [vp, vp, vp, vp]
    lib.skpangu_upsample.argtypes = [vp, vp, vp, vp]
    lib.skpangu_patch_recover.argtypes = [vp, vp, vp, vp, vp]
    lib.skpangu_debug_buffer.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    lib.skpangu_profile.argtypes = [vp, ci]
    lib.skpangu_profile_read.argtypes = [vp, ctypes.POINTER(SkStageStat), ci, ctypes.POINTER(ci)]
    for name in EXPORTS:
        getattr(lib, name)
    if lib.skpangu_abi_version() != ABI_VERSION:              # a stale build or a SKYRIM_PANGU_LIB variant from other sources
        raise RuntimeError(f"{path}: skpangu ABI {lib.skpangu_abi_version()}, this package binds ABI {ABI_VERSION} (include/skyrim_pangu.h); rebuild the library")
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().skpangu_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


PAD_MODES = {"centre": 0, "back": 1}


MLP_MODES = {"fused": 0, "split": 1}


SURFACE, QKV_ORDERS, BIAS_INDEX = {"first": 0, "last": 1}, {"3hd": 0, "h3d": 1}, {"qk": 0, "kq": 1}


def make_config(geom: PanguGeometry, precision: str = DEFAULT_PRECISION, roll_sign: int = -1, mask_value: float = -100.0, mlp: str = "fused",
                term_plan: int | None = None, surface: str = "first", qkv_order: str = "3hd", bias_index: str = "qk") -> SkConfig:
    """``skpangu_config`` of a geometry + the switchable conventions (include/skyrim_pangu.h; oracle: pangu_oracle.Conventions).
    ``term_plan`` overrides the precision name's per-layer term plan (bit l: layer l + 1 runs proj / fc1 / fc2 with two MFMA terms)."""
    if roll_sign not in (-1, 1):
        raise ValueError("roll_sign is -1 (Swin: roll by -(1,3,6) first) or +1 (pseudocode as written)")
    plan = TERM_PLANS.get(precision, 0) if term_plan is None else int(term_plan)
    if plan and (precision not in (*TERM_PLANS, "f16x3", "f16x3q") or mlp != "fused"):
        raise ValueError("a term plan needs fp16 planes (f16x2 / f16x3 / f16x3q) and the fused kernels")
    return SkConfig(geom.n_lat, geom.n_lon, PRECISIONS[precision], roll_sign, PAD_MODES[geom.pad], float(mask_value), MLP_MODES[mlp], plan,
                    SURFACE[surface], QKV_ORDERS[qkv_order], BIAS_INDEX[bias_index])


def query_sizes(geom: PanguGeometry, precision: str = DEFAULT_PRECISION, cfg: SkConfig | None = None) -> SkSizes:
    cfg = cfg or make_config(geom, precision)
    out = SkSizes()
    _check(load_library().skpangu_query_sizes(ctypes.byref(cfg), ctypes.byref(out)), "skpangu_query_sizes")
    return out


def param_table(geom: PanguGeometry, precision: str = DEFAULT_PRECISION) -> list[tuple[str, int, tuple[int, ...]]]:
    """(name, element offset, shape) of every master parameter, as the library lays them out."""
    lib = load_library()
    cfg = make_config(geom, precision)
    n = query_sizes(geom, precision).n_params
    out = []
    for i in range(n):
        name = ctypes.create_string_buffer(128)
        off, nd, shape = ctypes.c_longlong(), ctypes.c_int(), (ctypes.c_longlong * 6)()
        _check(lib.skpangu_param_info(ctypes.byref(cfg), i, name, 128, ctypes.byref(off), ctypes.byref(nd), ctypes.byref(shape)),
               "skpangu_param_info")
        out.append((name.value.decode(), off.value, tuple(shape[j] for j in range(nd.value))))
    return out


CALIBRATION_SEED = 20240


def calibration_state(geom: PanguGeometry, mean: torch.Tensor, std: torch.Tensor, seed: int = CALIBRATION_SEED) -> torch.Tensor:
    """The built-in calibration state of a term plan: mean_c + std_c x unit-variance smooth noise, from the model's OWN normalisation
    constants and a fixed seed, so that a set of weights is always prepared the same way whatever it forecasts first."""
    from .spec import smooth_noise
    n = smooth_noise(geom, seed)
    return (mean.reshape(-1, 1, 1).float().cpu() + std.reshape(-1, 1, 1).float().cpu() * n).contiguous()


class PanguEngine:
    """Device-resident Pangu 6-h step.  ``step`` maps a (69, n_lat, n_lon) fp32 CUDA tensor to the next state."""

    def __init__(self, geom: PanguGeometry | None = None, precision: str = DEFAULT_PRECISION, device: str | torch.device = "cuda:0",
                 roll_sign: int = -1, mask_value: float = -100.0, mlp: str = "fused", term_plan: int | None = None,
                 surface: str = "first", qkv_order: str = "3hd", bias_index: str = "qk"):
        """``roll_sign`` / ``mask_value`` / ``geom.pad``: the conventions the public pseudocode leaves open.
        ``mlp``: "fused" (default; one kernel per MLP in the 3-term modes, csrc/fused_mlp.hip) or "split" (two tiled GEMMs).
        ``term_plan``: per-layer two-term mask (include/skyrim_pangu.h); None = the precision name's own.
        ``surface`` / ``qkv_order`` / ``bias_index``: three more open conventions (oracle: Conventions of the same names), prepare-time only."""
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("PanguEngine needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.geom = geom or PanguGeometry()
        self.precision = precision
        self.device = torch.device(device)
        if mlp != "fused" and precision in TERM_PLANS and term_plan is None:
            term_plan = 0                                   # the tiled-GEMM path has no two-term kernels: a plan name + split = f16x3q + split
        self._conventions = dict(roll_sign=roll_sign, mask_value=mask_value, surface=surface, qkv_order=qkv_order, bias_index=bias_index)
        self.mlp = mlp
        self._ctx = None
        self._create(term_plan)
        self.term_plan_requested = self.term_plan
        self.guard_report = None                            # [(plan, sigma-unit error against the three-term engine), ...] of the last load / calibrate
        self.state_shape = (self.geom.n_channels, self.geom.n_lat, self.geom.n_lon)
        # compensated rounding pools the operand statistics of the calibration state and of this many of its successive 6-h forecasts
        self.calibration_forecasts = int(os.environ.get("SKYRIM_PANGU_CALIBRATION_FORECASTS", "1"))

    def _create(self, term_plan: "int | None"):
        """(Re)build the C context and its two arenas for ``term_plan`` (None: the precision name's own).  The guard's fall-back path."""
        self.release()
        c = self._conventions
        self.cfg = make_config(self.geom, self.precision, c["roll_sign"], c["mask_value"], self.mlp, term_plan, c["surface"], c["qkv_order"], c["bias_index"])
        self.term_plan = self.cfg.term_plan
        self.sizes = query_sizes(self.geom, self.precision, self.cfg)
        with torch.cuda.device(self.device):
            self._prepared = torch.empty(self.sizes.prepared_bytes, dtype=torch.uint8, device=self.device)
            self._workspace = torch.empty(self.sizes.workspace_bytes, dtype=torch.uint8, device=self.device)
        ctx = ctypes.c_void_p()
        _check(self.lib.skpangu_create(ctypes.byref(self.cfg), self._prepared.data_ptr(), self.sizes.prepared_bytes,
                                       self._workspace.data_ptr(), self.sizes.workspace_bytes, ctypes.byref(ctx)),
               "skpangu_create")
        self._ctx = ctx

    @property
    def term_plan_in_effect(self) -> int:
        """The plan the engine runs with (the requested one unless the load-time guard fell back)."""
        return self.term_plan

    def release(self):
        """Destroy the C context and drop the arenas (``skpangu_destroy``; GlobalModel.release_model).  The engine is unusable afterwards."""
        ctx = getattr(self, "_ctx", None)
        if ctx:
            self.lib.skpangu_destroy(ctx)
        self._ctx = None
        self._prepared = self._workspace = self._master = None

    def __del__(self):
        if getattr(self, "lib", None) is not None:
            self.release()

    # ------------------------------------------------------------------ #
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk_dev(self, t: torch.Tensor, shape=None):
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("expected a contiguous float32 tensor on the engine device")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return ctypes.c_void_p(t.data_ptr())

    def load_params(self, params: dict[str, torch.Tensor], calibration: "torch.Tensor | str | None" = "default", rounding: str = "default",
                    guard: "bool | None" = None):
        """Pack fp32 master parameters into the library's blob layout, upload and prepare.

        Engines with a term plan only (one-plane Linears; both are load-time choices that cost nothing per step):
        ``calibration``: the state their biases (and, with compensated rounding, their weights) are fitted on -- "synthetic":
        ``calibration_state`` built from the parameters' own normalisation constants, the same for every forecast; a (69, n_lat, n_lon)
        tensor: that state (e.g. a real analysis); None / "off": none.  "default" is "synthetic" unless SKYRIM_PANGU_CALIBRATION says
        otherwise (the counter profiles run with "off": calibration launches kernels once that would otherwise be summed into their
        per-step totals; it changes weights' last bits and biases, not timings).
        ``rounding``: how their weights reach the fp16 grid -- "nearest" (the C ABI's own; the bias fold is ``skpangu_calibrate``) or
        "compensated" (pangu/calibration.py: column-by-column error feedback against the operand covariance of the calibration state,
        computed here and handed to the library as the master weights); "default": SKYRIM_PANGU_ROUNDING or DEFAULT_ROUNDING.
        ``guard``: the load-time precision guard (``_guard``: one step of the plan against one of the three-term engine, fall back
        0x66F -> 0x6F -> 0x00 until inside GUARD_TOL); None: on unless SKYRIM_PANGU_GUARD=off; False: measurements of a plan as given."""
        self._guard_on = guard
        if isinstance(calibration, str) and calibration == "default":
            calibration = os.environ.get("SKYRIM_PANGU_CALIBRATION", "synthetic")
            if calibration == "first":                          # the time loop's mode: it calls calibrate() itself
                calibration = "off"
        if rounding == "default":
            rounding = os.environ.get("SKYRIM_PANGU_ROUNDING", DEFAULT_ROUNDING)
        if rounding not in ("nearest", "compensated"):
            raise ValueError(f"rounding = {rounding!r}: 'nearest' or 'compensated'")
        if isinstance(calibration, str) and calibration not in ("synthetic", "off"):
            raise ValueError(f"calibration = {calibration!r}: expected 'synthetic', 'off', None or a state tensor")
        off = calibration is None or (isinstance(calibration, str) and calibration == "off")
        self._params, self.rounding = params, rounding          # calibrate() starts over from these
        self.calibrated_on = None
        self._guard_pair = None
        if self.term_plan != self.term_plan_requested:          # an earlier load fell back: every load starts from the plan asked for
            self._create(self.term_plan_requested)
        self._prepare(params)
        fit = None
        if self.term_plan and not off:
            fit = calibration_state(self.geom, params["norm.mean"], params["norm.std"]) if isinstance(calibration, str) else calibration
            self._fit(fit)
        self._guard(fit)
        if fit is not None:
            self.calibrated_on = "synthetic" if isinstance(calibration, str) else "state"

    def _prepare(self, params: dict[str, torch.Tensor]):
        table = param_table(self.geom, self.precision)
        with torch.cuda.device(self.device):
            master = torch.zeros(self.sizes.master_floats, dtype=torch.float32, device=self.device)
            for name, off, shape in table:
                t = params[name]
                if tuple(t.shape) != shape:
                    raise ValueError(f"{name}: expected {shape}, got {tuple(t.shape)}")
                master[off:off + t.numel()] = t.reshape(-1).to(self.device, torch.float32)
            _check(self.lib.skpangu_prepare(self._ctx, master.data_ptr(), self._stream()), "skpangu_prepare")
            torch.cuda.current_stream(self.device).synchronize()
        self._master = master if self.term_plan else None       # skpangu_calibrate re-reads weights and biases from it

    def calibrate(self, state: "torch.Tensor | None"):
        """Fit the one-plane Linears of the term plan to ``state`` (None: back to nearest rounding and the master biases), then run the
        load-time guard on it; a no-op for engines without a plan.  Calling again starts over from the parameters handed to ``load_params``."""
        if not self.term_plan_requested:
            return
        if getattr(self, "_params", None) is None:
            raise RuntimeError("load_params() first")
        if self.term_plan != self.term_plan_requested:          # an earlier guard fell back: every fit starts from the plan asked for
            self._create(self.term_plan_requested)
            self._prepare(self._params)
        self._guard_pair = None
        self._fit(state)
        self._guard(state)

    def _fit(self, state: "torch.Tensor | None"):
        """rounding "nearest": fold the mean of the dropped term, A x (W - fp16(W)), into each bias -- one step on ``state`` through the
        three-term kernels, column means of every short Linear's operand (include/skyrim_pangu.h: skpangu_calibrate).
        rounding "compensated": the operands of one three-term step on ``state`` and of one on its forecast (a second, tiled-form
        engine that lives for the duration of this call), their pooled second moments, error-compensated fp16 weights + folded biases
        (pangu/calibration.py), prepared again."""
        if not self.term_plan:
            return
        if self.rounding == "compensated":
            params = self._params
            if state is not None:
                from .calibration import calibrated_params, engine_taps
                tap = PanguEngine(self.geom, "f16x3q", self.device, mlp="split", **self._conventions)
                tap.load_params(self._params, calibration="off")
                with torch.no_grad(), torch.cuda.device(self.device):
                    states = [state.to(self.device, torch.float32).contiguous()]
                    for _ in range(self.calibration_forecasts):  # the state and its own forecast(s): a rollout's later inputs are model outputs
                        states.append(tap.step(states[-1]))
                    params = calibrated_params(self._params, self.term_plan, engine_taps(tap, self._params, states))
                if len(states) > 1:
                    self._guard_pair = (states[0], states[1])   # the guard's reference: the three-term forecast of the calibration state
                tap.release()
                del tap
            self._prepare(params)                               # fp16-grid weights: the library's own rounding leaves them as they are
            torch.cuda.empty_cache()
        else:
            with torch.cuda.device(self.device):
                x = None if state is None else state.to(self.device, torch.float32).contiguous()   # None: back to the master biases
                _check(self.lib.skpangu_calibrate(self._ctx, self._master.data_ptr(), None if x is None else self._chk_dev(x, self.state_shape),
                                                  self._stream()), "skpangu_calibrate")
                torch.cuda.current_stream(self.device).synchronize()
        self.calibrated_on = None if state is None else "state"

    def _guard(self, state: "torch.Tensor | None"):
        """The load-time precision guard (GUARD_TOL above).  ``state``: what the plan was fitted on (None: the built-in calibration state).
        Leaves ``guard_report`` = [(plan, error), ...] in the order tried and the engine on the first plan below the tolerance; plan 0 is the
        three-term engine itself and always ends the chain."""
        self.guard_report = None
        on = getattr(self, "_guard_on", None)
        if on is None:
            on = os.environ.get("SKYRIM_PANGU_GUARD", "on").lower() not in ("off", "0", "no")
        if not self.term_plan or not on:
            return
        import time
        import warnings
        t0 = time.perf_counter()
        p = self._params
        fit = state
        with torch.no_grad(), torch.cuda.device(self.device):
            if self._guard_pair is not None:
                x, ref = self._guard_pair
            else:
                x = (calibration_state(self.geom, p["norm.mean"], p["norm.std"]) if state is None else state).to(self.device, torch.float32).contiguous()
                three = PanguEngine(self.geom, "f16x3q", self.device, mlp="split", **self._conventions)    # the tiled three-term form, like the fit's tap engine
                three.load_params(p, calibration="off")
                ref = three.step(x)
                three.release()
                del three
            sigma = p["norm.std"].to(self.device, torch.float32).reshape(-1)
            chain = []
            for plan in (self.term_plan, self.term_plan & 0xFF, 0):
                if plan not in chain:
                    chain.append(plan)
            report = []
            for plan in chain:
                if plan != self.term_plan:
                    self._create(plan)
                    self._prepare(p)
                    if fit is not None:
                        pair = self._guard_pair
                        self._fit(fit)
                        self._guard_pair = pair
                err = ((self.step(x) - ref).abs().amax(dim=(1, 2)) / sigma).max().item()
                report.append((plan, err))
                if err < GUARD_TOL:
                    break
        self._guard_pair = None
        self.guard_report = report
        self.guard_seconds = time.perf_counter() - t0
        torch.cuda.empty_cache()
        tried = ", ".join(f"{pl:#05x}: {e:.2e}" for pl, e in report)
        if report[-1][1] >= GUARD_TOL:
            # plan 0 IS three terms everywhere (the fused kernels); the reference is the tiled three-term form of the same arithmetic.  Two fp32-class
            # evaluations that far apart mean the WEIGHTS amplify rounding beyond the bar (measured: 1 % of every Linear's rows x30 -> 1e-2 in every
            # mode, the CPU oracle included): refused like a non-finite forecast, unless SKYRIM_PANGU_GUARD=warn.
            msg = (f"Pangu weights amplify fp32-class rounding differences beyond {GUARD_TOL:g} sigma on the calibration state in every term plan ({tried}): "
                   "no mode of this engine can promise the 1e-3 bar on them")
            if os.environ.get("SKYRIM_PANGU_GUARD", "on").lower() != "warn":
                raise FloatingPointError(msg + " (SKYRIM_PANGU_GUARD=warn to run anyway)")
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        elif len(report) > 1:
            warnings.warn(f"Pangu term plan {self.term_plan_requested:#05x} is outside {GUARD_TOL:g} sigma of the three-term engine on these weights "
                          f"({tried}); running plan {self.term_plan:#05x}", RuntimeWarning, stacklevel=3)

    def step(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(x)
        self._chk_dev(x, self.state_shape), self._chk_dev(out, self.state_shape)
        ops.hip.pangu_step(self._ctx.value, x, out)
        return out

    def capture(self, x: torch.Tensor) -> "torch.cuda.CUDAGraph":
        """Capture one in-place step ``x <- Pangu6(x)`` (its 72 kernel launches) as a HIP graph on ``x``'s storage; ``graph.replay()``
        then advances ``x`` by 6 h with ONE launch from the host.  The library allocates nothing and records no events outside
        profiling, so the step is capturable as is.  A step is GPU-bound at 721x1440 (the launches are queued ahead of
        the GPU either way); the graph matters on small grids and when the host is busy (ensemble drivers, I/O threads)."""
        self._chk_dev(x, self.state_shape)
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                  # warm-up outside the capture: hipFuncSetAttribute etc. happen here
                keep = x.clone()
                ops.hip.pangu_step(self._ctx.value, x, x)
                x.copy_(keep)
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ops.hip.pangu_step(self._ctx.value, x, x)
            x.copy_(keep)                                  # the capture itself does not run the step; leave x as handed in
        return g

    def profile(self, on: bool):
        """Record HIP events between the launches of ``step`` (per-stage kernel time, bench.py)."""
        _check(self.lib.skpangu_profile(self._ctx, 1 if on else 0), "skpangu_profile")

    def profile_read(self) -> list[dict]:
        arr = (SkStageStat * 32)()
        n = ctypes.c_int()
        _check(self.lib.skpangu_profile_read(self._ctx, arr, 32, ctypes.byref(n)), "skpangu_profile_read")
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n.value)]

    # ---- stage-level (tests) ------------------------------------------- #
    def tokens(self, layer: int):
        return self.geom.tokens(layer), self.geom.dim(layer)

    def patch_embed(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(self.tokens(1), dtype=torch.float32, device=self.device)
        self._chk_dev(x, self.state_shape)
        ops.hip.pangu_patch_embed(self._ctx.value, x, out)
        return out

    def block(self, layer: int, i: int, x: torch.Tensor) -> torch.Tensor:
        y = x.contiguous().clone()
        self._chk_dev(y, self.tokens(layer))
        ops.hip.pangu_block(self._ctx.value, layer, i, y)
        return y

    def downsample(self, x1: torch.Tensor) -> torch.Tensor:
        out = torch.empty(self.tokens(2), dtype=torch.float32, device=self.device)
        x1 = x1.contiguous()
        self._chk_dev(x1, self.tokens(1))
        ops.hip.pangu_downsample(self._ctx.value, x1, out)
        return out

    def upsample(self, x2: torch.Tensor) -> torch.Tensor:
        out = torch.empty(self.tokens(1), dtype=torch.float32, device=self.device)
        x2 = x2.contiguous()
        self._chk_dev(x2, self.tokens(2))
        ops.hip.pangu_upsample(self._ctx.value, x2, out)
        return out

    def patch_recover(self, skip: torch.Tensor, x4: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.state_shape, dtype=torch.float32, device=self.device)
        skip, x4 = skip.contiguous(), x4.contiguous()
        self._chk_dev(skip, self.tokens(1)), self._chk_dev(x4, self.tokens(1))
        ops.hip.pangu_patch_recover(self._ctx.value, skip, x4, out)
        return out

    def debug_buffer(self, name: str, dtype: torch.dtype) -> torch.Tensor:
        """Copy of an internal buffer as a flat tensor of ``dtype`` (tests only)."""
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        _check(self.lib.skpangu_debug_buffer(self._ctx, name.encode(), ctypes.byref(ptr), ctypes.byref(nbytes)), "skpangu_debug_buffer")
        torch.cuda.synchronize(self.device)
        for base in (self._workspace, self._prepared):
            off = ptr.value - base.data_ptr()
            if 0 <= off < base.numel():
                return base[off:off + nbytes.value].clone().view(dtype)
        raise RuntimeError("debug pointer outside the engine arenas")
