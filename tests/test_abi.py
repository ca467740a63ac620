"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/skyrim_pangu.h
declares, and its host-side planning (sizes, parameter table, argument errors) works without a GPU."""
import ctypes
import re
from pathlib import Path

import pytest

from skyrim_amd.pangu import engine as E
from skyrim_amd.pangu.spec import PanguGeometry, param_offsets

HEADER = Path(__file__).resolve().parent.parent / "include" / "skyrim_pangu.h"


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skpangu_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in skyrim_pangu.h but not exported"
    assert set(syms) == set(E.EXPORTS)
    assert lib.skpangu_abi_version() == 6


@pytest.mark.parametrize("prec", ["bf16x3", "f16x3", "f16x3q", "f16x2m", "f16x2c", "f16x1m"])
@pytest.mark.parametrize("grid", [(49, 192), (721, 1440)])
def test_param_table_matches_host_spec(grid, prec):
    g = PanguGeometry(*grid)
    table = E.param_table(g, prec)
    offs, total = param_offsets(g)
    sizes = E.query_sizes(g, prec)
    assert sizes.master_floats == total and sizes.n_params == len(offs)
    assert sizes.state_floats == 69 * grid[0] * grid[1]
    for name, off, shape in table:
        assert offs[name] == (off, shape)


def test_sizes_full_grid_fit_one_gpu():
    s = E.query_sizes(PanguGeometry(721, 1440))
    assert s.prepared_bytes + s.workspace_bytes < 16 * 2 ** 30      # a few GB of the 288 GB


def test_term_plan_sizes_and_validation():
    """skpangu_config.term_plan (ABI v3): the two-term layers prepare ONE weight plane in fragment order, so the prepared arena shrinks
    with every bit; the plan is refused outside the fp16-plane modes / fused kernels and beyond its 8 bits."""
    g = PanguGeometry(721, 1440)
    lib = E.load_library()
    sizes = {}
    for prec, plan in (("f16x3q", 0), ("f16x2", 0x0F), ("f16x2q", 0xFF)):
        sizes[prec] = E.query_sizes(g, "f16x3q", E.make_config(g, "f16x3q", term_plan=plan)).prepared_bytes
    assert sizes["f16x2q"] < sizes["f16x2"] < sizes["f16x3q"]
    assert E.make_config(g, "f16x3q").term_plan == 0 and E.make_config(g, "f16x2m").term_plan == 0x6F and E.make_config(g, "f16x2c").term_plan == 0x66
    assert E.DEFAULT_PRECISION == "f16x1m" and E.make_config(g).term_plan == 0x66F
    assert E.make_config(g, "f16x3q", term_plan=0x3).term_plan == 3
    with pytest.raises(ValueError):
        E.make_config(g, "bf16x3", term_plan=1)
    with pytest.raises(ValueError):
        E.make_config(g, "f16x2m", mlp="split")
    out = E.SkSizes()
    bad = E.make_config(g, "f16x2m")
    for plan, ok in ((0x100, False), (0x1000, False), (0x210, False), (0x101, True), (0x66F, True), (0xFFF, True)):
        bad.term_plan = plan                  # bits 8-11 (one-term block GEMMs) need the layer's two-term bit; nothing beyond 12 bits
        assert (lib.skpangu_query_sizes(ctypes.byref(bad), ctypes.byref(out)) == 0) == ok, hex(plan)
    assert E.make_config(g, "f16x1m").term_plan == 0x66F
    assert E.query_sizes(g, "f16x1m").prepared_bytes == E.query_sizes(g, "f16x2m").prepared_bytes      # same weights, fewer MFMAs


def test_bad_arguments_are_errors_not_crashes():
    lib = E.load_library()
    out = E.SkSizes()
    for cfg in (E.SkConfig(721, 1000, 0), E.SkConfig(4, 1440, 0), E.SkConfig(721, 1440, 7)):
        assert lib.skpangu_query_sizes(ctypes.byref(cfg), ctypes.byref(out)) == -1
    assert lib.skpangu_query_sizes(None, ctypes.byref(out)) == -1
    assert lib.skpangu_step(None, None, None, None) == -1
    assert lib.skpangu_prepare(None, None, None) == -1
    assert b"geometry" in lib.skpangu_error_string(-1)
    with pytest.raises(ValueError):
        PanguGeometry(721, 1000)


def test_engine_refuses_to_run_without_gpu_or_library(monkeypatch, tmp_path):
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            E.PanguEngine(PanguGeometry(49, 192))
    monkeypatch.setattr(E, "_lib", None)
    monkeypatch.setenv("SKYRIM_PANGU_LIB", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="not found"):
        E.load_library()


def test_custom_ops_are_registered_and_have_no_cpu_kernel():
    """the Python call path is torch.library custom ops (skyrim_hip::*) over the C ABI; only the CUDA (ROCm) key
    has an implementation, so CPU tensors fail in the dispatcher instead of falling back."""
    import torch
    from skyrim_amd import ops
    assert {"pangu_step", "sfno_gemm", "sfno_instance_norm", "gc_gather_gemm", "gc_linear_layer_norm", "gc_segment_sum"} <= set(ops.OP_NAMES)
    for name in ops.OP_NAMES:
        assert hasattr(torch.ops.skyrim_hip, name)
    with pytest.raises(NotImplementedError):
        torch.ops.skyrim_hip.pangu_step(0, torch.zeros(4), torch.zeros(4))
    with pytest.raises(NotImplementedError):
        torch.ops.skyrim_hip.gc_segment_sum(torch.zeros(4, 8), torch.zeros(3, dtype=torch.int32), torch.zeros(2, 8), None, 2, 8)
    schema = torch.ops.skyrim_hip.pangu_step.default._schema
    assert str(schema) == "skyrim_hip::pangu_step(int ctx, Tensor x, Tensor(a!) out) -> ()"
    ops.register()          # idempotent


def test_io_library_exports_every_declared_symbol():
    """include/skyrim_io.h (the delivery helpers of the save path): every declared entry point is exported, versions agree."""
    from skyrim_amd import deliver as D
    text = re.sub(r"/\*.*?\*/", "", (HEADER.parent / "skyrim_io.h").read_text(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(skio_[a-z0-9_]+)\s*\(", text)))
    lib = D.load_library()
    assert syms == sorted(D.EXPORTS) and len(syms) == 2
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in skyrim_io.h but not exported"
    assert lib.skio_abi_version() == D.ABI_VERSION == int(re.search(r"SKIO_ABI_VERSION (\d+)", text).group(1))
    assert lib.skio_bswap32(None, None, 0, None) == 0 and lib.skio_bswap32(None, None, 4, None) == -1      # argument checks run without a GPU
