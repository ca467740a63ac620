"""Real-weight ingestion: the dependency-free ONNX initializer reader and the slot mapper, checked
against a synthetic Pangu-shaped ONNX file written here with a minimal protobuf encoder (the real pangu_weather_6.onnx
cannot be obtained in this environment: parity with it is unpinned, see skyrim_amd/pangu/onnx_weights.py)."""
import json
import random
import struct

import numpy as np
import pytest

from skyrim_amd.pangu import onnx_weights as OW
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, param_spec


# ---- minimal protobuf encoder (wire format only) ---------------------------------------------- #
def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(fno, payload):
    return _vi(fno << 3 | 2) + _vi(len(payload)) + payload


def _tensor(name, arr, storage="raw"):
    arr = np.asarray(arr, order="C")
    code = {np.dtype("float32"): 1, np.dtype("float16"): 10, np.dtype("int64"): 7, np.dtype("float64"): 11}[arr.dtype]
    msg = b"".join(_vi(1 << 3 | 0) + _vi(d) for d in arr.shape) if storage != "packed_dims" else _ld(1, b"".join(_vi(d) for d in arr.shape))
    msg += _vi(2 << 3 | 0) + _vi(code) + _ld(8, name.encode())
    if storage == "float_data" and code == 1:
        msg += _ld(4, arr.astype("<f4").tobytes())
    elif storage == "int64_data" and code == 7:
        msg += _ld(7, b"".join(_vi(int(v) & (2 ** 64 - 1)) for v in arr.reshape(-1)))
    else:
        msg += _ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    return msg


def _node(op, inputs, outputs):
    return b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs) + _ld(4, op.encode())


def write_onnx(path, tensors, nodes, file_order_seed=0):
    """tensors: {name: (array, storage)}; nodes: [(op, inputs, outputs)]; initializers are written in shuffled order."""
    names = list(tensors)
    random.Random(file_order_seed).shuffle(names)
    graph = b"".join(_ld(1, _node(*n)) for n in nodes) + _ld(2, b"main_graph")
    graph += b"".join(_ld(5, _tensor(n, *tensors[n])) for n in names)
    model = _vi(1 << 3 | 0) + _vi(8) + _ld(2, b"test-writer") + _ld(7, graph)     # ir_version, producer_name, graph
    with open(path, "wb") as f:
        f.write(model)


def pangu_like_onnx(path, geom, params):
    """A graph that uses the Pangu parameters in forward order under opaque names, Linear weights transposed for MatMul."""
    tensors, nodes, truth = {}, [], {}
    counter = [0]

    def add(slot, op, how="id", storage="raw"):
        counter[0] += 7
        name = f"onnx::{op}_{counter[0]}" if how == "T" else f"m.{counter[0]}.{slot.split('.')[-1]}"
        a = params[slot].numpy()
        tensors[name] = (a.T.copy() if how == "T" else a, storage)
        nodes.append((op, [f"t{len(nodes)}", name], [f"t{len(nodes) + 1}"]))
        truth[slot] = name

    add("norm.mean", "Sub"); add("norm.std", "Div"); add("const_masks", "Concat", storage="packed_dims")
    add("embed.conv.weight", "Conv"); add("embed.conv.bias", "Conv", storage="float_data")
    add("embed.conv_surface.weight", "Conv"); add("embed.conv_surface.bias", "Conv")
    tensors["shape_const_0"] = (np.array([1, -1, 192], dtype=np.int64), "int64_data")
    nodes.append(("Reshape", ["t", "shape_const_0"], ["t_r"]))
    tensors["eps"] = (np.array(1e-5, dtype=np.float32), "raw")

    def block(prefix):
        add(prefix + "attn.qkv.weight", "MatMul", "T"); add(prefix + "attn.qkv.bias", "Add")
        add(prefix + "attn.bias_table", "Gather")
        add(prefix + "attn.proj.weight", "MatMul", "T"); add(prefix + "attn.proj.bias", "Add")
        nodes.append(("Add", ["x", "eps"], ["x_eps"]))
        add(prefix + "norm1.weight", "Mul"); add(prefix + "norm1.bias", "Add", storage="float_data")
        add(prefix + "mlp.fc1.weight", "MatMul", "T"); add(prefix + "mlp.fc1.bias", "Add")
        add(prefix + "mlp.fc2.weight", "MatMul", "T"); add(prefix + "mlp.fc2.bias", "Add")
        add(prefix + "norm2.weight", "Mul"); add(prefix + "norm2.bias", "Add")

    for layer, depth in ((1, 2), (2, 6)):
        for i in range(depth):
            block(f"layer{layer}.block{i}.")
        if layer == 1:
            add("down.norm.weight", "Mul"); add("down.norm.bias", "Add"); add("down.linear.weight", "MatMul", "T")
    for i in range(6):
        block(f"layer3.block{i}.")
    add("up.linear1.weight", "MatMul", "T"); add("up.norm.weight", "Mul"); add("up.norm.bias", "Add"); add("up.linear2.weight", "MatMul", "T")
    for i in range(2):
        block(f"layer4.block{i}.")
    add("recover.conv.weight", "ConvTranspose"); add("recover.conv.bias", "ConvTranspose")
    add("recover.conv_surface.weight", "ConvTranspose"); add("recover.conv_surface.bias", "ConvTranspose")
    write_onnx(path, tensors, nodes)
    return truth


@pytest.fixture(scope="module")
def toy_onnx(tmp_path_factory):
    g = PanguGeometry(49, 192)
    params = init_synthetic(g, 3)
    path = tmp_path_factory.mktemp("onnx") / "pangu_toy.onnx"
    truth = pangu_like_onnx(path, g, params)
    return g, params, path, truth


def test_reader_recovers_every_initializer(toy_onnx):
    g, params, path, truth = toy_onnx
    m = OW.read_model(path)
    assert len(m.nodes) > 200 and len(m.initializers) == len(truth) + 2
    for slot, name in truth.items():
        a = m.initializers[name].array()
        want = params[slot].numpy()
        assert a.dtype == np.float32
        assert np.array_equal(a, want.T if name.startswith("onnx::MatMul") else want), slot
    assert m.initializers["shape_const_0"].array().tolist() == [1, -1, 192]
    assert m.initializers["eps"].dims == () and m.initializers["eps"].array() == np.float32(1e-5)
    use = [t.name for t, _ in m.in_order_of_use()]
    assert use.index(truth["norm.mean"]) == 0 and use.index(truth["layer1.block0.attn.qkv.weight"]) < use.index(truth["layer1.block0.attn.bias_table"])


def test_auto_map_and_convert_reproduce_the_parameters(toy_onnx):
    g, params, path, truth = toy_onnx
    mapping, unresolved = OW.auto_map(OW.read_model(path), g)
    assert unresolved == []
    assert {s: v[0] for s, v in mapping.items()} == truth
    assert mapping["layer1.block0.attn.proj.weight"][1] == "T"          # square weight: decided by the consuming MatMul
    with pytest.raises(ValueError, match="no slot mapping given"):
        OW.convert(path, g)                                                # the shape-and-order mapping is never applied silently
    got = OW.convert(path, g, allow_auto=True)
    assert list(got) == [s for s, _ in param_spec(g)]
    for slot, _ in param_spec(g):
        assert np.array_equal(got[slot], params[slot].numpy()), slot


def test_explicit_mapping_extra_arrays_and_errors(toy_onnx, tmp_path):
    g, params, path, truth = toy_onnx
    mapping = {s: [n, "T" if n.startswith("onnx::MatMul") else "id"] for s, n in truth.items() if not s.startswith("norm.")}
    with pytest.raises(ValueError, match="unresolved"):
        OW.convert(path, g, mapping)                                       # a partially loaded network must not run
    got = OW.convert(path, g, mapping, extra={"norm.mean": params["norm.mean"].numpy(), "norm.std": params["norm.std"].numpy()})
    assert np.array_equal(got["layer2.block3.mlp.fc2.weight"], params["layer2.block3.mlp.fc2.weight"].numpy())
    bad = dict(mapping, **{"norm.mean": "nope", "norm.std": truth["norm.std"]})
    with pytest.raises(KeyError):
        OW.convert(path, g, bad)
    wrong = dict(mapping, **{"norm.mean": [truth["embed.conv.bias"], "id"], "norm.std": truth["norm.std"]})     # 192 values for a 69-slot
    with pytest.raises(ValueError):
        OW.convert(path, g, wrong)
    (tmp_path / "junk.onnx").write_bytes(struct.pack("<I", 0x0A020801) * 4)
    with pytest.raises(ValueError):
        OW.read_model(tmp_path / "junk.onnx")


def test_cli_inspect_and_automap(toy_onnx, capsys):
    g, params, path, truth = toy_onnx
    assert OW.main(["inspect", str(path)]) == 0
    lines = capsys.readouterr().out.splitlines()
    assert len(lines) == len(truth) + 2 and "MatMul" in "".join(lines)
    OW.main(["automap", str(path)])       # maps the toy file against the full-size geometry: shapes with lat/lon do not fit
    rep = json.loads(capsys.readouterr().out)
    assert "const_masks" in rep["unresolved"] and "layer1.block0.attn.qkv.weight" in rep["mapping"]


def test_validation_catches_a_swapped_gain_and_bias(toy_onnx):
    """A fully resolved mapping can still be wrong: norm1.weight and norm1.bias (and proj.bias, fc2.bias, norm2.*) share the shape
    (C,).  Swapping two of them passes every shape check; ``validate`` refuses the result (onnx_weights.py:230)."""
    g, params, path, truth = toy_onnx
    mapping = {s: [n, "T" if n.startswith("onnx::MatMul") else "id"] for s, n in truth.items()}
    ok = OW.convert(path, g, mapping)
    assert OW.validate(ok) == []
    swapped = dict(mapping)
    swapped["layer1.block0.norm1.weight"], swapped["layer1.block0.norm1.bias"] = mapping["layer1.block0.norm1.bias"], mapping["layer1.block0.norm1.weight"]
    with pytest.raises(ValueError, match="look wrong"):
        OW.convert(path, g, swapped)
    bad = dict(ok)
    bad["layer2.block1.attn.bias_table"] = ok["layer2.block1.attn.bias_table"] * np.float32(np.inf)
    assert any("non-finite" in b for b in OW.validate(bad))
