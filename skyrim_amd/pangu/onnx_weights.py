"""Real-weight ingestion: dependency-free reader of ONNX initializers -> Pangu engine parameter slots.

The reference obtains the Pangu weights as ``pangu_weather_{6,24}.onnx`` through earth2mip and runs them under
onnxruntime (/root/reference/skyrim/core/models/pangu.py:45-46); neither package nor the files exist in this
environment.  The HIP engine wants the same numbers as an fp32 blob laid out by
``skpangu_param_info``.  This module gets them out of the ``.onnx`` file without onnx / protobuf:

* ``read_model(path)``      - walks the protobuf wire format (ModelProto.graph -> initializers + nodes), zero-copy
                              over an mmap: a 1.2 GB file is not read into Python objects;
* ``inspect(path)``         - one line per initializer in order of first use by a node (name, dtype, shape, op);
* ``convert(path, geom, mapping)`` - fills the slots of ``spec.param_spec`` from an explicit mapping
                              {slot: onnx_name | [onnx_name, "T" | "perm:2,0,1" | "reshape"]};
* ``auto_map(model, geom)`` - best-effort mapping by shape in order of use (Linear weights are stored transposed by
                              torch.onnx's MatMul export), with a report of every slot it could not resolve.

PARITY UNPINNED: the structure of the real files could not be inspected here, so ``auto_map`` is verified only
against a synthetic ONNX file that tests/test_onnx_weights.py writes with its own protobuf encoder.  With the real
file, run ``python -m skyrim_amd.pangu.onnx_weights inspect pangu_weather_6.onnx`` and, if the automatic report
lists unresolved slots, pass an explicit JSON mapping.
"""
from __future__ import annotations

import json
import mmap
import sys
from dataclasses import dataclass, field

import numpy as np

from .spec import PanguGeometry, param_spec

# TensorProto.DataType -> numpy
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf, start=0, end=None):
    """Yield (field_number, wire_type, value) of one protobuf message; length-delimited values are memoryviews."""
    pos, end = start, len(buf) if end is None else end
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt} at byte {pos}")
        yield fno, wt, val


def _packed_varints(val, wt):
    if wt == 0:
        return [val]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


@dataclass
class Initializer:
    name: str
    dtype: int
    dims: tuple
    raw: object = None            # memoryview of raw_data, or None
    typed: list = field(default_factory=list)      # (field_no, wire_type, payload) of float_data / int32_data / int64_data / double_data
    external: dict = field(default_factory=dict)

    def array(self) -> np.ndarray:
        if self.external:
            raise ValueError(f"{self.name}: external_data initializers are not supported (re-export with raw data)")
        dt = _DTYPES.get(self.dtype)
        if dt is None:
            raise ValueError(f"{self.name}: unsupported ONNX data type {self.dtype}")
        n = int(np.prod(self.dims)) if self.dims else 1
        if self.raw is not None:
            a = np.frombuffer(self.raw, dtype=np.dtype(dt).newbyteorder("<"), count=n)
        else:
            parts = []
            for fno, wt, payload in self.typed:
                if fno == 4:       # float_data
                    parts.append(np.frombuffer(payload, dtype="<f4"))
                elif fno == 10:    # double_data
                    parts.append(np.frombuffer(payload, dtype="<f8"))
                else:              # int32_data (5) / int64_data (7): varints (float16 is stored as uint16 bit patterns in int32_data)
                    v = np.array([x - (1 << 64) if x >> 63 else x for x in _packed_varints(payload, wt)], dtype=np.int64)   # two's complement
                    parts.append(v.astype(np.uint16).view(np.float16) if self.dtype == 10 else v)
            a = np.concatenate(parts) if parts else np.zeros(0, dt)
            a = a.astype(dt, copy=False)
        if a.size != n:
            raise ValueError(f"{self.name}: {a.size} elements for shape {self.dims}")
        return a.reshape(self.dims)


@dataclass
class Node:
    op: str
    inputs: list
    outputs: list
    name: str = ""


@dataclass
class Model:
    initializers: dict      # name -> Initializer (file order)
    nodes: list
    _mm: object = None      # keeps the mmap alive

    def in_order_of_use(self) -> list:
        """Initializers ordered by the first node that consumes them (forward order of the network), with that node's op."""
        seen, out = set(), []
        for nd in self.nodes:
            for i in nd.inputs:
                if i in self.initializers and i not in seen:
                    seen.add(i)
                    out.append((self.initializers[i], nd.op))
        out += [(t, "") for n, t in self.initializers.items() if n not in seen]
        return out


def _parse_tensor(buf) -> Initializer:
    t = Initializer("", 0, ())
    dims = []
    for fno, wt, val in _fields(buf):
        if fno == 1:
            dims += _packed_varints(val, wt)
        elif fno == 2:
            t.dtype = val
        elif fno == 8:
            t.name = bytes(val).decode()
        elif fno == 9:
            t.raw = val
        elif fno in (4, 5, 7, 10):
            t.typed.append((fno, wt, val if wt == 2 else val))
        elif fno == 13:
            kv = {f: bytes(v).decode() for f, _, v in _fields(val)}
            t.external[kv.get(1, "")] = kv.get(2, "")
    t.dims = tuple(int(d) for d in dims)
    return t


def _parse_node(buf) -> Node:
    nd = Node("", [], [])
    for fno, wt, val in _fields(buf):
        if fno == 1:
            nd.inputs.append(bytes(val).decode())
        elif fno == 2:
            nd.outputs.append(bytes(val).decode())
        elif fno == 3:
            nd.name = bytes(val).decode()
        elif fno == 4:
            nd.op = bytes(val).decode()
    return nd


def read_model(path) -> Model:
    """ModelProto (field 7 = graph) -> GraphProto (1 = node, 5 = initializer)."""
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    buf = memoryview(mm)
    inits, nodes = {}, []
    for fno, wt, val in _fields(buf):
        if fno == 7 and wt == 2:
            for gno, gwt, gval in _fields(val):
                if gno == 5 and gwt == 2:
                    t = _parse_tensor(gval)
                    inits[t.name] = t
                elif gno == 1 and gwt == 2:
                    nodes.append(_parse_node(gval))
    if not inits:
        raise ValueError(f"{path}: no graph initializers found (not an ONNX ModelProto?)")
    return Model(inits, nodes, mm)


def inspect(path) -> list[str]:
    m = read_model(path)
    return [f"{t.name:48s} {np.dtype(_DTYPES.get(t.dtype, np.void)).name:8s} {str(t.dims):24s} {op}" for t, op in m.in_order_of_use()]


# ---- mapping ------------------------------------------------------------------------------------- #
def _apply(a: np.ndarray, how: str, shape: tuple) -> np.ndarray:
    if how in ("", "id"):
        pass
    elif how == "T":
        a = a.T
    elif how.startswith("perm:"):
        a = a.transpose([int(i) for i in how[5:].split(",")])
    elif how == "reshape":
        a = a.reshape(shape)
    else:
        raise ValueError(f"unknown transform {how!r}")
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"shape {a.shape} after {how!r}, slot wants {shape}")
    return np.ascontiguousarray(a, dtype=np.float32)


def _transform_for(onnx_shape: tuple, slot_shape: tuple, op: str = ""):
    """How an initializer of ``onnx_shape`` consumed by node type ``op`` can fill a slot of ``slot_shape`` (None if not)."""
    if len(slot_shape) == 2 and op == "MatMul":
        # torch Linear [out, in] is exported as MatMul's right operand [in, out]; for square weights only the op tells
        return "T" if tuple(onnx_shape) == tuple(slot_shape[::-1]) else None
    if tuple(onnx_shape) == tuple(slot_shape):
        return "id"
    if len(slot_shape) == 2 and tuple(onnx_shape) == tuple(slot_shape[::-1]):
        return "T"
    if len(slot_shape) == 3 and sorted(onnx_shape) == sorted(slot_shape) and len(set(slot_shape)) == 3:
        return "perm:" + ",".join(str(onnx_shape.index(d)) for d in slot_shape)
    if len(onnx_shape) != len(slot_shape) and int(np.prod(onnx_shape)) == int(np.prod(slot_shape)) and len(slot_shape) >= 3:
        return "reshape"         # e.g. conv weights flattened by constant folding
    return None


def auto_map(model: Model, geom: PanguGeometry, window: int = 16) -> tuple[dict, list]:
    """Best-effort {slot: [onnx_name, transform]}: walk the initializers in order of use and give each to the earliest
    still-empty slot (looking at most ``window`` slots ahead of the first empty one) that its shape can fill.  Returns the
    mapping and the list of unresolved slots.  Float tensors only; scalars and shape constants are skipped."""
    slots = param_spec(geom)
    mapping, taken = {}, [False] * len(slots)
    first = 0
    for t, op in model.in_order_of_use():
        if t.dtype not in (1, 10, 11) or len(t.dims) == 0 or int(np.prod(t.dims)) < 4:
            continue
        while first < len(slots) and taken[first]:
            first += 1
        for j in range(first, min(first + window, len(slots))):
            if taken[j]:
                continue
            how = _transform_for(t.dims, slots[j][1], op)
            if how is not None:
                mapping[slots[j][0]] = [t.name, how]
                taken[j] = True
                break
    return mapping, [s for (s, _), tk in zip(slots, taken) if not tk]


def validate(arrays: dict) -> list[str]:
    """Plausibility of a converted parameter set, slot class by slot class -- what a WRONG slot assignment (several slots of a block
    share a shape: proj.bias, norm1.weight, norm1.bias, fc2.bias, norm2.* are all (C,)) or a wrong layout would break:
    LayerNorm gains centred near 1 and positive, LayerNorm / linear biases centred near 0, weight matrices with a small non-zero
    spread, finite bias tables, positive normalisation stds.  Returns the list of complaints (empty = plausible).  It cannot prove a
    mapping right: the qkv split order and the bias_table index layout have only been checked against a synthetic Pangu-shaped file."""
    bad = []
    for name, a in arrays.items():
        a = np.asarray(a, dtype=np.float64)
        if not np.isfinite(a).all():
            bad.append(f"{name}: non-finite values")
            continue
        if name.endswith(("norm.weight", "norm1.weight", "norm2.weight")):
            if not (0.2 < a.mean() < 5.0) or (a <= 0).mean() > 0.1:
                bad.append(f"{name}: LayerNorm gain with mean {a.mean():.3g}, {100 * (a <= 0).mean():.0f}% non-positive (a bias in a gain slot?)")
        elif name.endswith(("norm.bias", "norm1.bias", "norm2.bias")) or (name.endswith(".bias") and a.ndim == 1 and not name.startswith("norm.")):
            if abs(a.mean()) > 0.5 and abs(a.mean()) > 3 * a.std():
                bad.append(f"{name}: bias centred at {a.mean():.3g} (a LayerNorm gain in a bias slot?)")
        elif name.endswith(".weight") and a.ndim >= 2:
            if not (1e-5 < a.std() < 2.0):
                bad.append(f"{name}: weight spread {a.std():.3g}")
        elif name.endswith("bias_table") and np.abs(a).max() > 100.0:
            bad.append(f"{name}: |entry| up to {np.abs(a).max():.3g}")
        elif name == "norm.std" and (a <= 0).any():
            bad.append("norm.std: non-positive entries")
    return bad


def convert(path, geom: PanguGeometry, mapping: dict | None = None, extra: dict | None = None, allow_auto: bool | None = None) -> dict:
    """-> {slot: float32 ndarray} for every slot of ``param_spec(geom)``.  ``mapping``: {slot: onnx name | [name, transform]}, written
    by hand or produced by ``auto_map`` and REVIEWED (``python -m skyrim_amd.pangu.onnx_weights automap FILE > FILE.map.json``);
    ``extra`` supplies arrays for slots that are not in the file (e.g. normalisation stats).  Without a mapping the automatic one is
    used only when ``allow_auto`` (or SKYRIM_ONNX_AUTOMAP=1) says so: it assigns by shape and order of use, and a block has many
    slots of identical shape, so a fully resolved mapping can still be wrong.  Every result is checked by ``validate``.
    Raises if any slot stays empty: a partially loaded network must not run."""
    import os
    model = read_model(path)
    unresolved = []
    if mapping is None:
        if allow_auto is None:
            allow_auto = os.environ.get("SKYRIM_ONNX_AUTOMAP") == "1"
        if not allow_auto:
            raise ValueError(f"{path}: no slot mapping given.  Write one with `python -m skyrim_amd.pangu.onnx_weights automap {path} > {path}.map.json`, "
                             "review it against `... inspect`, and load again (PanguTimeLoop reads <file>.map.json); or opt into the unreviewed "
                             "automatic mapping with SKYRIM_ONNX_AUTOMAP=1 / allow_auto=True")
        mapping, unresolved = auto_map(model, geom)
    out = {}
    for slot, shape in param_spec(geom):
        if extra and slot in extra:
            out[slot] = _apply(np.asarray(extra[slot]), "id", shape)
        elif slot in mapping:
            ent = mapping[slot]
            name, how = (ent, "id") if isinstance(ent, str) else (ent[0], ent[1])
            if name not in model.initializers:
                raise KeyError(f"{slot}: initializer {name!r} not in {path}")
            out[slot] = _apply(model.initializers[name].array(), how, shape)
    missing = [s for s, _ in param_spec(geom) if s not in out]
    if missing:
        raise ValueError(f"{len(missing)} parameter slots unresolved (first: {missing[:6]}); "
                         f"inspect the file and pass an explicit mapping / extra arrays")
    bad = validate(out)
    if bad:
        raise ValueError(f"{path}: the converted parameters look wrong in {len(bad)} slot(s) -- a mis-assigned mapping? " + "; ".join(bad[:6]))
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) >= 2 and argv[0] == "inspect":
        print("\n".join(inspect(argv[1])))
        return 0
    if len(argv) >= 2 and argv[0] == "automap":
        mapping, unresolved = auto_map(read_model(argv[1]), PanguGeometry(721, 1440))
        print(json.dumps({"mapping": mapping, "unresolved": unresolved}, indent=1))
        return 0 if not unresolved else 1
    print("usage: python -m skyrim_amd.pangu.onnx_weights inspect|automap FILE.onnx", file=sys.stderr)
    return 2


if __name__ == "__main__":
    raise SystemExit(main())
