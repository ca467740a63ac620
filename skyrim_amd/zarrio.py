"""Zarr-v2 directory store of a labelled array, written by hand (zarr is not installed here).

Layout = what ``xarray.DataArray.to_zarr(path, consolidated=True)`` produces
(/root/reference/skyrim/common.py:146-159,176-191): a group with one array per coordinate and one for
the payload, ``_ARRAY_DIMENSIONS`` attributes, CF-encoded time, ``.zmetadata`` consolidation.  Chunks are
uncompressed (``compressor: null``) C-order little-endian files, one chunk per (time entry, channel).
``append_dim`` appends new chunks along that dimension (the reference's intended per-step append;
its local branch names a non-existent "step" dim, common.py:148-153 -- see SURVEY.md 3.5).
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from .ncio import UNNAMED


def _zarray(shape, chunks, dtype):
    return {"zarr_format": 2, "shape": list(map(int, shape)), "chunks": list(map(int, chunks)), "dtype": dtype,
            "compressor": None, "fill_value": None, "order": "C", "filters": None}


def _write_chunks(root: Path, name: str, arr: np.ndarray, chunks, offset_blocks=None):
    d = root / name
    d.mkdir(parents=True, exist_ok=True)
    if arr.ndim == 0:
        (d / "0").write_bytes(arr.tobytes())
        return
    grid = [int(np.ceil(s / c)) for s, c in zip(arr.shape, chunks)]
    offset_blocks = offset_blocks or [0] * arr.ndim
    for idx in np.ndindex(*grid):
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        block = np.zeros(chunks, dtype=arr.dtype)
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
        key = ".".join(str(i + o) for i, o in zip(idx, offset_blocks))
        (d / key).write_bytes(np.ascontiguousarray(block).tobytes())


def _encode(vals: np.ndarray, units: str | None = None):
    """-> (array to store, zarr dtype string, attrs)"""
    if np.issubdtype(vals.dtype, np.datetime64):
        if units is None:
            t0 = vals.reshape(-1)[0].astype("datetime64[s]")
            units = f"hours since {str(t0).replace('T', ' ')}"
        t0 = np.datetime64(units.partition(" since ")[2].replace(" ", "T"), "s")
        enc = ((vals.astype("datetime64[s]") - t0).astype("timedelta64[s]").astype(np.int64) // 3600).astype("<i8")
        return enc, "<i8", {"units": units, "calendar": "proleptic_gregorian"}
    if vals.dtype.kind in "US":
        u = vals.astype(str)
        width = max(1, max((len(s) for s in u.reshape(-1)), default=1))
        return u.astype(f"<U{width}"), f"<U{width}", {}
    if vals.dtype.kind == "f":
        return vals.astype("<f8"), "<f8", {}
    if vals.dtype.kind in "iu":
        return vals.astype("<i8"), "<i8", {}
    raise TypeError(vals.dtype)


def _consolidate(root: Path):
    meta = {}
    for p in sorted(root.rglob(".z*")):
        if p.name in (".zarray", ".zattrs", ".zgroup"):
            meta[str(p.relative_to(root))] = json.loads(p.read_text())
    (root / ".zmetadata").write_text(json.dumps({"metadata": meta, "zarr_consolidated_format": 1}, indent=1))


def write_dataarray_zarr(da, store, mode="w", append_dim=None, consolidated=True):
    root = Path(store)
    name = da.name or UNNAMED
    payload = da.values.astype("<f4") if da.values.dtype != np.float64 else da.values.astype("<f8")
    pdtype = "<f4" if payload.dtype == np.dtype("<f4") else "<f8"
    chunks = tuple(1 if d in ("time", "channel") else n for d, n in zip(da.dims, da.shape))
    if mode == "a" and append_dim is not None and (root / name / ".zarray").exists():
        if append_dim not in da.dims:
            raise ValueError(f"append_dim {append_dim!r} is not a dimension of the array {da.dims}")
        ax = da.dims.index(append_dim)
        za = json.loads((root / name / ".zarray").read_text())
        old = za["shape"][ax]
        if [s for i, s in enumerate(za["shape"]) if i != ax] != [s for i, s in enumerate(payload.shape) if i != ax]:
            raise ValueError("appended array does not match the store's shape")
        off = [0] * payload.ndim
        off[ax] = old // za["chunks"][ax]
        _write_chunks(root, name, payload, za["chunks"], off)
        za["shape"][ax] = old + payload.shape[ax]
        (root / name / ".zarray").write_text(json.dumps(za))
        cz = json.loads((root / append_dim / ".zarray").read_text())
        cattrs = json.loads((root / append_dim / ".zattrs").read_text())
        enc, _, _ = _encode(da._coords[append_dim], cattrs.get("units"))
        _write_chunks(root, append_dim, enc.astype(cz["dtype"]), cz["chunks"], [cz["shape"][0] // cz["chunks"][0]])
        cz["shape"][0] += enc.shape[0]
        (root / append_dim / ".zarray").write_text(json.dumps(cz))
    else:
        root.mkdir(parents=True, exist_ok=True)
        (root / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
        (root / ".zattrs").write_text("{}")
        for cname, vals in da._coords.items():
            enc, zdtype, attrs = _encode(vals)
            cdims = [cname] if (cname in da.dims and vals.ndim == 1) else []
            cchunks = [1] if cname == append_dim or cname == "time" else list(enc.shape)
            _write_chunks(root, cname, enc, cchunks if cdims else [])
            (root / cname / ".zarray").write_text(json.dumps(_zarray(enc.shape, cchunks if cdims else [], zdtype)))
            attrs["_ARRAY_DIMENSIONS"] = cdims
            (root / cname / ".zattrs").write_text(json.dumps(attrs))
        _write_chunks(root, name, payload, chunks)
        (root / name / ".zarray").write_text(json.dumps(_zarray(payload.shape, chunks, pdtype)))
        pattrs = {"_ARRAY_DIMENSIONS": list(da.dims)}
        extra = " ".join(k for k in da._coords if k not in da.dims)
        if extra:
            pattrs["coordinates"] = extra
        (root / name / ".zattrs").write_text(json.dumps(pattrs))
    if consolidated:
        _consolidate(root)


def _read_array(root: Path, name: str):
    za = json.loads((root / name / ".zarray").read_text())
    attrs = json.loads((root / name / ".zattrs").read_text()) if (root / name / ".zattrs").exists() else {}
    if za["compressor"] is not None:
        raise NotImplementedError("compressed zarr chunks need the zarr package")
    dt = np.dtype(za["dtype"])
    shape, chunks = za["shape"], za["chunks"]
    if not shape:
        return np.frombuffer((root / name / "0").read_bytes(), dtype=dt).reshape(()), attrs
    out = np.zeros(shape, dtype=dt)
    grid = [int(np.ceil(s / c)) for s, c in zip(shape, chunks)]
    for idx in np.ndindex(*grid):
        f = root / name / ".".join(map(str, idx))
        if not f.exists():
            continue
        block = np.frombuffer(f.read_bytes(), dtype=dt).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out, attrs


def read_dataarray_zarr(store):
    from .labeled import DataArray
    from .ncio import _decode_time
    root = Path(store)
    arrays = [p.parent.name for p in root.glob("*/.zarray")]
    coords, payload = {}, None
    for n in arrays:
        vals, attrs = _read_array(root, n)
        dims = attrs.get("_ARRAY_DIMENSIONS", [])
        if len(dims) > 1:
            payload = (n, vals, dims)
            continue
        if "units" in attrs and " since " in attrs["units"]:
            vals = _decode_time(vals, attrs["units"])
        coords[n] = vals
    if payload is None:
        raise ValueError(f"no data array in {store}")
    name, vals, dims = payload
    return DataArray(vals, dims, coords, None if name == UNNAMED else name)
