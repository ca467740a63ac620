"""Minimal labelled N-d array with the slice of the ``xarray.DataArray`` surface the Skyrim API uses.

The reference returns ``xr.DataArray(dims=["time","channel","lat","lon"])`` everywhere
(/root/reference/skyrim/core/models/utils.py:42-49).  xarray is not installed in this image (nor on the
GPU box), so this module provides the accessors the reference's own code paths touch -- ``dims``,
``coords``, ``values``, ``shape``, ``sel`` (exact / list / slice / ``method="nearest"``), ``isel``,
``squeeze``, ``item``, ``assign_coords``, ``mean``, coordinate attributes (``da.channel``, ``da.lat`` ...)
-- and ``concat`` / ``open_dataarray`` / ``to_netcdf``.  When xarray is importable, ``as_xarray`` converts.
"""
from __future__ import annotations

import datetime as _dt
from pathlib import Path
from typing import Any, Iterable

import numpy as np


def _as_index_array(v) -> np.ndarray:
    if isinstance(v, np.ndarray):
        return v
    v = list(v)
    if v and isinstance(v[0], (_dt.datetime, np.datetime64)):
        return np.array(v, dtype="datetime64[ns]")
    return np.array(v)


class Coordinate:
    """1-d coordinate (what ``da.lat`` / ``da.coords["lat"]`` return)."""

    def __init__(self, name: str, values: np.ndarray):
        self.name = name
        self.values = values

    def __len__(self): return len(self.values)
    def __iter__(self): return iter(self.values)
    def __getitem__(self, i): return Coordinate(self.name, np.atleast_1d(self.values[i])) if not np.isscalar(self.values[i]) else self.values[i]
    def __contains__(self, v): return v in self.values
    def __eq__(self, other): return self.values == (other.values if isinstance(other, Coordinate) else other)
    def __repr__(self): return f"<Coordinate {self.name} {self.values!r}>"
    def item(self): return self.values.reshape(-1)[0].item() if self.values.size == 1 else self.values.item()
    def tolist(self): return self.values.tolist()
    @property
    def size(self): return self.values.size


class DataArray:
    def __init__(self, data, dims: Iterable[str], coords: dict[str, Any] | None = None, name: str | None = None, ready=None, image=None):
        """``ready``: a callable that returns once ``data`` holds its final contents (run_basic_inference hands over a pinned host
        buffer whose device-to-host copy may still be in flight, with the copy's event wait as ``ready``).  It is called the first
        time ``values`` is read -- shape, dims and coordinates never wait -- so a prediction fed straight back into the next step
        (``rollout``) lets its copy overlap that step; whoever reads the numbers (the save thread, the caller) waits first.
        ``image``: the same numbers as a big-endian host image (deliver.BigEndianImage) for the netCDF writer, which then never reads
        ``values``; it belongs to this object and this array only -- assigning ``values`` drops it, derived arrays do not inherit it."""
        self._values = np.asarray(data)
        self._ready = ready
        self._image = image
        self.dims = tuple(dims)
        if len(self.dims) != self._values.ndim:
            raise ValueError(f"{len(self.dims)} dims for a {self._values.ndim}-d array")
        self.name = name
        self._coords: dict[str, np.ndarray] = {}
        for k, v in (coords or {}).items():
            arr = _as_index_array(v) if not np.isscalar(v) else np.array(v)
            if k in self.dims and arr.shape != (self._values.shape[self.dims.index(k)],):
                raise ValueError(f"coordinate {k} has length {arr.shape}, dim has {self._values.shape[self.dims.index(k)]}")
            self._coords[k] = arr

    @property
    def values(self) -> np.ndarray:
        ready = self.__dict__.get("_ready")
        if ready is not None:
            ready()                                 # (an event wait: harmless if two threads get here together)
            self._ready = None
        return self._values

    @values.setter
    def values(self, v):
        self._values, self._ready, self._image = np.asarray(v), None, None

    # -- basic accessors --------------------------------------------------- #
    @property
    def coords(self) -> dict[str, Coordinate]:
        return {k: Coordinate(k, v) for k, v in self._coords.items()}
    @property
    def shape(self): return self._values.shape
    @property
    def size(self): return self._values.size
    @property
    def dtype(self): return self._values.dtype
    @property
    def ndim(self): return self._values.ndim

    def __getattr__(self, name):
        c = self.__dict__.get("_coords", {})
        if name in c:
            return Coordinate(name, c[name])
        raise AttributeError(name)

    def __repr__(self):
        return f"<DataArray {dict(zip(self.dims, self.shape))} coords={list(self._coords)}>"

    def item(self): return self.values.item()
    def copy(self): return DataArray(self.values.copy(), self.dims, dict(self._coords), self.name)
    def __array__(self, dtype=None): return self.values if dtype is None else self.values.astype(dtype)

    # -- indexing ----------------------------------------------------------- #
    def _index(self, dim: str, indexer, drop_scalar=True) -> "DataArray":
        ax = self.dims.index(dim)
        scalar = np.isscalar(indexer) or (isinstance(indexer, np.ndarray) and indexer.ndim == 0)
        data = np.take(self.values, indexer, axis=ax) if not isinstance(indexer, slice) else self.values[(slice(None),) * ax + (indexer,)]
        coords = dict(self._coords)
        if dim in coords:
            coords[dim] = coords[dim][indexer]
        dims = self.dims
        if scalar and drop_scalar:
            dims = tuple(d for d in dims if d != dim)
        return DataArray(data, dims, coords, self.name)

    def isel(self, **kw) -> "DataArray":
        out = self
        for dim, i in kw.items():
            if dim not in out.dims:
                raise KeyError(dim)
            if isinstance(i, (list, tuple)):
                i = np.asarray(i)
            out = out._index(dim, i)
        return out

    def sel(self, method: str | None = None, **kw) -> "DataArray":
        out = self
        for dim, key in kw.items():
            if dim not in out.dims:
                raise KeyError(dim)
            c = out._coords[dim]
            if isinstance(key, slice):
                lo = key.start if key.start is not None else -np.inf
                hi = key.stop if key.stop is not None else np.inf
                lo, hi = min(lo, hi), max(lo, hi)
                out = out._index(dim, np.nonzero((c >= lo) & (c <= hi))[0])
            elif isinstance(key, (list, tuple, np.ndarray, Coordinate)):
                keys = key.values if isinstance(key, Coordinate) else key
                pos = {v: i for i, v in enumerate(c.tolist())}
                try:
                    out = out._index(dim, np.array([pos[k.item() if hasattr(k, "item") else k] for k in keys], dtype=np.int64))
                except KeyError as e:
                    raise KeyError(f"{e.args[0]!r} not found in coordinate {dim}") from None
            else:
                if isinstance(key, _dt.datetime):
                    key = np.datetime64(key, "ns")
                if method == "nearest":
                    i = int(np.abs(c - key).argmin())
                else:
                    hit = np.nonzero(c == key)[0]
                    if hit.size == 0:
                        raise KeyError(f"{key!r} not found in coordinate {dim}")
                    i = int(hit[0])
                out = out._index(dim, i)
        return out

    def squeeze(self) -> "DataArray":
        keep = [i for i, n in enumerate(self.shape) if n != 1]
        coords = {}
        for k, v in self._coords.items():
            coords[k] = v.reshape(()) if (k in self.dims and self.shape[self.dims.index(k)] == 1) else v
        return DataArray(self.values.reshape([self.shape[i] for i in keep]), [self.dims[i] for i in keep], coords, self.name)

    def assign_coords(self, **kw) -> "DataArray":
        c = dict(self._coords)
        c.update(kw)
        return DataArray(self.values, self.dims, c, self.name)

    def mean(self, dim: str) -> "DataArray":
        ax = self.dims.index(dim)
        coords = {k: v for k, v in self._coords.items() if k != dim}
        return DataArray(self.values.mean(axis=ax), [d for d in self.dims if d != dim], coords, self.name)

    def expand_dims(self, dim: str) -> "DataArray":
        """New leading dim of size 1; a scalar coordinate of that name becomes its 1-element index."""
        coords = dict(self._coords)
        if dim in coords:
            coords[dim] = np.atleast_1d(coords[dim])
        return DataArray(self.values[None], (dim,) + self.dims, coords, self.name)

    def rename(self, names: dict) -> "DataArray":
        return DataArray(self.values, [names.get(d, d) for d in self.dims], {names.get(k, k): v for k, v in self._coords.items()}, self.name)

    def transpose(self, *dims) -> "DataArray":
        return DataArray(self.values.transpose([self.dims.index(d) for d in dims]), dims, dict(self._coords), self.name)

    # -- persistence --------------------------------------------------------- #
    def to_netcdf(self, path, engine: str = "scipy"):
        from .ncio import write_dataarray_netcdf3
        write_dataarray_netcdf3(self, path)

    def to_zarr(self, store, mode: str = "w", append_dim: str | None = None, consolidated: bool = True):
        from .zarrio import write_dataarray_zarr
        write_dataarray_zarr(self, store, mode=mode, append_dim=append_dim, consolidated=consolidated)


class Dataset:
    """Named DataArrays over shared dims -- the part of ``xarray.Dataset`` GraphCast's stepper state is read through
    (/root/reference/skyrim/core/models/graphcast.py:68-91: ``ds.squeeze(dim=...)``, ``ds[name]``, ``.isel``, ``.expand_dims``)."""

    def __init__(self, data_vars: dict[str, DataArray]):
        self.data_vars = dict(data_vars)

    def __getitem__(self, name: str) -> DataArray:
        return self.data_vars[name]

    def __contains__(self, name) -> bool:
        return name in self.data_vars

    def keys(self):
        return self.data_vars.keys()

    def _map(self, fn) -> "Dataset":
        return Dataset({k: fn(v) for k, v in self.data_vars.items()})

    def squeeze(self, dim: str) -> "Dataset":
        return self._map(lambda v: v.isel(**{dim: 0}) if dim in v.dims else v)

    def isel(self, **kw) -> "Dataset":
        return self._map(lambda v: v.isel(**{d: i for d, i in kw.items() if d in v.dims}))

    def expand_dims(self, dim: str) -> "Dataset":
        return self._map(lambda v: v.expand_dims(dim))

    def __repr__(self):
        return f"<Dataset {list(self.data_vars)}>"


def concat(arrays: list[DataArray], dim: str) -> DataArray:
    """Concatenate along an existing dim, or stack along a new one (``xr.concat`` semantics used by the
    reference: dim="time" in graphcast.py, dim="model" in ensemble.py:64)."""
    first = arrays[0]
    if dim in first.dims:
        ax = first.dims.index(dim)
        coords = dict(first._coords)
        coords[dim] = np.concatenate([a._coords[dim] for a in arrays])
        return DataArray(np.concatenate([a.values for a in arrays], axis=ax), first.dims, coords, first.name)
    coords = dict(first._coords)
    return DataArray(np.stack([a.values for a in arrays], axis=0), (dim,) + first.dims, coords, first.name)


def open_dataarray(path) -> DataArray:
    p = Path(path)
    if p.is_dir():
        from .zarrio import read_dataarray_zarr
        return read_dataarray_zarr(p)
    from .ncio import read_dataarray_netcdf3
    return read_dataarray_netcdf3(p)


def as_xarray(da: DataArray):
    """Convert to a real ``xarray.DataArray`` when xarray is installed."""
    import xarray as xr  # noqa: WPS433
    return xr.DataArray(da.values, dims=da.dims, coords={k: (k, v) if v.ndim == 1 else v for k, v in da._coords.items()}, name=da.name)
