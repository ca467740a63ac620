"""netCDF-3 persistence of a labelled array through scipy.io.netcdf_file.

The reference writes per-step files with ``pred.to_netcdf(output_path, engine="scipy")``
(/root/reference/skyrim/common.py:144), i.e. netCDF-3 classic in xarray's CF encoding; this module
produces the same on-disk convention without xarray: dimension variables for every coordinate,
datetimes as "hours since <t0>" (float64, proleptic_gregorian), strings as fixed-width char arrays
with a trailing ``string<N>`` dimension, and the payload variable named like xarray's unnamed
DataArray (``__xarray_dataarray_variable__``).  Files written here open with
``xarray.open_dataarray(path)``; files written by xarray's scipy engine read back here.
"""
from __future__ import annotations

import numpy as np

UNNAMED = "__xarray_dataarray_variable__"


def _encode_time(vals: np.ndarray):
    t0 = vals.reshape(-1)[0].astype("datetime64[s]")
    hours = (vals.astype("datetime64[s]") - t0).astype("timedelta64[s]").astype(np.float64) / 3600.0
    return hours, f"hours since {str(t0).replace('T', ' ')}"


def _decode_time(vals: np.ndarray, units: str) -> np.ndarray:
    unit, _, epoch = units.partition(" since ")
    scale = {"seconds": 1.0, "minutes": 60.0, "hours": 3600.0, "days": 86400.0}[unit.strip()]
    t0 = np.datetime64(epoch.strip().replace(" ", "T"), "s")
    return (t0 + (np.asarray(vals, dtype=np.float64) * scale).round().astype("timedelta64[s]")).astype("datetime64[ns]")


def _native(a: np.ndarray) -> np.ndarray:
    return a.astype(a.dtype.newbyteorder("=")) if a.dtype.byteorder not in ("=", "|") else a


FAST_PAYLOAD_BYTES = 64 << 20      # payloads of at least this size go through the parallel writer below
_SCRATCH = 1 << 20                 # elements converted per pwrite


def _parallel_payload_write(path, offset: int, payload: np.ndarray, threads: int | None = None, use_mmap: bool | None = None):
    """Big-endian float32 bytes of ``payload`` into ``path`` at ``offset``: the array is cut into pieces, every worker converts its pieces
    (numpy releases the GIL in the cast loop) and writes them with ``os.pwrite`` (released in the system call).  A 573 MB forecast step
    otherwise spends ~0.5 s in three serial passes (byte swap, ``tobytes``, ``write``) of scipy's writer.

    What bounds it (tools/predict_cost.py on the MI355X host, 573 MB into tmpfs): buffered writes to ONE file serialise on its inode, so the
    page cache takes the bytes at ~6 GB/s whatever the worker count (8 workers 89 ms, 16: 94, 64: 100) -- the conversion only has to keep up,
    hence few workers (4; SKYRIM_NC_THREADS); ``rollout`` gets its throughput from writing several steps' files at once instead
    (core/models/base.py SAVE_WORKERS; 8 full-size steps with their files, ms per step: 1 file at a time 151, 3: 76, 4: 61 - 70, 6: 50, 8: 47 - 52).
    ``use_mmap`` / ``SKYRIM_NC_MMAP=1``: the workers convert straight into a shared mapping of the file (no intermediate buffer, no system
    call per piece).  Twice as fast on a small host (8 cores: 110 vs 230 ms) but page-fault bound and slower the more threads fault on a
    large one (256 hardware threads: 8 workers 122 ms, 32: 233, 64: 406), so it is opt-in."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    flat = payload.reshape(-1)
    n = flat.shape[0]
    threads = threads or int(os.environ.get("SKYRIM_NC_THREADS", 0)) or max(1, min(4, (os.cpu_count() or 4) // 2))
    chunk = max(1 << 18, -(-n // (threads * 4)))                     # elements per piece: a few pieces per worker
    if use_mmap is None:
        use_mmap = os.environ.get("SKYRIM_NC_MMAP", "0") == "1"
    fd = os.open(str(path), os.O_RDWR)
    try:
        view = None
        if use_mmap and n:
            import mmap
            base = offset // mmap.ALLOCATIONGRANULARITY * mmap.ALLOCATIONGRANULARITY
            try:
                if os.fstat(fd).st_size < offset + 4 * n:              # the hole is the file's tail: a store past the end would be a SIGBUS
                    os.ftruncate(fd, offset + 4 * n)
                view = mmap.mmap(fd, offset - base + 4 * n, offset=base)
            except (OSError, ValueError):
                view = None                                            # not mappable (some network / FUSE targets): pwrite below
        if view is not None:
            dst = np.frombuffer(view, dtype=">f4", count=n, offset=offset - base)
            try:
                def put(i0):
                    dst[i0:i0 + chunk] = flat[i0:i0 + chunk]           # cast + byte swap in one pass, into the mapping

                with ThreadPoolExecutor(max_workers=threads) as pool:
                    list(pool.map(put, range(0, n, chunk)))
            finally:
                # the array exports the mapping's buffer: it has to go BEFORE close(), also when a store raised -- otherwise close() raises
                # BufferError on top of (and instead of) the store's own error
                del dst, put
                try:
                    view.close()
                except BufferError:                                    # a traceback still holds a frame that holds the array: leave the unmap to the GC
                    pass
            return

        def put(i0):
            # converted through a 4 MB scratch that stays in the core's cache between the cast and the system call (a fresh array per
            # piece is a fresh mapping: page faults on every piece, and the converted bytes go through DRAM twice)
            i1 = min(i0 + chunk, n)
            scratch = np.empty(min(_SCRATCH, i1 - i0), dtype=">f4")
            for j in range(i0, i1, _SCRATCH):
                k = min(_SCRATCH, i1 - j)
                np.copyto(scratch[:k], flat[j:j + k])
                mv, off = memoryview(scratch[:k]).cast("B"), offset + 4 * j
                while len(mv):                                        # pwrite may write less than asked
                    w = os.pwrite(fd, mv, off)
                    mv, off = mv[w:], off + w
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(put, range(0, n, chunk)))
    finally:
        os.close(fd)


def write_dataarray_netcdf3(da, path, fast_threshold: int | None = None):
    """``fast_threshold`` (bytes, default FAST_PAYLOAD_BYTES): float32 payloads at least this large are written by
    ``_parallel_payload_write`` into the hole scipy's writer leaves for them -- header, offsets and coordinate variables are scipy's own, the
    payload bytes are the same big-endian values: the file is byte-identical to the plain path (tests/test_ncio.py).

    The fast path leans on scipy internals (``_write_var_data``, ``_begin``, ``_pack_begin``, ``_vsize``): it writes to ``<path>.part`` and
    renames on success; if those internals are gone (AttributeError / a size that does not add up) the partial file is removed and scipy's
    plain writer takes over; an I/O error (ENOSPC ...) removes the partial file and is raised -- never a valid header over a garbage payload."""
    image = _image_of(da)
    payload0 = da.__dict__["_values"] if image is not None else da.values      # (with an image nobody waits for, or fills, the native array)
    thr = FAST_PAYLOAD_BYTES if fast_threshold is None else fast_threshold
    fast = payload0.dtype == np.float32 and payload0.flags.c_contiguous and payload0.nbytes >= thr
    if image is not None and not fast:
        image = None
    if fast:
        import os
        part = f"{os.fspath(path)}.part"
        try:
            _write_netcdf3(da, part, True, image)
            os.replace(part, os.fspath(path))
            return
        except (AttributeError, _FastPathMismatch):
            _unlink(part)                                  # scipy changed underneath: the plain writer below
        except BaseException:
            _unlink(part)
            raise
    _write_netcdf3(da, path, False)


class _FastPathMismatch(RuntimeError):
    pass


def _unlink(p):
    import os
    try:
        os.unlink(p)
    except OSError:
        pass


def _image_of(da):
    """The big-endian host image a delivered prediction carries (deliver.BigEndianImage), if it still mirrors the array ``da`` holds."""
    image, raw = da.__dict__.get("_image"), da.__dict__.get("_values")
    if image is None or raw is None or not image.mirrors(raw):
        return None
    return image


def _image_payload_write(path, offset: int, image):
    """The payload is already in the file's byte order in (pinned) host memory: one pwrite loop.  Writes to one file serialise on its
    inode whatever the thread count, so there is nothing to split; the throughput of a rollout comes from its files being written side
    by side (core/models/base.py SAVE_WORKERS)."""
    import os
    off = offset
    fd = os.open(str(path), os.O_RDWR)
    try:
        for mv in image.segments():                       # (one per part of the image: the state the step started from, the states it produced)
            while len(mv):
                w = os.pwrite(fd, mv[:64 << 20], off)
                mv, off = mv[w:], off + w
    finally:
        os.close(fd)


def _write_netcdf3(da, path, fast: bool, image=None):
    from scipy.io import netcdf_file
    payload0 = da.__dict__["_values"] if image is not None else da.values
    pname = da.name or UNNAMED
    hole = {}

    class _File(netcdf_file):
        def _write_var_data(self, name):                 # scipy: begin offset into the header, then the variable's bytes
            if not (fast and name == pname):
                return super()._write_var_data(name)
            var = self.variables[name]
            begin = self.fp.tell()
            self.fp.seek(var._begin)
            self._pack_begin(begin)
            self.fp.seek(begin + var._vsize)             # leave a hole of the variable's size; whatever scipy writes next starts behind it
            hole["begin"], hole["vsize"] = begin, var._vsize

    with (_File if fast else netcdf_file)(str(path), "w", version=2) as f:
        for d, n in zip(da.dims, da.shape):
            f.createDimension(d, int(n))
        for name, vals in da._coords.items():
            dims = (name,) if (name in da.dims and vals.ndim == 1) else ()
            if np.issubdtype(vals.dtype, np.datetime64):
                enc, units = _encode_time(vals)
                v = f.createVariable(name, "d", dims)
                v[...] = enc if dims else enc.reshape(())
                v.units = units
                v.calendar = "proleptic_gregorian"
            elif vals.dtype.kind in "US":
                b = np.char.encode(vals.astype(str), "utf-8") if vals.dtype.kind == "U" else vals
                width = max(1, b.dtype.itemsize)
                sdim = f"string{width}"
                if sdim not in f.dimensions:
                    f.createDimension(sdim, width)
                v = f.createVariable(name, "c", dims + (sdim,))
                v[...] = b.astype(f"S{width}").reshape(b.shape + (1,)).view("S1").reshape(b.shape + (width,))
            else:
                arr = vals.astype(np.float64) if vals.dtype.kind == "f" else vals.astype(np.int32)
                v = f.createVariable(name, arr.dtype.char, dims)
                v[...] = arr
        payload = payload0
        if payload.dtype == np.float16 or payload.dtype.kind not in "fi":
            payload = payload.astype(np.float32)
        v = f.createVariable(pname, payload.dtype.char, da.dims)
        if not fast:
            v[...] = payload                              # (fast: scipy's own buffer for it stays untouched -- never paged in)
        coord_names = " ".join(k for k in da._coords if k not in da.dims)
        if coord_names:
            v.coordinates = coord_names
    if fast:
        if hole.get("vsize") != payload0.nbytes:
            raise _FastPathMismatch(f"payload hole {hole} does not match {payload0.nbytes} bytes")
        if image is not None:
            _image_payload_write(path, hole["begin"], image)
        else:
            _parallel_payload_write(path, hole["begin"], payload0)


def read_dataarray_netcdf3(path):
    from scipy.io import netcdf_file
    from .labeled import DataArray
    with netcdf_file(str(path), "r", mmap=False) as f:
        names = list(f.variables)
        dimnames = set(f.dimensions)
        data_vars = [n for n in names if n not in dimnames and len(f.variables[n].dimensions) > 1
                     and not (f.variables[n].typecode() == "c" and len(f.variables[n].dimensions) == 2)]
        if len(data_vars) != 1:
            raise ValueError(f"expected exactly one data variable in {path}, found {data_vars}")
        var = f.variables[data_vars[0]]
        coords = {}
        for n in names:
            if n == data_vars[0]:
                continue
            v = f.variables[n]
            vals = _native(np.array(v[...]))
            if v.typecode() == "c":
                vals = np.array([b"".join(row).decode("utf-8").rstrip("\x00 ") for row in vals.reshape(-1, vals.shape[-1])]).reshape(vals.shape[:-1])
            elif hasattr(v, "units") and " since " in (v.units.decode() if isinstance(v.units, bytes) else v.units):
                vals = _decode_time(vals, v.units.decode() if isinstance(v.units, bytes) else v.units)
            coords[n] = vals
        name = None if data_vars[0] == UNNAMED else data_vars[0]
        return DataArray(_native(np.array(var[...])), var.dimensions, coords, name)
