"""ncio: the parallel payload writer produces the file scipy's plain writer produces, byte for byte (the per-step files of a rollout are
written by it on the save thread: core/models/base.py)."""
import datetime
import filecmp

import numpy as np
import pytest

from skyrim_amd import ncio
from skyrim_amd.labeled import DataArray, open_dataarray


def _da(shape=(2, 5, 37, 64), seed=0):
    rng = np.random.default_rng(seed)
    t0 = datetime.datetime(2024, 5, 13, 18)
    return DataArray(rng.standard_normal(shape).astype(np.float32), ["time", "channel", "lat", "lon"],
                     dict(time=[t0, t0 + datetime.timedelta(hours=6)], channel=[f"c{i}" for i in range(shape[1])],
                          lat=np.linspace(90, -90, shape[2]), lon=np.arange(shape[3]) * (360.0 / shape[3])))


def test_parallel_writer_is_byte_identical_to_the_plain_path(tmp_path):
    da = _da()
    ncio.write_dataarray_netcdf3(da, tmp_path / "plain.nc", fast_threshold=1 << 60)
    ncio.write_dataarray_netcdf3(da, tmp_path / "fast.nc", fast_threshold=0)
    assert filecmp.cmp(tmp_path / "plain.nc", tmp_path / "fast.nc", shallow=False)
    back = open_dataarray(str(tmp_path / "fast.nc"))
    assert np.array_equal(back.values, da.values) and back.dims == da.dims
    # a read-only payload (what run_basic_inference delivers) and odd sizes (pieces that do not divide the array)
    ro = _da((2, 3, 33, 50), seed=1)
    ro.values.flags.writeable = False
    ncio.write_dataarray_netcdf3(ro, tmp_path / "ro_fast.nc", fast_threshold=0)
    ncio.write_dataarray_netcdf3(ro, tmp_path / "ro_plain.nc", fast_threshold=1 << 60)
    assert filecmp.cmp(tmp_path / "ro_plain.nc", tmp_path / "ro_fast.nc", shallow=False)
    # many small pieces over few workers
    big = _da((2, 9, 181, 360), seed=2)
    ncio.write_dataarray_netcdf3(big, tmp_path / "big_plain.nc", fast_threshold=1 << 60)
    for mapped in (True, False):                                   # the mapped writer and its pwrite fallback: same bytes
        orig_w = ncio._parallel_payload_write
        try:
            ncio._parallel_payload_write = lambda path, off, payload, m=mapped: orig_w(path, off, payload, threads=2, use_mmap=m)
            ncio.write_dataarray_netcdf3(big, tmp_path / f"big_{mapped}.nc", fast_threshold=0)
        finally:
            ncio._parallel_payload_write = orig_w
        assert filecmp.cmp(tmp_path / "big_plain.nc", tmp_path / f"big_{mapped}.nc", shallow=False)
    orig = ncio._parallel_payload_write
    try:
        ncio._parallel_payload_write = lambda path, off, payload: orig(path, off, payload, threads=3)
        ncio.write_dataarray_netcdf3(big, tmp_path / "big_fast.nc", fast_threshold=0)
    finally:
        ncio._parallel_payload_write = orig
    assert filecmp.cmp(tmp_path / "big_plain.nc", tmp_path / "big_fast.nc", shallow=False)


def test_non_float32_and_small_payloads_take_the_plain_path(tmp_path):
    da = _da()
    da64 = DataArray(da.values.astype(np.float64), da.dims, dict(da._coords))
    ncio.write_dataarray_netcdf3(da64, tmp_path / "f64.nc", fast_threshold=0)
    assert np.array_equal(open_dataarray(str(tmp_path / "f64.nc")).values, da64.values)
    ncio.write_dataarray_netcdf3(da, tmp_path / "small.nc")                      # below FAST_PAYLOAD_BYTES
    assert np.array_equal(open_dataarray(str(tmp_path / "small.nc")).values, da.values)


def test_fast_path_failures_never_leave_a_half_written_file(tmp_path, monkeypatch):
    """The fast payload path writes to <path>.part and renames: if scipy's internals are gone it falls back to the plain writer (same bytes);
    an I/O error removes the partial file and propagates."""
    from skyrim_amd import ncio
    from skyrim_amd.labeled import DataArray
    vals = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(2, 3, 4, 5)
    da = DataArray(vals, dims=["time", "channel", "lat", "lon"],
                   coords=dict(time=np.array(["2024-01-01T00", "2024-01-01T06"], dtype="datetime64[ns]"), channel=["a", "b", "c"],
                               lat=np.linspace(90, -90, 4), lon=np.arange(5.0)))
    plain, fast, fallback = tmp_path / "plain.nc", tmp_path / "fast.nc", tmp_path / "fallback.nc"
    ncio.write_dataarray_netcdf3(da, plain, fast_threshold=1 << 40)
    ncio.write_dataarray_netcdf3(da, fast, fast_threshold=0)
    assert plain.read_bytes() == fast.read_bytes() and not (tmp_path / "fast.nc.part").exists()

    def gone(*a, **k):
        raise AttributeError("scipy changed")
    monkeypatch.setattr(ncio, "_parallel_payload_write", gone)
    ncio.write_dataarray_netcdf3(da, fallback, fast_threshold=0)
    assert fallback.read_bytes() == plain.read_bytes() and not (tmp_path / "fallback.nc.part").exists()

    def full(*a, **k):
        raise OSError(28, "No space left on device")
    monkeypatch.setattr(ncio, "_parallel_payload_write", full)
    with pytest.raises(OSError):
        ncio.write_dataarray_netcdf3(da, tmp_path / "enospc.nc", fast_threshold=0)
    assert not (tmp_path / "enospc.nc").exists() and not (tmp_path / "enospc.nc.part").exists()


def _with_image(da):
    """``da`` as a saving rollout delivers an intermediate step: the native array allocated but NOT filled, the numbers present only as
    the big-endian image (skyrim_amd/deliver.py); ``values`` fills the native array from the image on first read."""
    from skyrim_amd.deliver import BigEndianImage
    truth = da.values.copy()
    be = truth.astype(">f4")
    native = np.full_like(truth, np.nan)
    waited = []
    image = BigEndianImage(be, of=native, wait=lambda: waited.append(1))
    alias = native[...]                                             # the writable view the fill goes through
    out = DataArray(native, da.dims, dict(da._coords), ready=lambda: image.fill_native(alias), image=image)
    native.flags.writeable = False                                  # as ResidentState leaves a delivered array
    return out, truth, waited


def test_a_delivered_big_endian_image_is_written_as_it_is(tmp_path):
    """The file of a prediction that carries its big-endian image: byte-identical to the plain writer's, written without reading (hence
    without filling) the native array; the image is dropped when the array is replaced, and derived arrays do not inherit it."""
    da = _da((2, 9, 181, 360), seed=3)
    ncio.write_dataarray_netcdf3(da, tmp_path / "plain.nc", fast_threshold=1 << 60)
    img, truth, waited = _with_image(da)
    ncio.write_dataarray_netcdf3(img, tmp_path / "image.nc", fast_threshold=0)
    assert filecmp.cmp(tmp_path / "plain.nc", tmp_path / "image.nc", shallow=False)
    assert waited == [1] and img.__dict__["_ready"] is not None and np.isnan(img.__dict__["_values"]).all()      # nobody read ``values``
    # a reader gets the numbers, filled from the image once
    assert np.array_equal(img.values, truth) and img.__dict__["_ready"] is None
    # below the fast-path threshold the plain writer runs: it reads ``values`` (filled from the image)
    small, truth_s, _ = _with_image(_da())
    ncio.write_dataarray_netcdf3(small, tmp_path / "small.nc")
    assert np.array_equal(open_dataarray(str(tmp_path / "small.nc")).values, truth_s)
    # derived arrays and assigned values never use a stale image
    img2, truth2, _ = _with_image(da)
    sel = img2.isel(channel=[0, 2])
    assert sel.__dict__.get("_image") is None and np.array_equal(sel.values, truth2[:, [0, 2]])
    img3, truth3, _ = _with_image(da)
    img3.values = truth3 * 2
    assert img3.__dict__["_image"] is None
    ncio.write_dataarray_netcdf3(img3, tmp_path / "doubled.nc", fast_threshold=0)
    assert np.array_equal(open_dataarray(str(tmp_path / "doubled.nc")).values, truth3 * 2)


def test_an_image_of_another_array_is_refused():
    from skyrim_amd.deliver import BigEndianImage
    a = np.zeros((2, 3), np.float32)
    with pytest.raises(ValueError):
        BigEndianImage(a.astype(">f4")[:1], of=a)
    with pytest.raises(ValueError):
        BigEndianImage(a, of=a)                                     # native byte order is not an image
    image = BigEndianImage(a.astype(">f4"), of=a)
    assert image.mirrors(a) and not image.mirrors(a.copy())
