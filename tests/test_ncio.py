"""ncio: the parallel payload writer produces the file scipy's plain writer produces, byte for byte (the per-step files of a rollout are
written by it on the save thread: core/models/base.py)."""
import datetime
import filecmp

import numpy as np

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
