"""Delivery of predicted states in the byte order of their files (include/skyrim_io.h).

The reference writes every step with ``pred.to_netcdf(output_path, engine="scipy")`` (/root/reference/skyrim/common.py:144): netCDF-3
holds big-endian floats, so the host swaps the bytes of 573 MB per Pangu step on the way to the file.  Here the swap runs in HBM
(``skio_bswap32``, ~0.1 ms per state) on the copy stream, the swapped image is what travels to pinned host memory, and the save threads
hand those bytes to ``pwrite`` as they are (ncio.py).  ``rollout`` asks for it per step (``run_basic_inference(..., deliver=)``):

* ``"be"``   -- an intermediate step of a saving rollout: only the big-endian image is copied; the native array the DataArray carries is
               filled from the image on the host the first time somebody reads ``values`` (nobody does on the default path: the next
               step takes the state from HBM, the writer takes the image);
* ``"both"`` -- the last step: the caller gets native numbers, the writer the image;
* ``None``   -- no image.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libskyrim_io.so"
ABI_VERSION = 1             # include/skyrim_io.h SKIO_ABI_VERSION
EXPORTS = ["skio_abi_version", "skio_bswap32"]
_lib = None


def load_library() -> ctypes.CDLL:
    """Load libskyrim_io.so (built in-tree by ``__graft_entry__.build()`` / ``make -C skyrim_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SKYRIM_IO_LIB", str(_LIB_PATH))
    if not os.path.exists(path):
        raise RuntimeError(f"HIP delivery library not found at {path}; build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                           " -- the big-endian delivery of a saving rollout has no host fallback (SKYRIM_SAVE_BE=0 turns it off)")
    lib = ctypes.CDLL(path)
    lib.skio_abi_version.restype = ctypes.c_int
    lib.skio_bswap32.restype = ctypes.c_int
    lib.skio_bswap32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    got = lib.skio_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path} is ABI v{got}, this host code binds v{ABI_VERSION}: rebuild (`make -C skyrim_amd/csrc`)")
    _lib = lib
    return lib


def bswap32(src, dst, stream) -> None:
    """``dst`` = the 32-bit words of ``src`` byte-reversed, queued on ``stream`` (a torch.cuda.Stream).  Both: contiguous 4-byte-element
    CUDA tensors of one size."""
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("bswap32 wants contiguous device tensors")
    if src.element_size() != 4 or dst.element_size() != 4 or src.numel() != dst.numel():
        raise ValueError(f"bswap32: {tuple(src.shape)} {src.dtype} -> {tuple(dst.shape)} {dst.dtype}")
    rc = load_library().skio_bswap32(src.data_ptr(), dst.data_ptr(), src.numel(), stream.cuda_stream)
    if rc != 0:
        raise RuntimeError(f"skio_bswap32 failed ({rc})")


class ImagePart:
    """Consecutive time entries of a prediction in file byte order: a C-contiguous ``>f4`` array in (pinned) host memory, the wait for its
    device-to-host copy, and whatever owns the memory."""
    __slots__ = ("array", "_wait", "keep")

    def __init__(self, array: np.ndarray, wait=None, keep=None):
        if array.dtype != np.dtype(">f4") or not array.flags.c_contiguous:
            raise ValueError(f"an image part is a C-contiguous big-endian float32 array, not {array.dtype} (contiguous: {array.flags.c_contiguous})")
        self.array, self._wait, self.keep = array, wait, keep

    def wait(self) -> None:
        w = self._wait
        if w is not None:
            w()
            self._wait = None

    def tail(self, rows: int) -> "ImagePart":
        return ImagePart(self.array[self.array.shape[0] - rows:], self._wait, self.keep)


class BigEndianImage:
    """The payload of a delivered DataArray as the file wants it: ``>f4`` parts that follow each other along the first (time) axis, in
    pinned host memory whose device-to-host copies may still be in flight; ``wait()`` returns once they have landed.  ``of`` remembers
    which native array the image mirrors: the writer uses the image only while the DataArray still carries that very array.
    ``head``: parts in front of ``array`` -- in a rollout the state a step starts from IS the previous step's prediction, whose image is
    already on the host: the new image borrows those bytes (``tail``) instead of bringing them over again."""

    def __init__(self, array: np.ndarray, of: np.ndarray, wait=None, keep=None, head=()):
        self.parts = list(head) + [ImagePart(array, wait, keep)]
        rows = sum(p.array.shape[0] for p in self.parts)
        if any(p.array.shape[1:] != of.shape[1:] for p in self.parts) or rows != of.shape[0]:
            raise ValueError(f"big-endian image of {[p.array.shape for p in self.parts]} for a {of.dtype} {of.shape} array")
        self._of = of

    @property
    def array(self) -> np.ndarray:
        """The whole image as one array (a copy when it has several parts: tests and diagnostics, not the writer)."""
        return self.parts[0].array if len(self.parts) == 1 else np.concatenate([p.array for p in self.parts])

    def wait(self) -> None:
        for p in self.parts:
            p.wait()

    def segments(self) -> list:
        """The bytes of the payload in file order (after ``wait``)."""
        self.wait()
        return [memoryview(p.array.reshape(-1).view(np.uint8)) for p in self.parts]

    def mirrors(self, values: np.ndarray) -> bool:
        return values is self._of

    def tail(self, rows: int = 1) -> "ImagePart | None":
        """The last ``rows`` time entries as a part another image can start with (None when they straddle two parts)."""
        last = self.parts[-1]
        return last.tail(rows) if last.array.shape[0] >= rows else None

    def fill_native(self, out: np.ndarray) -> None:
        """``out`` (native float32, same shape) = the image's numbers: the host-side swap the default path never needs."""
        self.wait()
        r = 0
        for p in self.parts:
            n = p.array.shape[0]
            np.copyto(out[r:r + n], p.array)
            r += n


def enabled() -> bool:
    return os.environ.get("SKYRIM_SAVE_BE", "1").lower() not in ("0", "off", "no", "false")
