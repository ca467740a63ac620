/* C ABI of the delivery helpers: what happens to a predicted state between the engine's output buffer and a file.
 *
 * The reference brings every state to the host with `.cpu().numpy()` (/root/reference/skyrim/core/models/utils.py:36) and writes
 * it with `pred.to_netcdf(output_path, engine="scipy")` (/root/reference/skyrim/common.py:144, called from
 * /root/reference/skyrim/core/models/base.py:134-143): netCDF-3 stores IEEE floats big-endian, so 573 MB per Pangu step go
 * through a byte swap on the host between the two.  skio_bswap32 does that swap in HBM; the swapped image is what the copy stream
 * brings to pinned host memory, and the save threads hand it to pwrite() untouched (skyrim_amd/deliver.py, ncio.py).
 * All pointers are device pointers; the call is asynchronous on `stream` (a hipStream_t); nothing is allocated inside. */
#ifndef SKYRIM_IO_H
#define SKYRIM_IO_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKIO_ABI_VERSION 1
#define SKIO_E_ARG (-1) /* bad argument: NULL pointer, misaligned pointer (both must be 4-byte aligned) */
#define SKIO_E_HIP (-2) /* the launch failed */

int skio_abi_version(void);

/* dst[i] = byte-reversed src[i] for i < n_words (32-bit words; src == dst is allowed).  HBM-bound: 8 bytes of traffic per word. */
int skio_bswap32(const void* src, void* dst, size_t n_words, void* stream);

#ifdef __cplusplus
}
#endif
#endif
