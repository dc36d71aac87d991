// Shared helpers for libqrec_hip.so (gfx950 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/qrec_hip.h"

namespace qrec {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define QREC_HIP_CHECK(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::qrec::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                              __FILE__, __LINE__);                                       \
            return QREC_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define QREC_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::qrec::set_error(__VA_ARGS__);     \
            return QREC_ERR_INVALID;            \
        }                                       \
    } while (0)

#define QREC_LAUNCH_CHECK() QREC_HIP_CHECK(hipGetLastError())

constexpr int kWave = 64;  // gfx950 wavefront

// ---- buffer resources: the only way to get 16-byte loads/stores with an explicit cache
// policy (sc1 = bypass the non-coherent per-XCD caches) and compiler-tracked waitcnts.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kAuxPlain = 0;
constexpr int kAuxSc1 = 16;  // gfx940+ cache-policy bit 4 = sc1

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    // raw buffer, no swizzle, bounds-checked against `bytes`
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <int AUX>
__device__ inline f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX);
    return __builtin_bit_cast(f32x4, v);
}
template <int AUX>
__device__ inline void buf_store4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, AUX);
}

// Sum across the lanes of one row of WIDTH (power of two <= 64) consecutive lanes; every
// lane of the row receives the total.  xor-butterfly: the compiler lowers the small
// strides to DPP and the rest to ds_bpermute/permlane.
template <int WIDTH, typename T>
__device__ inline T row_allreduce_sum(T v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

}  // namespace qrec
