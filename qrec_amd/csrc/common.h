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

// ---- ordered scatter-add (ordered.hip): the deterministic counterpart of the float-atomic gradient scatters.  A producer
// writes slot s's row to contrib[s][ld] and its destination row to keys[s] (< 0: none); ordered_scatter_run sorts (key, slot)
// stably and adds each row's slots in ascending slot order, class by class (class = slot / class_size; 0: one class).
struct OrderedScatterWs {
    float *contrib; int32_t *keys, *keys_sorted, *slots_sorted; void *temp; size_t temp_bytes;
};
int ordered_ws_bytes(int64_t n_slots, int ld, int64_t *bytes);
int ordered_ws_carve(void *ws, int64_t ws_bytes, int64_t n_slots, int ld, OrderedScatterWs *w);
int ordered_scatter_run(const OrderedScatterWs &w, int64_t n_slots, int ld, int64_t class_size, float *out, hipStream_t st);

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

// Sum over the 64 lanes of a wavefront, every lane receives it.  DPP adds inside the 16-lane rows (row_shr 1,2,3 of the
// input, then row_shr 4 and 8 of the running sum), row_bcast 15 / 31 across the rows, total read from lane 63: six
// VALU-rate steps.  The xor butterfly this replaces goes through ds_bpermute -- the LDS crossbar, >100 clocks per step and
// two of them per fp64 value: a third of the order-exact kernels' per-triplet chain (measured).
// A fixed summation tree, so the result is deterministic; it is NOT the butterfly's tree (last-bit differences).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ inline float dpp_take(float v) {      // lanes without a source (bounds, masks) receive 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ inline double dpp_take(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, BANK_MASK, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, BANK_MASK, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ inline float read_lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ inline double read_lane63(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ inline T wave_sum_dpp(T v) {
    T t = v + dpp_take<0x111, 0xf, 0xf>(v);          // row_shr:1
    t = t + dpp_take<0x112, 0xf, 0xf>(v);            // row_shr:2
    t = t + dpp_take<0x113, 0xf, 0xf>(v);            // row_shr:3   -> lanes 3, 7, 11, 15 of a row hold their quad's sum
    t = t + dpp_take<0x114, 0xf, 0xe>(t);            // row_shr:4, banks 1-3
    t = t + dpp_take<0x118, 0xf, 0xc>(t);            // row_shr:8, banks 2-3 -> lane 15 of a row holds the row's sum
    t = t + dpp_take<0x142, 0xa, 0xf>(t);            // row_bcast:15 into rows 1 and 3
    t = t + dpp_take<0x143, 0xc, 0xf>(t);            // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return read_lane63(t);
}

}  // namespace qrec
