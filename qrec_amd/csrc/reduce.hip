// Full-table reductions used by the epoch-end loss terms (model/ranking/BPR.py:40:
// regU*(P*P).sum() + regI*(Q*Q).sum()).  Pure streaming read: 4 B/element, 16 B per lane per
// load.  The pad columns [d, ld) of a table are zero by contract (include/qrec_hip.h), so
// the whole [rows x ld] block is summed without any column test.
#include "common.h"

using namespace qrec;

namespace {

template <typename T, typename V4>
__global__ __launch_bounds__(256) void sumsq_kernel(const T *__restrict__ x, int64_t n_elems,
                                                    double *__restrict__ out) {
    double acc = 0.0;
    const int64_t n4 = n_elems >> 2;
    const V4 *x4 = reinterpret_cast<const V4 *>(x);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4;
         k += (int64_t)gridDim.x * blockDim.x) {
        const V4 v = x4[k];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_elems & 3)) {  // tail (ld is a multiple of 4 for fp32)
        const double v = (double)x[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
    __shared__ double s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

}  // namespace

extern "C" int qrec_sumsq(const void *d_x, int dtype, int64_t rows, int32_t d, int32_t ld,
                          double *d_out, void *stream) {
    QREC_REQUIRE(d_out && rows >= 0 && d >= 1 && ld >= d, "qrec_sumsq: bad arguments");
    QREC_REQUIRE(dtype == QREC_F32 || dtype == QREC_F64, "qrec_sumsq: bad dtype %d", dtype);
    hipStream_t st = as_stream(stream);
    QREC_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(double), st));
    if (rows == 0) return QREC_OK;
    QREC_REQUIRE(d_x, "qrec_sumsq: null table");
    const int64_t n = rows * (int64_t)ld;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (dtype == QREC_F32)
        hipLaunchKernelGGL((sumsq_kernel<float, float4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float *)d_x, n, d_out);
    else
        hipLaunchKernelGGL((sumsq_kernel<double, double4>), dim3((unsigned)blocks), dim3(256), 0, st,
                           (const double *)d_x, n, d_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
