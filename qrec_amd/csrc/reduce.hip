// Full-table reductions used by the epoch-end loss terms (model/ranking/BPR.py:40:
// regU*(P*P).sum() + regI*(Q*Q).sum()).  Pure streaming read: HBM-bound, 4 B/element.
#include "common.h"

using namespace qrec;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T *__restrict__ x, int64_t rows, int d,
                                                    int ld, double *__restrict__ out) {
    // one wavefront per row slice keeps the loads coalesced even when ld > d
    double acc = 0.0;
    const int64_t total = rows * (int64_t)ld;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(k % ld);
        if (col < d) { const double v = (double)x[k]; acc += v * v; }
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
    int64_t blocks = (rows * ld + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == QREC_F32)
        hipLaunchKernelGGL(sumsq_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const float *)d_x, rows, d, ld, d_out);
    else
        hipLaunchKernelGGL(sumsq_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st,
                           (const double *)d_x, rows, d, ld, d_out);
    QREC_LAUNCH_CHECK();
    return QREC_OK;
}
