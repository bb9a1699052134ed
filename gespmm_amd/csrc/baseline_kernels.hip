// baseline_kernels.hip — the comparison column the reference's Gunrock app provides.
//
// gunrock-test/app/spmm/spmm_enactor.cuh:92-105 is an edge map ("advance" over ALL_EDGES of the
// CSR): for every edge (src -> dest) and every feature j,
//     atomicAdd(output + dest * feature_len + j, input[src * feature_len + j]);
// i.e. the SCATTER form out = A^T * in on the pattern (A == 1), one atomic per edge and feature,
// output zeroed beforehand. It is NOT a product path of this library: the row-product kernels
// need no atomics and keep a fixed summation order; this one exists so the driver can print the
// same "atomic baseline" column (spmm_test --atomic-baseline) and as a second, independent
// device-side checker (the order of the additions is whatever the hardware serialises, so it is
// tolerance-checked).
//
// Layout on wave64: a lane group of W = min(64, pow2 >= N/V) lanes takes one edge at a time and
// covers the feature row with dwordx{V} loads; edges are dealt to groups grid-stride. The source
// row of an edge position comes from a binary search in rowptr (once per edge per group).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_kernels.h"

namespace gespmm {

namespace {

__global__ __launch_bounds__(kThreads) void atomic_scatter_kernel(const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ colind,
                                                                   const float* __restrict__ in,
                                                                   float* __restrict__ out, int M, int64_t nnz, int N,
                                                                   int W) {
    const int lane = threadIdx.x & 63;
    const int groups_per_wave = 64 / W;
    const int g = lane / W;
    const int l = lane % W;
    const int64_t gid = ((int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6)) * groups_per_wave + g;
    const int64_t ngroups = (int64_t)gridDim.x * kWaves * groups_per_wave;
    for (int64_t e = gid; e < nnz; e += ngroups) {
        int lo = 0, hi = M;  // rowptr[lo] <= e < rowptr[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int64_t)rowptr[mid] <= e) lo = mid;
            else hi = mid;
        }
        const float* src = in + (size_t)lo * (size_t)N;
        float* dst = out + (size_t)colind[e] * (size_t)N;
        for (int j = l; j < N; j += W) atomicAdd(dst + j, src[j]);
    }
}

}  // namespace

hipError_t launch_atomic_scatter(const int32_t* rowptr, const int32_t* colind, const float* in, float* out, int64_t M,
                                 int64_t K, int64_t N, int64_t nnz, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)K * (size_t)N * sizeof(float), st);
    if (e != hipSuccess || nnz == 0 || N == 0) return e;
    int W = 4;
    while (W < 64 && W < N) W <<= 1;
    const int64_t groups = nnz;
    int64_t blocks = (groups + (int64_t)kWaves * (64 / W) - 1) / ((int64_t)kWaves * (64 / W));
    if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 workgroups per CU
    hipLaunchKernelGGL(atomic_scatter_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, rowptr, colind, in, out,
                       (int)M, nnz, (int)N, W);
    return hipGetLastError();
}

}  // namespace gespmm
