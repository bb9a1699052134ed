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
#include <stdlib.h>

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

// Yardstick, not a product path: a plain streaming copy — one non-temporal dwordx4 load + store per lane, one workgroup per 4 KB
// (6.5 TB/s read + write on the MI355X; the grid-stride form of the same copy and torch's own copy_ reach 5.3, four loads in flight
// per lane 5.5-6.2: profiles/r05/copy_yardstick.log — the yardstick is the fastest copy we can write, not a convenient one). bench.py times it in the same process as the product to price `roofline.ceiling_frac` with the rate
// THIS box reaches for read + write traffic (MI355X_MICROARCH.md quotes 6.29 TB/s for a copy; boxes differ by a few percent).
using cf4 = float __attribute__((ext_vector_type(4)));
// (shapes measured on the MI355X, profiles/r05/copy_yardstick.log; GESPMM_COPY_MODE picks one for that experiment)
__global__ __launch_bounds__(kThreads) void copy_kernel(const cf4* __restrict__ src, cf4* __restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
__global__ __launch_bounds__(kThreads) void copy_unrolled_kernel(const cf4* __restrict__ src, cf4* __restrict__ dst, int64_t n4) {
    // a workgroup moves U consecutive 4 KB pieces: U loads in flight per lane, then U stores
    const int64_t base = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
    cf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * kThreads;
        if (i < n4) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * kThreads;
        if (i < n4) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
}

__global__ void copy_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t begin, int64_t n) {
    const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

hipError_t launch_copy(const float* src, float* dst, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        static const int mode = getenv("GESPMM_COPY_MODE") ? atoi(getenv("GESPMM_COPY_MODE")) : 5;  // 5: the fastest shape measured
        const cf4* s4 = reinterpret_cast<const cf4*>(src);
        cf4* d4 = reinterpret_cast<cf4*>(dst);
        auto grid_u = [&](int U) { return dim3((unsigned)((n4 + (int64_t)kThreads * U - 1) / ((int64_t)kThreads * U))); };
        if (mode == 1) hipLaunchKernelGGL((copy_unrolled_kernel<1, false>), grid_u(1), dim3(kThreads), 0, st, s4, d4, n4);
        else if (mode == 2) hipLaunchKernelGGL((copy_unrolled_kernel<4, false>), grid_u(4), dim3(kThreads), 0, st, s4, d4, n4);
        else if (mode == 3) hipLaunchKernelGGL((copy_unrolled_kernel<4, true>), grid_u(4), dim3(kThreads), 0, st, s4, d4, n4);
        else if (mode == 4) hipLaunchKernelGGL((copy_unrolled_kernel<8, false>), grid_u(8), dim3(kThreads), 0, st, s4, d4, n4);
        else if (mode == 5) hipLaunchKernelGGL((copy_unrolled_kernel<1, true>), grid_u(1), dim3(kThreads), 0, st, s4, d4, n4);
        else {
            int64_t blocks = (n4 + kThreads - 1) / kThreads;
            if (blocks > 256 * 16) blocks = 256 * 16;
            hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, s4, d4, n4);
        }
    }
    const int64_t done = n4 * 4;
    if (done < n) {
        const int64_t rest = n - done;
        if (rest > (int64_t)0x7fffffff * 256) return hipErrorInvalidValue;
        hipLaunchKernelGGL(copy_tail_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, st, src, dst, done, n);
    }
    return hipGetLastError();
}

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
