// csr2csc.hip — CSR -> CSC on the device (transpose of the sparsity pattern).
//
// Replaces csr2csc_cuda()/csr2cscKernel (pytorch-custom/spmm_kernel.cu:381-476),
// which calls cusparseCsr2cscEx2 through a handle that is never created and is
// therefore unusable as shipped. The op's backward pass is SpMM on the CSC
// arrays (op.py:20-36), so the transpose must be DETERMINISTIC: entries of one
// column stay in ascending row order. That is a stable sort of the CSR positions
// by column index:
//   1. histogram of colind -> colptr (integer atomics: order-independent result),
//      inclusive scan in place;
//   2. stable LSD radix sort of (colind[p], p) pairs (rocPRIM);
//   3. gather: rowind[i] = row owning position perm[i], csc_val[i] = csr_val[perm[i]].

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "spmm_kernels.h"

namespace gespmm {

namespace {

constexpr int64_t kAlign = 256;
inline int64_t align_up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

// Column indices are the caller's and are NOT validated against K anywhere else on the device (the SpMM / SDDMM
// kernels trust them, as the reference's do); here an index outside [0, K) would be an out-of-bounds atomic on
// colptr, so such entries are simply not counted (the transpose of a malformed matrix is then short, not corrupt).
__global__ void k_hist_iota(const int32_t* __restrict__ colind, int32_t* __restrict__ colptr,
                            int32_t* __restrict__ iota, int nnz, int K) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) {
        const int c = colind[p];
        if ((uint32_t)c < (uint32_t)K) atomicAdd(&colptr[c + 1], 1);
        iota[p] = p;
    }
}

__global__ void k_gather(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                         const float* __restrict__ csr_val, int32_t* __restrict__ rowind,
                         float* __restrict__ csc_val, int M, int nnz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int p = perm[i];
    int lo = 0, hi = M;  // rowptr[lo] <= p < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= p) lo = mid;
        else hi = mid;
    }
    rowind[i] = lo;
    if (csr_val) csc_val[i] = csr_val[p];
}

int key_bits(int64_t K) {
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < K) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t nnz, int64_t K) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                              (int32_t*)nullptr, (size_t)nnz, 0, key_bits(K));
    return bytes;
}

size_t scan_temp_bytes(int64_t K) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)(K + 1),
                            rocprim::plus<int32_t>());
    return bytes;
}

}  // namespace

int64_t csr2csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
    (void)M;
    const int64_t n = nnz > 0 ? nnz : 1;
    const size_t t1 = sort_temp_bytes(n, K), t2 = scan_temp_bytes(K);
    return 3 * align_up(n * 4) + align_up((int64_t)(t1 > t2 ? t1 : t2)) + kAlign;
}

hipError_t launch_csr2csc(const int32_t* rowptr, const int32_t* colind, const float* csr_val, int32_t* colptr,
                          int32_t* rowind, float* csc_val, int64_t M, int64_t K, int64_t nnz, void* workspace,
                          hipStream_t st) {
    hipError_t e = hipMemsetAsync(colptr, 0, (size_t)(K + 1) * 4, st);
    if (e != hipSuccess) return e;
    if (nnz == 0) return hipSuccess;

    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + kAlign - 1) / kAlign * kAlign);
    int32_t* iota = reinterpret_cast<int32_t*>(w);
    w += align_up(nnz * 4);
    int32_t* keys_out = reinterpret_cast<int32_t*>(w);
    w += align_up(nnz * 4);
    int32_t* perm = reinterpret_cast<int32_t*>(w);
    w += align_up(nnz * 4);
    void* temp = w;

    const int threads = 256;
    const int blocks = (int)((nnz + threads - 1) / threads);
    hipLaunchKernelGGL(k_hist_iota, dim3(blocks), dim3(threads), 0, st, colind, colptr, iota, (int)nnz, (int)K);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    size_t tb = scan_temp_bytes(K);
    e = rocprim::inclusive_scan(temp, tb, colptr, colptr, (size_t)(K + 1), rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return e;

    tb = sort_temp_bytes(nnz, K);
    e = rocprim::radix_sort_pairs(temp, tb, colind, keys_out, (const int32_t*)iota, perm, (size_t)nnz, 0,
                                  key_bits(K), st);
    if (e != hipSuccess) return e;

    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, st, rowptr, (const int32_t*)perm, csr_val, rowind,
                       csc_val, (int)M, (int)nnz);
    return hipGetLastError();
}

}  // namespace gespmm
