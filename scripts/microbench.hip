// microbench.hip — hardware yardsticks for the SpMM gather (tuning aid, not product):
//   mb_stream_read : grid-stride dwordx4 streaming read of a buffer (sum kept live)
//   mb_row_gather  : random 512-byte row gathers (a half-wave per row, 8 rows in flight
//                    per half-wave), no CSR walk, one 512-byte store per `per_store` rows
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/microbench.hip -o /tmp/libmb.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream_read(const f4* __restrict__ p, size_t n, float* out) {
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void k_row_gather(const float* __restrict__ B, const int* __restrict__ idx,
                                                      int nidx, int rowfloats, int per_store, float* __restrict__ C) {
    // half-wave (32 lanes x float4 = 512 B) per row; each half-wave walks a contiguous slice of idx
    const int lane = threadIdx.x & 31;
    const int hw = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int nhw = (gridDim.x * blockDim.x) >> 5;
    const int per = (nidx + nhw - 1) / nhw;
    int k = hw * per, ke = min(nidx, k + per);
    f4 acc = {0, 0, 0, 0};
    int done = 0;
    for (; k + 8 <= ke; k += 8) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f4*)(B + (size_t)idx[k + j] * rowfloats + lane * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
        done += 8;
        if (per_store > 0 && done >= per_store) {
            *(f4*)(C + (size_t)(k / per_store) * rowfloats + lane * 4) = acc;
            done = 0;
            acc = (f4){0, 0, 0, 0};
        }
    }
    for (; k < ke; ++k) acc += *(const f4*)(B + (size_t)idx[k] * rowfloats + lane * 4);  // leftover rows
    if (acc.x == 12345.678f) C[0] = acc.x;
}

// same for 128-byte rows (N = 32): 8 lanes per row, 8 rows per wavefront at a time
__global__ __launch_bounds__(256) void k_row_gather128(const float* __restrict__ B, const int* __restrict__ idx,
                                                         int nidx, int per_store, float* __restrict__ C) {
    const int lane = threadIdx.x & 7;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int ngrp = (gridDim.x * blockDim.x) >> 3;
    const int per = (nidx + ngrp - 1) / ngrp;
    int k = grp * per, ke = min(nidx, k + per);
    f4 acc = {0, 0, 0, 0};
    int done = 0;
    for (; k + 8 <= ke; k += 8) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f4*)(B + (size_t)idx[k + j] * 32 + lane * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
        done += 8;
        if (per_store > 0 && done >= per_store) {
            *(f4*)(C + (size_t)(k / per_store) * 32 + lane * 4) = acc;
            done = 0;
            acc = (f4){0, 0, 0, 0};
        }
    }
    for (; k < ke; ++k) acc += *(const f4*)(B + (size_t)idx[k] * 32 + lane * 4);
    if (acc.x == 12345.678f) C[0] = acc.x;
}
extern "C" int mb_row_gather128(const void* B, const void* idx, int nidx, int per_store, void* C, int blocks,
                                void* stream) {
    hipLaunchKernelGGL(k_row_gather128, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)B,
                       (const int*)idx, nidx, per_store, (float*)C);
    return (int)hipGetLastError();
}

extern "C" int mb_stream_read(const void* p, size_t bytes, void* out, int blocks, void* stream) {
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4*)p, bytes / 16, (float*)out);
    return (int)hipGetLastError();
}
extern "C" int mb_row_gather(const void* B, const void* idx, int nidx, int rowfloats, int per_store, void* C,
                             int blocks, void* stream) {
    hipLaunchKernelGGL(k_row_gather, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)B, (const int*)idx,
                       nidx, rowfloats, per_store, (float*)C);
    return (int)hipGetLastError();
}
