// spmm_device.h — device-side helpers shared by the kernel translation units (spmm_kernels.hip, spmm_stream_plan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "spmm_kernels.h"

namespace gespmm {

// ----------------------------------------------------------------------------- small helpers

template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { using type = float __attribute__((ext_vector_type(4))); };

template <int V>
__device__ __forceinline__ void load_vec(float (&dst)[V], const char* base) {
    using T = typename VecT<V>::type;
    T v = *reinterpret_cast<const T*>(base);
    if constexpr (V == 1) {
        dst[0] = v;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = v[i];
    }
}

template <int V>
__device__ __forceinline__ void load_vec_nt(float (&dst)[V], const char* base) {
    using T = typename VecT<V>::type;
    T v = __builtin_nontemporal_load(reinterpret_cast<const T*>(base));
    if constexpr (V == 1) {
        dst[0] = v;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = v[i];
    }
}

template <int V, bool NT>
__device__ __forceinline__ void store_vec(float* p, const float (&src)[V]) {
    using T = typename VecT<V>::type;
    T v;
    if constexpr (V == 1) {
        v = src[0];
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = src[i];
    }
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
    else *reinterpret_cast<T*>(p) = v;
}

// Store with system scope (`sc1`): the line is written through instead of staying in the XCD's L2. C is written once and never
// read by the launch, and in the cache-resident regimes every C line kept in L2 evicts a B row somebody is about to reuse:
// 512-byte rows gathered from a 2 MB table while 1 GB of rows is written run at 25.5 TB/s with plain stores, 27.1 with `nt`,
// 31.3 with `sc1` (31.9 without any store; profiles/r02/l2_gather_steps.log).
template <int V>
__device__ __forceinline__ void store_vec_sc1(float* p, const float (&src)[V]) {
    using T = typename VecT<V>::type;
    T v;
    if constexpr (V == 1) {
        v = src[0];
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = src[i];
        if constexpr (V == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    }
}

// Workgroup id -> work item id such that XCD x (which receives ids == x mod 8)
// gets a contiguous slice of the item range. Bijective for every n.
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Ordering of a wavefront's own LDS traffic (write by lane i, read by lane j of
// the same wavefront). DS operations of one wavefront execute in issue order, so
// no hardware barrier is needed — only the compiler must keep the order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// CSR stream loads are non-temporal: each entry is used once per launch. (Plain loads
// were measured too — no gain, profiles/r01/kernel_generations_cache_regimes.log.)
template <typename T>
__device__ __forceinline__ T load_csr(const T* p) {
    return __builtin_nontemporal_load(p);
}

template <int RED, bool VALUED>
__device__ __forceinline__ float combine(float acc, float a, float b) {
    if constexpr (RED == kReduceMax) {
        return fmaxf(acc, b);
    } else if constexpr (VALUED) {
        return __builtin_fmaf(a, b, acc);
    } else {
        return acc + b;  // A == 1: identical to fma(1, b, acc)
    }
}

}  // namespace gespmm
