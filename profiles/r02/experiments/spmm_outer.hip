// spmm_outer.hip — "task-outer" kernel for row-clustered plans: every DISTINCT B row of a task is loaded once, into
// registers, and applied to all the rows of the task that use it.
//
// Rows that a clustered plan (plan.cpp) puts next to each other share most of their columns (a community of a
// co-purchase graph is a near-clique). The streaming kernels still gather a B row once per non-zero. Here a task is a
// handful of rows (<= 8) and the sorted union of their columns (<= 32); the whole 64-lane wavefront walks that column
// list — one `global_load_dwordx{V}` per distinct column, eight in flight — and for every column runs through the
// entries that use it, adding val * B[col, :] to the accumulator of the entry's row. All rows' accumulators live in
// registers (8 rows x V floats per lane), the task description sits in the registers that loaded it and is read with
// v_readlane at wave-uniform positions, so there is no LDS, no cross-lane traffic and no per-lane control flow at all.
//
// Order of the additions: the columns of a task are walked in ASCENDING order, and a row takes part in a multi-row
// task only if its own columns are strictly ascending (the plan checks) — so every row still adds its non-zeros in
// its own CSR order, one fused multiply-add each: the bits of every other variant (spmm_test.cu:182-203 semantics).
// Rows with unsorted or repeated columns, or too long for one record, are tasks of their own whose "columns" are their
// entries in CSR order; rows longer than a record are a chain of records walked by one wavefront that keeps the
// accumulator (flag bits as in spmm_ldsrow.hip).
//
// Lanes x V floats cover a column tile of 64 V columns (N = 128: V = 2; N = 256: V = 4; wider N: several tiles).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "spmm_kernels.h"

namespace gespmm {

namespace {

template <int V> struct OVec;
template <> struct OVec<1> { using type = float; };
template <> struct OVec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct OVec<4> { using type = float __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ int xcd_contiguous_o(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int RED, bool VALUED>
__device__ __forceinline__ float combine_o(float acc, float a, float b) {
    if constexpr (RED == kReduceMax) return fmaxf(acc, b);
    else if constexpr (VALUED) return __builtin_fmaf(a, b, acc);
    else return acc + b;
}

template <int V, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_outer_kernel(OuterArgs a) {
    constexpr int U = 8;  // distinct B rows in flight
    using T = typename OVec<V>::type;
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int item = xcd_contiguous_o(blockIdx.x, a.nblk * a.ntile);
    int tile = 0, rb = item;
    if (a.ntile > 1) {
        tile = item % a.ntile;
        rb = item / a.ntile;
    }
    int rec_i = rb * kWaves + wave;
    if (rec_i >= a.nrec) return;
    const int32_t* rec = a.recs + (size_t)rec_i * kOutWords;
    int4 h = *reinterpret_cast<const int4*>(rec);
    int flags = __builtin_amdgcn_readfirstlane(h.w);
    if (flags & 1) return;  // continuation of a long row: the wavefront of its first record walks the chain

    const int col0 = tile * (64 * V) + lane * V;
    const bool colok = col0 < a.N;  // N % V == 0
    const off_t rowbytes = (off_t)a.N * 4u;
    const off_t cbyte = colok ? (off_t)col0 * 4u : (off_t)0;
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;

    float acc[kOutRows][V];
#pragma unroll
    for (int r = 0; r < kOutRows; ++r)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[r][i] = init;

    for (;;) {  // the records of this task's chain (one, unless the row is longer than a record)
        const int nrows = __builtin_amdgcn_readfirstlane(h.x);
        const int ndist = __builtin_amdgcn_readfirstlane(h.z);
        // the record: one word load (C rows in lanes 0-7, distinct columns in lanes 8-39), the values, two byte loads
        const int w = rec[kOutOffCrow + (lane < 40 ? lane : 39)];
        int vb = 0;
        if constexpr (VALUED) vb = rec[kOutOffVal + lane];
        const int erow = reinterpret_cast<const uint8_t*>(rec)[kOutOffRowBytes + lane];
        const int cptr = reinterpret_cast<const uint8_t*>(rec)[kOutOffCptrBytes + (lane <= kOutDistinct ? lane : kOutDistinct)];

        for (int j0 = 0; j0 < ndist; j0 += U) {
            // all U loads are issued unconditionally (slots past the end re-read the last column: same cache line) —
            // a branch around a load, even a wave-uniform one, makes hipcc wait for every load on its own
            T b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ju = (j0 + u < ndist) ? j0 + u : ndist - 1;
                const int c = __builtin_amdgcn_readlane(w, 8 + ju);
                b[u] = *reinterpret_cast<const T*>(Bbase + (off_t)((off_t)(uint32_t)c * rowbytes + cbyte));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < ndist) {
                    const int e0 = __builtin_amdgcn_readlane(cptr, j0 + u);
                    const int e1 = __builtin_amdgcn_readlane(cptr, j0 + u + 1);
                    for (int e = e0; e < e1; ++e) {
                        const int r = __builtin_amdgcn_readlane(erow, e);
                        const float v = VALUED ? __int_as_float(__builtin_amdgcn_readlane(vb, e)) : 1.0f;
                        // wave-uniform row index: a scalar jump into one of kOutRows register sets
                        switch (r) {
#define GESPMM_OUT_CASE(k)                                                                                     \
    case k:                                                                                                    \
        if constexpr (V == 1) acc[k][0] = combine_o<RED, VALUED>(acc[k][0], v, b[u]);                             \
        else {                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < V; ++i) acc[k][i] = combine_o<RED, VALUED>(acc[k][i], v, b[u][i]); \
        }                                                                                                      \
        break;
                            GESPMM_OUT_CASE(0)
                            GESPMM_OUT_CASE(1)
                            GESPMM_OUT_CASE(2)
                            GESPMM_OUT_CASE(3)
                            GESPMM_OUT_CASE(4)
                            GESPMM_OUT_CASE(5)
                            GESPMM_OUT_CASE(6)
                            GESPMM_OUT_CASE(7)
#undef GESPMM_OUT_CASE
                        }
                    }
                }
            }
        }
        if (flags & 2) {  // the (single) row goes on in the next record: keep acc[0]
            ++rec_i;
            rec += kOutWords;
            h = *reinterpret_cast<const int4*>(rec);
            flags = __builtin_amdgcn_readfirstlane(h.w);
            continue;
        }
        // ---- C rows (row k of the record -> C row crow[k])
#pragma unroll
        for (int k = 0; k < kOutRows; ++k) {
            if (k < nrows) {  // wave-uniform
                const int crow = __builtin_amdgcn_readlane(w, k);
                if (colok) {
                    float* dst = a.C + (size_t)crow * (size_t)a.N + col0;
                    T o;
                    if constexpr (V == 1) o = acc[k][0];
                    else {
#pragma unroll
                        for (int i = 0; i < V; ++i) o[i] = acc[k][i];
                    }
                    *reinterpret_cast<T*>(dst) = o;
                }
            }
        }
        break;
    }
}

template <int V, bool VALUED, bool IDX64, int RED>
hipError_t launch_v(const OuterArgs& a0, hipStream_t st) {
    OuterArgs a = a0;
    a.ntile = (a.N + 64 * V - 1) / (64 * V);
    a.nblk = (a.nrec + kWaves - 1) / kWaves;
    const int64_t nitems = (int64_t)a.nblk * a.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((spmm_outer_kernel<V, VALUED, IDX64, RED>), dim3((unsigned)nitems), dim3(kThreads), 0, st, a);
    return hipGetLastError();
}

template <bool VALUED, bool IDX64, int RED>
hipError_t launch_vs(const OuterArgs& a, int V, hipStream_t st) {
    switch (V) {
        case 1: return launch_v<1, VALUED, IDX64, RED>(a, st);
        case 2: return launch_v<2, VALUED, IDX64, RED>(a, st);
        case 4: return launch_v<4, VALUED, IDX64, RED>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// Floats per lane: 64 lanes x V floats make a column tile. N >= 256 (multiple of 4): 4; N >= 128 (even): 2; N >= 64: 1;
// narrower N would leave lanes idle: not served (0).
int outer_vec_width(int64_t N) {
    if (N >= 256 && N % 4 == 0) return 4;
    if (N >= 128 && N % 2 == 0) return 2;
    if (N >= 64) return 1;
    return 0;
}

hipError_t launch_spmm_outer(const OuterArgs& a, bool valued, bool idx64, int reduce, hipStream_t st) {
    const int V = outer_vec_width(a.N);
    if (V == 0) return hipErrorInvalidValue;
    if (reduce == kReduceMax) {
        if (valued) return hipErrorInvalidValue;
        return idx64 ? launch_vs<false, true, kReduceMax>(a, V, st) : launch_vs<false, false, kReduceMax>(a, V, st);
    }
    if (valued) return idx64 ? launch_vs<true, true, kReduceSum>(a, V, st) : launch_vs<true, false, kReduceSum>(a, V, st);
    return idx64 ? launch_vs<false, true, kReduceSum>(a, V, st) : launch_vs<false, false, kReduceSum>(a, V, st);
}

}  // namespace gespmm
