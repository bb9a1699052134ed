// sddmm_kernels.hip — sampled dense-dense product on a sparse pattern (gfx950).
//
//     out[e] = sum_{j < N} D1[row(e), j] * D2[col(e), j]        (pattern order)
//
// Reference semantics: pytorch-custom/sddmm.cu:7-424 + computeUtil.h:11-28,115-124
// (COO: row(e) = rowind[e]; CSR: row(e) found by binary search in rowptr). The
// reference packs 4 edges per 8/16/32-lane slice of a 32-lane warp and needs
// 16-byte-aligned index arrays plus single-edge tail blocks; none of that shape
// is kept. Here a W-lane group of a 64-lane wavefront owns one edge at a time, with W
// chosen so that a lane walks ~2 dwordx4 vectors (8 scalars for odd N) of both rows; two
// to four edges per group are in flight, their slices requested before the first FMA, and
// the W partial dot products meet in an xor butterfly (cross-lane ds_bpermute / DPP
// moves). Edges of a wavefront are consecutive, so the out[] stores and the index loads
// coalesce. In CSR form, long rows (mean degree >= 64) are walked a row per wavefront with the D1
// slice in registers (sddmm_slab_kernel), one launch per ~6 MB column slab of D2 when the pattern
// is dense enough for cache blocking, one launch otherwise. Summation order is not sequential (neither is the
// reference's shuffle tree): parity for SDDMM is tolerance-based; COO, CSR and the
// cache-blocked form agree with each other bit for bit.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_kernels.h"
#include "workspace.h"

namespace gespmm {

template <int V> struct SdVec;
template <> struct SdVec<1> { using type = float; };
template <> struct SdVec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct SdVec<4> { using type = float __attribute__((ext_vector_type(4))); };

// Row that owns CSR position e: largest r with rowptr[r] <= e (empty rows skipped).
__device__ __forceinline__ int row_of_edge(const int32_t* __restrict__ rowptr, int M, int e) {
    int lo = 0, hi = M;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The same for a wavefront-uniform e, all 64 lanes probing at once: every round cuts [lo, hi) into 64 pieces with one
// coalesced-ish load per lane and a ballot, so a matrix of 3 * 10^5 rows takes 4 dependent round trips instead of 18
// (the prologue of every CSR-form wavefront: com-Amazon-shaped N = 128, CSR 166 -> COO's 148 us was this search).
__device__ __forceinline__ int row_of_edge_wave(const int32_t* __restrict__ rowptr, int M, int e, int lane) {
    int lo = 0, hi = M;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + (lane + 1) * step;
        const bool le = idx < hi && rowptr[idx] <= e;  // monotone in the lane: a prefix of the lanes says yes
        const int cnt = __popcll(__ballot(le));
        lo += cnt * step;
        hi = (lo + step < hi) ? lo + step : hi;
    }
    return lo;
}

template <int V, int W, bool CSR>
__global__ __launch_bounds__(kThreads) void sddmm_kernel(const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ colind,
                                                          const float* __restrict__ D1,
                                                          const float* __restrict__ D2, float* __restrict__ out,
                                                          int M, int nnz, int N, int epw) {
    constexpr int G = 64 / W;
    constexpr int EPW = 256;  // most edges a CSR-form wavefront owns (epw <= EPW): one row search per epw edges
    constexpr int UE = 4;    // edges per lane group per step
    constexpr int IT = (V == 4) ? 2 : (V == 2) ? 4 : 8;  // vectors per lane that cover a row under launch_sddmm's width rule
    using T = typename SdVec<V>::type;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;

    // ---- the wavefront's edges [e_lo, e_hi) and where their (row, column) ids come from.
    // COO form: G * UE consecutive edges, ids straight from the caller's arrays.
    // CSR form: epw consecutive edges. ONE wavefront-wide search finds the row of the first edge; the rows of all
    // its edges then lie in a window of at most epw + 1 row pointers, staged in LDS, where every lane resolves the
    // row of one edge (the reference searches rowptr in global memory once per edge, computeUtil.h:11-28). After that
    // both forms run the SAME loop: UE edges per lane group in flight, all slices requested before the first FMA.
    __shared__ int s_rp[CSR ? kWaves : 1][CSR ? EPW + 2 : 1];
    __shared__ int s_row[CSR ? kWaves : 1][CSR ? EPW : 1];
    __shared__ int s_col[CSR ? kWaves : 1][CSR ? EPW : 1];
    const int per_wave = CSR ? epw : G * UE;
    const int e_lo = (blockIdx.x * kWaves + wave) * per_wave;
    if (e_lo >= nnz) return;  // whole wavefront
    const int e_hi = (e_lo + per_wave < nnz) ? e_lo + per_wave : nnz;
    if constexpr (CSR) {
        const int r0 = row_of_edge_wave(rows, M, e_lo, lane);
        for (int i = lane; i < epw + 2; i += 64) s_rp[wave][i] = (r0 + i <= M) ? rows[r0 + i] : 0x7fffffff;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = e_lo + lane; e < e_hi; e += 64) {
            int lo = 0;
            if (s_rp[wave][1] <= e) {  // (long rows: the wavefront's edges usually lie inside ONE row — no search then)
                // largest i in [0, epw + 1] with s_rp[i] <= e   (s_rp[0] = rowptr[r0] <= e_lo <= e)
                int hi = epw + 1;
                if (s_rp[wave][hi] <= e) {  // more than epw empty rows in the window: search the whole array
                    lo = row_of_edge(rows, M, e) - r0;
                    hi = lo + 1;
                }
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_rp[wave][mid] <= e) lo = mid;
                    else hi = mid;
                }
            }
            s_row[wave][e - e_lo] = r0 + lo;
            s_col[wave][e - e_lo] = colind[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto row_id = [&](int e) { return CSR ? s_row[wave][e - e_lo] : rows[e]; };
    auto col_id = [&](int e) { return CSR ? s_col[wave][e - e_lo] : colind[e]; };

    for (int rb = 0; rb < e_hi - e_lo; rb += G * UE) {  // (COO form: one round)
        const int ebase = e_lo + rb + g * UE;
        float part[UE];
        if (N <= W * V) {
            // one slice per row: all 2 * UE slices of the group's UE edges are requested before any is used
            T x[UE], y[UE];
            const int j = l * V;
#pragma unroll
            for (int u = 0; u < UE; ++u) {
                const int e = ebase + u;
                x[u] = T{};
                y[u] = T{};
                if (e < e_hi && j < N) {
                    x[u] = *reinterpret_cast<const T*>(D1 + (size_t)row_id(e) * (size_t)N + j);
                    y[u] = *reinterpret_cast<const T*>(D2 + (size_t)col_id(e) * (size_t)N + j);
                }
            }
#pragma unroll
            for (int u = 0; u < UE; ++u) {
                part[u] = 0.0f;
                if constexpr (V == 1) {
                    part[u] = __builtin_fmaf(x[u], y[u], part[u]);
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) part[u] = __builtin_fmaf(x[u][i], y[u][i], part[u]);
                }
            }
        } else if (N <= W * V * IT) {
            // a lane walks up to IT vectors of each row: two edges at a time, all 4*IT slices requested
            // before the first FMA (an edge past the end re-reads the last edge and is dropped)
#pragma unroll
            for (int u0 = 0; u0 < UE; u0 += 2) {
                T x[2][IT], y[2][IT];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = (ebase + u0 + u < e_hi) ? ebase + u0 + u : e_hi - 1;
                    const float* p1 = D1 + (size_t)row_id(e) * (size_t)N;
                    const float* p2 = D2 + (size_t)col_id(e) * (size_t)N;
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int j = l * V + it * W * V;
                        x[u][it] = *reinterpret_cast<const T*>(p1 + (j < N ? j : 0));
                        y[u][it] = *reinterpret_cast<const T*>(p2 + (j < N ? j : 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float acc = 0.0f;
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        if (l * V + it * W * V < N) {  // the FMAs of the plain loop, in its order
                            if constexpr (V == 1) {
                                acc = __builtin_fmaf(x[u][it], y[u][it], acc);
                            } else {
#pragma unroll
                                for (int i = 0; i < V; ++i) acc = __builtin_fmaf(x[u][it][i], y[u][it][i], acc);
                            }
                        }
                    }
                    part[u0 + u] = acc;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < UE; ++u) {
                const int e = ebase + u;
                part[u] = 0.0f;
                if (e < e_hi) {
                    const float* p1 = D1 + (size_t)row_id(e) * (size_t)N;
                    const float* p2 = D2 + (size_t)col_id(e) * (size_t)N;
                    for (int j = l * V; j < N; j += W * V) {
                        const T x = *reinterpret_cast<const T*>(p1 + j);
                        const T y = *reinterpret_cast<const T*>(p2 + j);
                        if constexpr (V == 1) {
                            part[u] = __builtin_fmaf(x, y, part[u]);
                        } else {
#pragma unroll
                            for (int i = 0; i < V; ++i) part[u] = __builtin_fmaf(x[i], y[i], part[u]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
#pragma unroll
            for (int m = W >> 1; m > 0; m >>= 1) part[u] += __shfl_xor(part[u], m, 64);
            if (l == 0 && ebase + u < e_hi) out[ebase + u] = part[u];
        }
    }
}

template <int V, int W>
__global__ __launch_bounds__(kThreads) void sddmm_slab_kernel(const int32_t* __restrict__ row_begin,
                                                               const int32_t* __restrict__ row_end,
                                                               const int32_t* __restrict__ colind,
                                                               const float* __restrict__ D1,
                                                               const float* __restrict__ D2, float* __restrict__ out,
                                                               int M, int N, int rows_per_wave) {
    constexpr int G = 64 / W;
    constexpr int IT = (V == 4) ? 2 : (V == 2) ? 4 : 8;  // vectors per lane that cover a row (launch_sddmm's width rule)
    constexpr int UE = 4;                 // edges per lane group in flight
    using T = typename SdVec<V>::type;
    __shared__ int s_col[kWaves][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const bool in_regs = N <= W * V * IT;  // else (N > 512 at V = 4): plain per-edge loop
    const int row0 = (blockIdx.x * kWaves + wave) * rows_per_wave;
    for (int i = 0; i < rows_per_wave; ++i) {
        const int r = row0 + i;
        if (r >= M) break;  // wave-uniform
        const int b = row_begin[r], e = row_end[r];
        if (b >= e) continue;
        const float* p1 = D1 + (size_t)r * (size_t)N;
        T x[IT];
        if (in_regs) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int j = l * V + it * W * V;
                x[it] = *reinterpret_cast<const T*>(p1 + (j < N ? j : 0));
            }
        }
        for (int base = b; base < e; base += 64) {
            const int cnt = (e - base < 64) ? e - base : 64;
            if (lane < cnt) s_col[wave][lane] = colind[base + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (in_regs) {
                // UE edges per group per step: their D2 slices are requested back to back (an edge past the
                // end re-reads the step's first edge and is dropped), the row's D1 slice sits in registers
                for (int k = g; k < cnt; k += G * UE) {
                    T y[UE][IT];
#pragma unroll
                    for (int u = 0; u < UE; ++u) {
                        const int kk = (k + u * G < cnt) ? k + u * G : k;
                        const float* p2 = D2 + (size_t)s_col[wave][kk] * (size_t)N;
#pragma unroll
                        for (int it = 0; it < IT; ++it) {
                            const int j = l * V + it * W * V;
                            y[u][it] = *reinterpret_cast<const T*>(p2 + (j < N ? j : 0));
                        }
                    }
                    float part[UE];
#pragma unroll
                    for (int u = 0; u < UE; ++u) {
                        part[u] = 0.0f;
#pragma unroll
                        for (int it = 0; it < IT; ++it) {
                            if (l * V + it * W * V < N) {  // same FMAs, same order as the streaming loop
                                if constexpr (V == 1) {
                                    part[u] = __builtin_fmaf(x[it], y[u][it], part[u]);
                                } else {
#pragma unroll
                                    for (int q = 0; q < V; ++q) part[u] = __builtin_fmaf(x[it][q], y[u][it][q], part[u]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int m = W >> 1; m > 0; m >>= 1)
#pragma unroll
                        for (int u = 0; u < UE; ++u) part[u] += __shfl_xor(part[u], m, 64);
#pragma unroll
                    for (int u = 0; u < UE; ++u)
                        if (l == 0 && k + u * G < cnt) out[base + k + u * G] = part[u];
                }
            } else {
                for (int k = g; k < cnt; k += G) {
                    const float* p2 = D2 + (size_t)s_col[wave][k] * (size_t)N;
                    float part = 0.0f;
                    for (int j = l * V; j < N; j += W * V) {
                        const T xx = *reinterpret_cast<const T*>(p1 + j);
                        const T yy = *reinterpret_cast<const T*>(p2 + j);
                        if constexpr (V == 1) {
                            part = __builtin_fmaf(xx, yy, part);
                        } else {
#pragma unroll
                            for (int q = 0; q < V; ++q) part = __builtin_fmaf(xx[q], yy[q], part);
                        }
                    }
#pragma unroll
                    for (int m = W >> 1; m > 0; m >>= 1) part += __shfl_xor(part, m, 64);
                    if (l == 0) out[base + k] = part;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// Row-walking (one wavefront per row) is skew-sensitive — a hub row is walked by one wavefront — and only pays
// for long rows: parity with the edge-parallel form at degree 32-48, up to 2x at degree >= 64 on patterns too
// small to block (profiles/r01/sddmm_heuristic_audit.log).
constexpr int64_t kSddmmRowWalkMinDegree = 64;

template <int V>
static hipError_t sddmm_slab_w(int W, const int32_t* rb, const int32_t* re, const int32_t* colind, const float* D1,
                               const float* D2, float* out, int M, int N, hipStream_t st) {
    constexpr int kRowsPerWave = 2;
    const int nblk = (M + kWaves * kRowsPerWave - 1) / (kWaves * kRowsPerWave);
#define GESPMM_SDS(WW)                                                                                          \
    case WW:                                                                                                     \
        hipLaunchKernelGGL((sddmm_slab_kernel<V, WW>), dim3(nblk), dim3(kThreads), 0, st, rb, re, colind, D1, D2, \
                           out, M, N, kRowsPerWave);                                                             \
        return hipGetLastError();
    switch (W) {
        GESPMM_SDS(4)
        GESPMM_SDS(8)
        GESPMM_SDS(16)
        GESPMM_SDS(32)
        GESPMM_SDS(64)
    }
#undef GESPMM_SDS
    return hipErrorInvalidValue;
}

template <int V, bool CSR>
static hipError_t sddmm_w(int W, const int32_t* rows, const int32_t* colind, const float* D1, const float* D2,
                          float* out, int M, int nnz, int N, hipStream_t st) {
#define GESPMM_SD(WW)                                                                                         \
    case WW: {                                                                                                 \
        constexpr int G = 64 / WW;                                                                             \
        /* edges per wavefront: COO G * UE (UE = 4); CSR 256 — one row search per 256 edges — unless that leaves \
           the chip short of wavefronts (small patterns: pubmed-sized N = 128 ran 18.4 us against COO's 9.8) */   \
        int per_wave = G * 4;                                                                                  \
        if (CSR) per_wave = (nnz >= 256 * 16384) ? 256 : (nnz >= 64 * 16384) ? 64 : (G * 4 > 16 ? G * 4 : 16); \
        const int nblk = (int)(((int64_t)nnz + kWaves * per_wave - 1) / (kWaves * per_wave));                  \
        hipLaunchKernelGGL((sddmm_kernel<V, WW, CSR>), dim3(nblk), dim3(kThreads), 0, st, rows, colind, D1, D2, \
                           out, M, nnz, N, per_wave);                                                          \
        return hipGetLastError();                                                                              \
    }
    switch (W) {
        GESPMM_SD(4)
        GESPMM_SD(8)
        GESPMM_SD(16)
        GESPMM_SD(32)
        GESPMM_SD(64)
    }
#undef GESPMM_SD
    return hipErrorInvalidValue;
}

hipError_t launch_sddmm(const int32_t* rows, bool csr, const int32_t* colind, const float* D1, const float* D2,
                        float* out, int64_t M, int64_t nnz, int64_t N, int flags, hipStream_t st) {
    if (nnz == 0) return hipSuccess;
    int V = 4;
    while (V > 1 && ((N % V) != 0 || (reinterpret_cast<uintptr_t>(D1) % (4u * V)) != 0 ||
                     (reinterpret_cast<uintptr_t>(D2) % (4u * V)) != 0))
        V >>= 1;
    // Lanes per edge: a lane walks ~2 dwordx4 vectors (8 scalars when the rows allow no vector loads)
    // of both rows, so a wavefront has 64/W edges in flight and the butterfly is log2(W) steps.
    // Measured against "just enough lanes to cover N" (profiles/r01/sddmm_group_width.log):
    // N=41 2.4-2.8x, N=64 1.35x, N=128 1.2x faster on reddit-like, equal or better on com-Amazon-like.
    // Both forms use the same width, so COO and CSR results agree bit for bit.
    const int64_t per_lane = (V == 4) ? 2 : (V == 2) ? 4 : 8;
    int W = 4;
    while (W < 64 && (int64_t)W * V * per_lane < N) W <<= 1;
    const int m = (int)M, z = (int)nnz, n = (int)N;
    if (csr && M > 0 && (flags & kSddmmNoSlab) == 0) {
        // Dense pattern (mean degree >= 64, >= 4.5 KB gathered per row and slab): cache-blocked form, ~6 MB
        // slabs of D2. The number of D2 rows is not part of the call: the pattern is taken as square for
        // the slab count (columns past M land in the last slab — fewer hits, same result). Needs a
        // stream-ordered temporary for the split points, so not on a stream under capture.
        const int64_t avg_deg = nnz / M;
        int64_t slab_rows = (6 << 20) / (N * 4 > 0 ? N * 4 : 4);  // 3..6 MB measured best (sddmm_slab.log)
        if (slab_rows < 64) slab_rows = 64;
        const int64_t nslab = (M + slab_rows - 1) / slab_rows;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
        // (the edges of a row that fall into one slab must still gather >= ~4.5 KB of D2: with less the per-row
        // overhead of 'nslab' launches outweighs the L2 hits —
        // profiles/r01/sddmm_heuristic_audit.log; M = 10^6, degree 100, 41 slabs was 4.5x slower than streaming)
        if (!capturing && N * 4 >= 256 && nslab >= 4 && nslab <= 4096 && avg_deg >= 64 &&
            avg_deg * N * 4 >= 4608 * nslab) {
            int32_t* split = nullptr;
            hipError_t e = workspace_alloc(reinterpret_cast<void**>(&split), (size_t)(nslab + 1) * (size_t)M * 4, st);
            if (e != hipSuccess) return e;
            e = launch_slabplan(rows, colind, split, m, (int)nslab, (int)slab_rows, st);
            for (int64_t sl = 0; sl < nslab && e == hipSuccess; ++sl) {
                const int32_t* rb = split + (size_t)sl * M;
                const int32_t* re = split + (size_t)(sl + 1) * M;
                if (V == 4) e = sddmm_slab_w<4>(W, rb, re, colind, D1, D2, out, m, n, st);
                else if (V == 2) e = sddmm_slab_w<2>(W, rb, re, colind, D1, D2, out, m, n, st);
                else e = sddmm_slab_w<1>(W, rb, re, colind, D1, D2, out, m, n, st);
            }
            const hipError_t ef = workspace_free(split, st);
            return e != hipSuccess ? e : ef;
        }
        (void)hipGetLastError();
        // Long rows but not worth blocking: the same row-walking kernel over the whole row ([rowptr[r], rowptr[r+1])
        // is "one slab") — D1 slice in registers, four edges in flight, no per-edge row search. The edge-parallel
        // form below stays for short rows, where a row per wavefront would leave lanes idle.
        if (avg_deg >= kSddmmRowWalkMinDegree) {
            if (V == 4) return sddmm_slab_w<4>(W, rows, rows + 1, colind, D1, D2, out, m, n, st);
            if (V == 2) return sddmm_slab_w<2>(W, rows, rows + 1, colind, D1, D2, out, m, n, st);
            return sddmm_slab_w<1>(W, rows, rows + 1, colind, D1, D2, out, m, n, st);
        }
    }
    if (csr) {
        if (V == 4) return sddmm_w<4, true>(W, rows, colind, D1, D2, out, m, z, n, st);
        if (V == 2) return sddmm_w<2, true>(W, rows, colind, D1, D2, out, m, z, n, st);
        return sddmm_w<1, true>(W, rows, colind, D1, D2, out, m, z, n, st);
    }
    if (V == 4) return sddmm_w<4, false>(W, rows, colind, D1, D2, out, m, z, n, st);
    if (V == 2) return sddmm_w<2, false>(W, rows, colind, D1, D2, out, m, z, n, st);
    return sddmm_w<1, false>(W, rows, colind, D1, D2, out, m, z, n, st);
}

}  // namespace gespmm
