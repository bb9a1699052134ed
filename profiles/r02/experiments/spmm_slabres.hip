// spmm_slabres.hip — cache-blocked SpMM for dense graphs with the partial sums parked in LDS ("slab-resident").
//
// The slab-blocked path of spmm_kernels.hip walks the column slabs of A (= row slabs of B) one LAUNCH per slab and
// carries every row's partial sum from slab to slab through C: with s slabs that is s reads and s writes of C, which
// is what keeps its slabs at 6 MB (reddit-like, N = 128: 19 slabs, 4.5 GB of C traffic) although a 4 MiB L2 only
// holds ~3 MB of B next to the streams passing through it — 40 % of the gathers of that path miss L2
// (profiles/r02/pmcdeep_reddit_slab.txt). Here a workgroup owns 8 rows per lane group for the WHOLE sweep and keeps
// their accumulators in LDS (8 groups x 8 rows x 512 B = 32 KB at N = 128), so the slabs can be as small as the L2
// likes: C is written once, and the only price of a smaller slab is a shorter run of gathers per (row, slab) visit.
//
//   * all workgroups of a launch are resident at once (the launcher cuts the rows into PANELS of that many rows) and
//     walk the slabs in the same order at the same pace — equal work per slab on these graphs — so an XCD's L2 holds
//     the slab everybody is reading; the slabs stream through every L2 once per panel;
//   * per lane group: split points of its rows for slabs s..s+2 sit in a 4-deep LDS ring, the ones of slab s+3 are
//     in flight; the first CSR tile of the NEXT visit is prefetched while the current one gathers;
//   * one visit = the row's entries in [split[s][row], split[s+1][row]) as a U-deep gather stream through the
//     group's LDS tile, accumulator read from / written back to LDS (one ds_read_b128 + one ds_write_b128 per lane).
//
// Every output element is still ONE fp32 chain over the row's entries in CSR order (a value parked in LDS and read
// back is the same value): same bits as every other variant (spmm_test.cu:182-203 semantics).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "spmm_kernels.h"
#include "workspace.h"

namespace gespmm {

namespace {

template <int V> struct RVec;
template <> struct RVec<1> { using type = float; };
template <> struct RVec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct RVec<4> { using type = float __attribute__((ext_vector_type(4))); };

__device__ __forceinline__ void wave_lds_sync_r() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int RED, bool VALUED>
__device__ __forceinline__ float combine_r(float acc, float a, float b) {
    if constexpr (RED == kReduceMax) return fmaxf(acc, b);
    else if constexpr (VALUED) return __builtin_fmaf(a, b, acc);
    else return acc + b;
}

struct SlabResArgs {
    const int32_t* rowptr;
    const int32_t* colind;
    const float* val;
    const float* B;
    float* C;
    const int32_t* split;  // [(nslab + 1)][M] forward-scan split points (launch_slabplan)
    int32_t M, N;
    int32_t s0, s1;        // slabs [s0, s1) of this launch; s0 > 0: the partial sums come from C
    int32_t row_lo, row_hi;  // rows of this launch (a panel)
    int32_t ntile;
    int32_t hub_thr;       // > 0: rows with more entries are left to spmm_hubrow_kernel (skipped here, C untouched)
    float empty;
};

template <int V, int W, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_slabres_kernel(SlabResArgs a) {
    constexpr int G = 64 / W;
    constexpr int T = (W > 32) ? W : 32;  // entries per group tile
    constexpr int E = T / W;
    constexpr int U = 8;
    constexpr int R = kSlabRowsPerGroup;
    using VT = typename RVec<V>::type;
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ VT s_acc[kWaves][G][R][W];
    __shared__ off_t s_off[kWaves][G][T];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? G : 1][VALUED ? T : 1];
    __shared__ int s_sp[kWaves][G][4][R];
    __shared__ int s_hub[kWaves][G][R];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    int tile = 0, rb = blockIdx.x;
    if (a.ntile > 1) {
        tile = (int)blockIdx.x % a.ntile;
        rb = (int)blockIdx.x / a.ntile;
    }
    if (a.row_lo + ((rb * kWaves + wave) * G) * R >= a.row_hi) return;  // whole wavefront past the panel
    const int row0 = a.row_lo + ((rb * kWaves + wave) * G + g) * R;
    int nrows = a.row_hi - row0;
    nrows = nrows < 0 ? 0 : (nrows > R ? R : nrows);

    const int col0 = tile * (W * V) + l * V;
    const bool colok = col0 < a.N;
    const off_t cbytes = colok ? (off_t)col0 * 4u : (off_t)0;
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const off_t rowbytes = (off_t)a.N * 4u;
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;

    // lane l < nrows looks after the split points of row row0 + l (clamped to the row's own CSR range: a stale
    // caller workspace must not turn into out-of-range reads)
    const bool mine = l < nrows;
    int lb = 0, hb = 0;
    if (mine) {
        lb = a.rowptr[row0 + l];
        hb = a.rowptr[row0 + l + 1];
    }
    {
        const bool hub = a.hub_thr > 0 && hb - lb > a.hub_thr;
        if (hub) lb = hb;  // every split point clamps to an empty range
        if (l < R) s_hub[wave][g][l] = hub ? 1 : 0;
    }
    auto ld_split = [&](int s) {
        int v = 0;
        if (mine) {
            v = __builtin_nontemporal_load(a.split + (size_t)s * (size_t)a.M + (size_t)(row0 + l));
            v = v < lb ? lb : (v > hb ? hb : v);
        }
        return v;
    };
    {
        const int v0 = ld_split(a.s0);
        const int v1 = ld_split(a.s0 + 1 <= a.s1 ? a.s0 + 1 : a.s1);
        const int v2 = ld_split(a.s0 + 2 <= a.s1 ? a.s0 + 2 : a.s1);
        if (l < R) {
            s_sp[wave][g][a.s0 & 3][l] = v0;
            s_sp[wave][g][(a.s0 + 1) & 3][l] = v1;
            s_sp[wave][g][(a.s0 + 2) & 3][l] = v2;
        }
    }
    int nxt = (a.s0 + 3 <= a.s1) ? ld_split(a.s0 + 3) : 0;

    // accumulators -> LDS
#pragma unroll
    for (int i = 0; i < R; ++i) {
        VT v;
        if constexpr (V == 1) v = init;
        else
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = init;
        if (a.s0 > 0 && i < nrows && colok)
            v = *reinterpret_cast<const VT*>(a.C + (size_t)(row0 + i) * (size_t)a.N + col0);
        s_acc[wave][g][i][l] = v;
    }
    wave_lds_sync_r();

    // the first CSR tile of the next visit, loaded while the current visit gathers
    int qc[E];
    float qv[E];
    auto prefetch_visit = [&](int s, int i) {
        int b = 0, e = 0;
        if (s < a.s1) {
            b = s_sp[wave][g][s & 3][i];
            e = s_sp[wave][g][(s + 1) & 3][i];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int p = b + l * E + k;
            qc[k] = 0;
            qv[k] = 0.0f;
            if (p < e) {
                qc[k] = __builtin_nontemporal_load(a.colind + p);
                if constexpr (VALUED) qv[k] = __builtin_nontemporal_load(a.val + p);
            }
        }
    };
    prefetch_visit(a.s0, 0);

    for (int s = a.s0; s < a.s1; ++s) {
        if (s > a.s0) {  // advance the ring: split[s + 2] was loaded a slab ago, split[s + 3] starts now
            if (s + 2 <= a.s1 && l < R) s_sp[wave][g][(s + 2) & 3][l] = nxt;
            if (s + 3 <= a.s1) nxt = ld_split(s + 3);
            wave_lds_sync_r();
        }
        for (int i = 0; i < R; ++i) {
            const int gb = s_sp[wave][g][s & 3][i];
            int ge = s_sp[wave][g][(s + 1) & 3][i];
            if (ge < gb) ge = gb;
            int pc[E];
            float pv[E];
#pragma unroll
            for (int k = 0; k < E; ++k) {
                pc[k] = qc[k];
                pv[k] = qv[k];
            }
            if (i + 1 < R) prefetch_visit(s, i + 1);
            else prefetch_visit(s + 1, 0);
            if (gb >= ge) continue;  // this row has nothing in this slab

            float acc[V];
            {
                const VT v = s_acc[wave][g][i][l];
                if constexpr (V == 1) acc[0] = v;
                else
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] = v[k];
            }
            int tbase = gb;
            for (int k = gb; k < ge; k += U) {
                if (k == tbase) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        s_off[wave][g][l * E + e] = (off_t)(uint32_t)pc[e] * rowbytes;
                        if constexpr (VALUED) s_val[wave][g][l * E + e] = pv[e];
                    }
                    if (tbase + T < ge) {  // long segment: next tile of the same visit
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const int p = tbase + T + l * E + e;
                            if (p < ge) {
                                pc[e] = __builtin_nontemporal_load(a.colind + p);
                                if constexpr (VALUED) pv[e] = __builtin_nontemporal_load(a.val + p);
                            }
                        }
                    }
                    wave_lds_sync_r();
                }
                const int cnt = ge - k;
                const int t = k - tbase;
                off_t off[U];
                float v[U];
                float bv[U][V];
#pragma unroll
                for (int j = 0; j < U; ++j) {  // LDS reads are unconditional (clamped slot)
                    const int tj = t + ((j < cnt) ? j : cnt - 1);
                    off[j] = s_off[wave][g][tj];
                    if constexpr (VALUED) v[j] = s_val[wave][g][tj];
                    else v[j] = 1.0f;
                }
                if (cnt >= U) {
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const VT x = *reinterpret_cast<const VT*>(Bbase + (off_t)(off[j] + cbytes));
                        if constexpr (V == 1) bv[j][0] = x;
                        else
#pragma unroll
                            for (int q = 0; q < V; ++q) bv[j][q] = x[q];
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j)
#pragma unroll
                        for (int q = 0; q < V; ++q) acc[q] = combine_r<RED, VALUED>(acc[q], v[j], bv[j][q]);
                } else {  // last step of the visit: only the cnt live gathers are issued
#pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        if (j < cnt) {
                            const VT x = *reinterpret_cast<const VT*>(Bbase + (off_t)(off[j] + cbytes));
                            if constexpr (V == 1) bv[j][0] = x;
                            else
#pragma unroll
                                for (int q = 0; q < V; ++q) bv[j][q] = x[q];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        if (j < cnt) {
#pragma unroll
                            for (int q = 0; q < V; ++q) acc[q] = combine_r<RED, VALUED>(acc[q], v[j], bv[j][q]);
                        }
                    }
                }
                if (k + U >= tbase + T) {
                    wave_lds_sync_r();
                    tbase += T;
                }
            }
            wave_lds_sync_r();
            {
                VT v;
                if constexpr (V == 1) v = acc[0];
                else
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] = acc[k];
                s_acc[wave][g][i][l] = v;
            }
        }
    }

    if (colok) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i < nrows && !s_hub[wave][g][i]) {
                const VT v = s_acc[wave][g][i][l];
                __builtin_nontemporal_store(v, reinterpret_cast<VT*>(a.C + (size_t)(row0 + i) * (size_t)a.N + col0));
            }
        }
    }
}

template <int V, int W, bool VALUED, bool IDX64, int RED>
hipError_t run_panels(const SlabResArgs& base, int nslab, int slabs_per_launch, hipStream_t st) {
    constexpr int G = 64 / W;
    constexpr int rows_per_wg = kWaves * G * kSlabRowsPerGroup;
    auto kern = spmm_slabres_kernel<V, W, VALUED, IDX64, RED>;
    static int resident = 0;  // workgroups the device holds at once (per instantiation; one device type per process)
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kThreads, 0);
        if (e != hipSuccess) return e;
        if (per_cu < 1 || cus < 1) return hipErrorInvalidConfiguration;
        resident = per_cu * cus;
    }
    if (const char* s = getenv("GESPMM_SLABRES_WGS")) {  // experiments: workgroups per panel
        const int v = atoi(s);
        if (v > 0) resident = v;
    }
    SlabResArgs a = base;
    int64_t panel_wgs = resident / a.ntile;
    if (panel_wgs < 1) panel_wgs = 1;
    const int64_t cap_rows = panel_wgs * rows_per_wg;
    const int64_t npanel = ((int64_t)a.M + cap_rows - 1) / cap_rows;
    int64_t panel_rows = ((int64_t)a.M + npanel - 1) / npanel;
    panel_rows = (panel_rows + rows_per_wg - 1) / rows_per_wg * rows_per_wg;
    const int k = slabs_per_launch > 0 ? slabs_per_launch : nslab;
    for (int64_t lo = 0; lo < a.M; lo += panel_rows) {
        a.row_lo = (int32_t)lo;
        a.row_hi = (int32_t)((lo + panel_rows < a.M) ? lo + panel_rows : a.M);
        const int64_t nblk = ((int64_t)(a.row_hi - a.row_lo) + rows_per_wg - 1) / rows_per_wg;
        for (int s0 = 0; s0 < nslab; s0 += k) {
            a.s0 = s0;
            a.s1 = (s0 + k < nslab) ? s0 + k : nslab;
            hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * a.ntile)), dim3(kThreads), 0, st, a);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

template <int V, int W, bool IDX64>
hipError_t run_kind(const SlabResArgs& a, int reduce, int nslab, int k, hipStream_t st) {
    if (reduce == kReduceMax) return run_panels<V, W, false, IDX64, kReduceMax>(a, nslab, k, st);
    if (a.val) return run_panels<V, W, true, IDX64, kReduceSum>(a, nslab, k, st);
    return run_panels<V, W, false, IDX64, kReduceSum>(a, nslab, k, st);
}

template <int V, int W>
hipError_t run_idx(const SlabResArgs& a, const Geometry& geo, int nslab, int k, hipStream_t st) {
    return geo.idx64 ? run_kind<V, W, true>(a, geo.reduce, nslab, k, st) : run_kind<V, W, false>(a, geo.reduce, nslab, k, st);
}

}  // namespace

bool slabresident_serves(const Geometry& geo) {
    if (geo.strips != 1) return false;
    if (geo.vec == 4) return geo.group == 16 || geo.group == 32 || geo.group == 64;
    return (geo.vec == 1 || geo.vec == 2) && geo.group == 64;
}

hipError_t launch_spmm_slabresident(const SpmmArgs& a0, const Geometry& geo, void* ext_ws, size_t ext_bytes,
                                    hipStream_t st) {
    if (!slabresident_serves(geo)) return hipErrorNotSupported;
    if (geo.reduce == kReduceMax && a0.val != nullptr) return hipErrorInvalidValue;
    const int M = a0.M;
    const int nslab = (int)(((int64_t)geo.K + geo.slab_rows - 1) / geo.slab_rows);
    if (nslab < 1 || M <= 0) return hipErrorInvalidValue;
    int32_t* split = nullptr;
    const size_t bytes = (size_t)(nslab + 1) * (size_t)M * 4;
    hipError_t e = hipSuccess;
    const bool own = !(ext_ws && ext_bytes >= bytes && (reinterpret_cast<uintptr_t>(ext_ws) & 15) == 0);
    if (own) e = workspace_alloc(reinterpret_cast<void**>(&split), bytes, st);
    else split = static_cast<int32_t*>(ext_ws);
    if (e != hipSuccess) return e;
    if (!(!own && (a0.flags & kFlagReuseSplit)))
        e = launch_slabplan(a0.rowptr, a0.colind, split, M, nslab, geo.slab_rows, st);
    if (e == hipSuccess) {
        SlabResArgs a{};
        a.rowptr = a0.rowptr;
        a.colind = a0.colind;
        a.val = a0.val;
        a.B = a0.B;
        a.C = a0.C;
        a.split = split;
        a.M = M;
        a.N = a0.N;
        a.ntile = (a0.N + geo.group * geo.vec - 1) / (geo.group * geo.vec);
        a.empty = a0.empty;
        a.hub_thr = 0;
        if (const char* s = getenv("GESPMM_SLABRES_HUB")) a.hub_thr = atoi(s);  // experiments
        int k = 0;
        if (const char* s = getenv("GESPMM_SLABRES_K")) k = atoi(s);  // experiments: slabs per launch (0 = all)
        if (geo.vec == 4) {
            if (geo.group == 16) e = run_idx<4, 16>(a, geo, nslab, k, st);
            else if (geo.group == 32) e = run_idx<4, 32>(a, geo, nslab, k, st);
            else e = run_idx<4, 64>(a, geo, nslab, k, st);
        } else if (geo.vec == 2) {
            e = run_idx<2, 64>(a, geo, nslab, k, st);
        } else {
            e = run_idx<1, 64>(a, geo, nslab, k, st);
        }
    }
    const hipError_t ef = own ? workspace_free(split, st) : hipSuccess;
    return e != hipSuccess ? e : ef;
}

}  // namespace gespmm
