// spmm_stream.h — the two streaming kernels (batch-stream, segmented-stream) and their launch tables, as templates over
// PLANNED so that the plain instantiations (spmm_kernels.hip) and the plan-mode ones (spmm_stream_plan.hip) are separate
// kernels in separate translation units.
#pragma once
#include "spmm_device.h"

namespace gespmm {

// ----------------------------------------------------------------------------- streaming CRC (+CWM) kernel
//
// The production kernel for variants 1-4. One wavefront owns `rpw` CONSECUTIVE rows
// (rpw a multiple of G, <= 32), i.e. one contiguous CSR range:
//
//   * row pointers of all its rows: ONE coalesced load, parked in LDS;
//   * the CSR range streams through the wavefront's 64-entry LDS tile, one coalesced
//     load per tile, the next tile always prefetched in registers — independent of
//     where the row boundaries fall;
//   * rows are walked G at a time ("batch"); a batch consumes the part of its rows
//     that lies in the current tile, the tile advances when the batch reaches past
//     it, the batch advances when its rows are finished. Both decisions are
//     wave-uniform;
//   * inside a batch the non-zeros are gathered U at a time, and the tail (< U) is
//     issued as ONE predicated group, so a row of <= U non-zeros costs a single
//     memory round trip instead of one per leftover entry.
//
// Per row this removes two of the three dependent global round trips of a
// row-per-wave design (rowptr -> colind/val -> B) and keeps the accumulation order
// (ascending CSR position, one FMA per non-zero) untouched.

//
// PLANNED (plan.cpp): the wavefront's rows and CSR range come from a task table instead of blockIdx, the matrix is the plan's
// row-permuted copy and row i is written to C row perm[i]. A template parameter, not a runtime branch: the plain
// instantiation is the round-1 kernel instruction for instruction (a runtime `planned` flag, a second LDS array and a task
// loop around the body cost the plain call 6-22 %: com-Amazon-shaped N = 128 150 -> 160 us, N = 32 48 -> 55 us, RMAT-22
// N = 128 4.77 -> 5.83 ms; profiles/r02/plain_path_regression.log).

template <int V, int S, int W, bool VALUED, bool IDX64, int RED, int U, bool PLANNED>
__global__ __launch_bounds__(kThreads) void spmm_stream_kernel(SpmmArgs a) {
    constexpr int G = 64 / W;
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ off_t s_off[kWaves][kTile];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? kTile : 1];
    __shared__ int s_ptr[kWaves][kMaxRowsPerWave + 1];
    __shared__ int s_perm[PLANNED ? kWaves : 1][PLANNED ? kMaxRowsPerWave : 1];

    if (a.guard != nullptr && *a.guard != a.guard_want) return;  // (guarded launch: spmm_kernels.h, SpmmArgs::guard — wave-uniform)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;

    const int nitems = a.nblk * a.ntile;
    const int item = (a.flags & kFlagNoXcdRemap) ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, nitems);
    int tile = 0, rb = item;
    if (a.ntile > 1) {
        tile = item % a.ntile;
        rb = item / a.ntile;
    }
    int row_first, nrows, wb, we;  // wave-uniform
    int rp_plan = 0, pm_plan = 0;

    const int col0 = tile * (W * V * S) + l * V;
    bool colok[S];
    off_t cbytes[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        colok[s] = (col0 + s * W * V) < a.N;
        // lanes/strips past N gather column 0 (valid, same line as lane 0) and skip the store
        cbytes[s] = colok[s] ? (off_t)(col0 + s * W * V) * 4u : (off_t)0;
    }
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const off_t rowbytes = (off_t)a.N * 4u;
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;

    // Tile stream state: `t0` = CSR position of the tile resident in LDS.
    int pc = 0;
    float pv = 0.0f;
    auto fetch_tile_regs = [&](int base) {
        const int p = base + lane;
        if (p < we) {
            pc = load_csr(a.colind + p);
            if constexpr (VALUED) pv = load_csr(a.val + p);
        }
    };
    auto publish_tile = [&]() {
        s_off[wave][lane] = (off_t)(uint32_t)pc * rowbytes;
        if constexpr (VALUED) s_val[wave][lane] = pv;
    };

    if constexpr (PLANNED) {
        // Plan mode: the task table names the rows and the CSR range, so the first CSR tile, the row
        // pointers and the C-row indices are three independent loads instead of a dependent chain.
        const int task_id = rb * kWaves + wave;
        if (task_id >= a.ntasks) return;
        const int4 t = reinterpret_cast<const int4*>(a.tasks)[task_id];
        row_first = __builtin_amdgcn_readfirstlane(t.x);
        nrows = __builtin_amdgcn_readfirstlane(t.y);
        wb = __builtin_amdgcn_readfirstlane(t.z);
        we = __builtin_amdgcn_readfirstlane(t.w);
        rp_plan = a.rowptr[row_first + (lane <= nrows ? lane : nrows)];
        pm_plan = a.perm[row_first + (lane < nrows ? lane : nrows - 1)];
    } else {
        const int rpw = a.rpw;
        row_first = (rb * kWaves + wave) * rpw;
        if (row_first >= a.M) return;  // whole wavefront leaves together
        nrows = (a.M - row_first < rpw) ? a.M - row_first : rpw;
        // Row pointers of this wavefront's rows -> LDS (one coalesced load, rpw <= 32).
        const int rp = a.rowptr[row_first + (lane <= nrows ? lane : nrows)];
        if (lane <= kMaxRowsPerWave) s_ptr[wave][lane] = rp;
        wb = __builtin_amdgcn_readfirstlane(rp);
        we = __builtin_amdgcn_readlane(rp, nrows);
    }
    fetch_tile_regs(wb);
    {
        int t0 = wb;
        if constexpr (PLANNED) {  // (after the tile loads are on their way: the three loads of a planned task overlap)
            if (lane <= kMaxRowsPerWave) s_ptr[wave][lane] = rp_plan;
            if (lane < kMaxRowsPerWave) s_perm[wave][lane] = pm_plan;
        }
        publish_tile();
        fetch_tile_regs(t0 + kTile);
        wave_lds_sync();

        for (int b = 0; b < nrows; b += G) {
            const int r = b + g;
            const bool rowok = r < nrows;
            int lb = 0, hb = 0;
            bool rowok2 = rowok;
            if (rowok) {
                lb = s_ptr[wave][r];
                hb = s_ptr[wave][r + 1];
                if (a.long_row > 0 && hb - lb > a.long_row) {  // left to the long-row pass
                    if (l == 0 && tile == 0 && a.lr_hdr) {  // one lane registers the row: chunk slots + list entry
                        const int nch = (hb - lb + a.lr_chunk - 1) / a.lr_chunk;
                        const int base = atomicAdd(a.lr_hdr + 0, nch);
                        const int j = atomicAdd(a.lr_hdr + 1, 1);
                        if (j < a.lr_max_rows && base + nch <= a.lr_max_chunks) {
                            reinterpret_cast<int4*>(a.lr_rows)[j] = make_int4(row_first + r, base, nch, 0);
                            for (int c = 0; c < nch; ++c)
                                reinterpret_cast<int2*>(a.lr_chunks)[base + c] = make_int2(row_first + r, c);
                        }
                    }
                    hb = lb;
                    rowok2 = false;
                }
            }
            const int be = __builtin_amdgcn_readfirstlane(s_ptr[wave][(b + G < nrows) ? b + G : nrows]);
            if constexpr (G == 1) {
                lb = __builtin_amdgcn_readfirstlane(lb);
                hb = __builtin_amdgcn_readfirstlane(hb);
            }

            float acc[S][V];
    #pragma unroll
            for (int s = 0; s < S; ++s)
    #pragma unroll
                for (int i = 0; i < V; ++i) acc[s][i] = init;

            for (;;) {
                const int tend = t0 + kTile;
                int k = (lb > t0 ? lb : t0) - t0;
                const int ke = (hb < tend ? hb : tend) - t0;
                // Full steps: U gathers issued back to back, no predicates.
                for (; k + U <= ke; k += U) {
                    off_t off[U];
                    float v[U];
                    float bv[U][S][V];
    #pragma unroll
                    for (int j = 0; j < U; ++j) {
                        off[j] = s_off[wave][k + j];
                        if constexpr (VALUED) v[j] = s_val[wave][k + j];
                        else v[j] = 1.0f;
                    }
    #pragma unroll
                    for (int j = 0; j < U; ++j)
    #pragma unroll
                        for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
    #pragma unroll
                    for (int j = 0; j < U; ++j)
    #pragma unroll
                        for (int s = 0; s < S; ++s)
    #pragma unroll
                            for (int i = 0; i < V; ++i) acc[s][i] = combine<RED, VALUED>(acc[s][i], v[j], bv[j][s][i]);
                }
                // Tail (1..U-1 entries): ONE predicated group, so a short row is a single round
                // trip. It is not inside a loop, so there is no loop-carried register hazard and
                // the compiler keeps the predicated loads in flight together.
                const int rem = ke - k;
                if (rem > 0) {
                    off_t off[U - 1];
                    float v[U - 1];
                    float bv[U - 1][S][V];
                    // LDS reads first, all of them (clamped slot: always inside the tile), THEN the
                    // predicated gathers: with the read inside the predicate every gather waited for
                    // its own LDS round trip (tail of r entries cost r serial LDS latencies).
    #pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        const int kj = k + ((j < rem) ? j : rem - 1);
                        off[j] = s_off[wave][kj];
                        if constexpr (VALUED) v[j] = s_val[wave][kj];
                        else v[j] = 1.0f;
                    }
    #pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        if (j < rem) {
    #pragma unroll
                            for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
                        }
                    }
    #pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        if (j < rem) {
    #pragma unroll
                            for (int s = 0; s < S; ++s)
    #pragma unroll
                                for (int i = 0; i < V; ++i)
                                    acc[s][i] = combine<RED, VALUED>(acc[s][i], v[j], bv[j][s][i]);
                        }
                    }
                }
                if (be <= tend) break;  // every row of this batch ends inside the resident tile
                wave_lds_sync();        // all reads of the old tile are issued before it is overwritten
                t0 = tend;
                publish_tile();
                fetch_tile_regs(t0 + kTile);
                wave_lds_sync();
            }

            if (rowok2) {
                int crow = row_first + r;
                if constexpr (PLANNED) crow = s_perm[wave][r];
                float* Crow = a.C + (size_t)crow * (size_t)a.N + col0;
                const bool nts = (a.flags & kFlagNtStore) != 0;
                const bool sc1 = (a.flags & kFlagSc1Store) != 0;
    #pragma unroll
                for (int s = 0; s < S; ++s)
                    if (colok[s]) {
                        if (sc1) store_vec_sc1<V>(Crow + s * (W * V), acc[s]);
                        else if (nts) store_vec<V, true>(Crow + s * (W * V), acc[s]);
                        else store_vec<V, false>(Crow + s * (W * V), acc[s]);
                    }
            }
        }
    }
}


// ----------------------------------------------------------------------------- segmented-stream kernel
//
// Every W-lane group owns `rpg` CONSECUTIVE rows, i.e. one contiguous CSR range
// [gb, ge), and treats it as ONE stream: U entries per step are gathered back to
// back no matter where the row boundaries fall, then consumed in CSR order; whenever
// the position passes the end of the current row the accumulator is flushed to C
// (also for empty rows) and restarted. The per-element sum is still one fp32 chain
// in ascending CSR position, so results are bit-identical to the other variants —
// but a graph of 5-nnz rows keeps 8 B-row loads in flight per group continuously
// instead of draining the memory pipeline at every row batch.
//
// CRC staging is per group: the group's lanes load T = max(W, 32) entries per refill
// (E = T/W consecutive entries per lane), next tile prefetched in registers.

template <int V, int S, int W, bool VALUED, bool IDX64, int RED, int U, bool PLANNED>
__global__ __launch_bounds__(kThreads) void spmm_segstream_kernel(SpmmArgs a) {
    constexpr int G = 64 / W;
    constexpr int T = (W > 32) ? W : 32;  // entries per group tile
    constexpr int E = T / W;              // entries each lane stages per refill
    static_assert(T % U == 0, "tile must hold whole steps");
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ off_t s_off[kWaves][G][T];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? G : 1][VALUED ? T : 1];
    __shared__ int s_ptr[kWaves][G][kMaxRowsPerWave + 1];
    __shared__ int s_perm[PLANNED ? kWaves : 1][PLANNED ? G : 1][PLANNED ? kMaxRowsPerWave : 1];

    if (a.guard != nullptr && *a.guard != a.guard_want) return;  // (guarded launch: spmm_kernels.h, SpmmArgs::guard — wave-uniform)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;

    const int nitems = a.nblk * a.ntile;
    const int item = (a.flags & kFlagNoXcdRemap) ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, nitems);
    int tile = 0, rb = item;
    if (a.ntile > 1) {
        tile = item % a.ntile;
        rb = item / a.ntile;
    }
    int task_first, nrows, gb = 0, ge = 0;
    if constexpr (PLANNED) {
        // Plan mode: lane group q works on gtasks[q] = {first permuted row, #rows, CSR begin, CSR end}; row i of the
        // permuted matrix is written to C row perm[i].
        const int q0 = (rb * kWaves + wave) * G;
        if (q0 >= a.ngtasks) return;  // whole wavefront past the end
        int4 t = make_int4(0, 0, 0, 0);
        if (q0 + g < a.ngtasks) t = reinterpret_cast<const int4*>(a.gtasks)[q0 + g];
        task_first = t.x;
        nrows = t.y;
        gb = t.z;
        ge = t.w;  // (row pointers and C rows are loaded after the first CSR tile is on its way, below)
    } else {
        const int rpg = a.rpw;  // rows per GROUP in this kernel
        task_first = ((rb * kWaves + wave) * G + g) * rpg;
        if (((rb * kWaves + wave) * G) * rpg >= a.M) return;  // whole wavefront past the end
        nrows = a.M - task_first;                              // rows of this group's task
        nrows = nrows < 0 ? 0 : (nrows > rpg ? rpg : nrows);
        // Row pointers of the task -> LDS (rpg <= 32; lanes of a group cover 0..rpg by striding W).
        if (nrows > 0) {
            for (int i = l; i <= nrows; i += W) s_ptr[wave][g][i] = a.rowptr[task_first + i];
        }
        wave_lds_sync();
        if (nrows > 0) {
            gb = s_ptr[wave][g][0];
            ge = s_ptr[wave][g][nrows];
        }
    }

    const int col0 = tile * (W * V * S) + l * V;
    bool colok[S];
    off_t cbytes[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        colok[s] = (col0 + s * W * V) < a.N;
        cbytes[s] = colok[s] ? (off_t)(col0 + s * W * V) * 4u : (off_t)0;
    }
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const off_t rowbytes = (off_t)a.N * 4u;
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;
    const bool nts = (a.flags & kFlagNtStore) != 0;
    const bool sc1 = (a.flags & kFlagSc1Store) != 0;

    // Per-group tile stream.
    int pc[E];
    float pv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        pc[e] = 0;
        pv[e] = 0.0f;
    }
    auto fetch_tile_regs = [&](int base) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int p = base + l * E + e;
            if (p < ge) {
                pc[e] = load_csr(a.colind + p);
                if constexpr (VALUED) pv[e] = load_csr(a.val + p);
            }
        }
    };
    auto publish_tile = [&]() {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            s_off[wave][g][l * E + e] = (off_t)(uint32_t)pc[e] * rowbytes;
            if constexpr (VALUED) s_val[wave][g][l * E + e] = pv[e];
        }
    };

    int tbase = gb;  // CSR position of the group's resident tile
    fetch_tile_regs(tbase);
    if constexpr (PLANNED) {
        if (nrows > 0) {
            for (int i = l; i <= nrows; i += W) s_ptr[wave][g][i] = a.rowptr[task_first + i];
            for (int i = l; i < nrows; i += W) s_perm[wave][g][i] = a.perm[task_first + i];
        }
    }
    publish_tile();
    fetch_tile_regs(tbase + T);
    wave_lds_sync();

    float acc[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[s][i] = init;
    int cur = 0;                                       // current row of the task
    int rend = (nrows > 0) ? s_ptr[wave][g][1] : 0;    // CSR end of the current row
    auto flush_row = [&]() {
        int crow = task_first + cur;
        if constexpr (PLANNED) crow = s_perm[wave][g][cur];
        float* Crow = a.C + (size_t)crow * (size_t)a.N + col0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (colok[s]) {
                if (sc1) store_vec_sc1<V>(Crow + s * (W * V), acc[s]);
                else if (nts) store_vec<V, true>(Crow + s * (W * V), acc[s]);
                else store_vec<V, false>(Crow + s * (W * V), acc[s]);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) acc[s][i] = init;
        }
        ++cur;
        rend = s_ptr[wave][g][(cur + 1 <= nrows) ? cur + 1 : nrows];
    };

    for (int k = gb; k < ge; k += U) {
        if (k >= tbase + T) {  // group-uniform: the step crosses into the next tile
            tbase += T;
            publish_tile();
            fetch_tile_regs(tbase + T);
            wave_lds_sync();
        }
        const int cnt = ge - k;
        const int t = k - tbase;  // a multiple of U: tiles hold whole steps
        off_t off[U];
        float v[U];
        float bv[U][S][V];
        if (cnt >= U) {
            // Full step: the U tile slots are read with constant offsets (no per-slot clamping: ~4 instead of ~9 VALU
            // instructions per gather, which is what the kernel is short of when the B rows come from L2) ...
#pragma unroll
            for (int j = 0; j < U; ++j) {
                off[j] = s_off[wave][g][t + j];
                if constexpr (VALUED) v[j] = s_val[wave][g][t + j];
                else v[j] = 1.0f;
            }
#pragma unroll
            for (int j = 0; j < U; ++j)
#pragma unroll
                for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
            if (k + U <= rend) {
                // ... and every entry belongs to the current row (rend = its CSR end and the previous entry already
                // did): long rows run without boundary checks
#pragma unroll
                for (int j = 0; j < U; ++j)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[s][i] = combine<RED, VALUED>(acc[s][i], v[j], bv[j][s][i]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    while (k + j >= rend) flush_row();  // rows ending before this entry (incl. empty ones)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[s][i] = combine<RED, VALUED>(acc[s][i], v[j], bv[j][s][i]);
                }
            }
        } else {
            // Last step of the stream: straight-line issue of all U gathers, slots past the end re-read the last valid
            // entry (same cache line, no extra memory traffic) instead of being branched around — control flow around
            // the individual loads makes the compiler serialise them with s_waitcnt vmcnt(0). Predicates apply only when
            // the values are consumed.
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int tj = t + ((j < cnt) ? j : cnt - 1);
                off[j] = s_off[wave][g][tj];
                if constexpr (VALUED) v[j] = s_val[wave][g][tj];
                else v[j] = 1.0f;
#pragma unroll
                for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (j < cnt) {
                    while (k + j >= rend) flush_row();
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[s][i] = combine<RED, VALUED>(acc[s][i], v[j], bv[j][s][i]);
                }
            }
        }
        wave_lds_sync();  // reads of this step precede a possible tile publish of the next step
    }
    while (cur < nrows) flush_row();  // last row and any trailing empty rows
}


// ----------------------------------------------------------------------------- launch tables

template <int V, int S, int W, bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t launch_stream(const SpmmArgs& a, int rpw, hipStream_t st) {
    constexpr int G = 64 / W;
    SpmmArgs args = a;
    if (rpw < G) rpw = G;
    if (rpw > kMaxRowsPerWave) rpw = kMaxRowsPerWave;
    rpw = rpw / G * G;
    args.ntile = (a.N + W * V * S - 1) / (W * V * S);
    // HIP caps a launch at 2^32 threads (gridDim.x * blockDim.x): with 256-thread workgroups that is
    // kMaxGridBlocks workgroups. Tasks grow until the grid fits (M = 2^26 rows: >= 2 rows per task).
    while (rpw < kMaxRowsPerWave &&
           (((int64_t)a.M + kWaves * rpw - 1) / (kWaves * rpw)) * args.ntile > kMaxGridBlocks) {
        rpw *= 2;  // (never past the kernel's row-pointer staging, whatever the caller's rows_per_wave was)
        if (rpw > kMaxRowsPerWave) rpw = kMaxRowsPerWave;
        rpw = rpw / G * G;
    }
    args.rpw = rpw;
    args.nblk = (int)(((int64_t)a.M + kWaves * rpw - 1) / (kWaves * rpw));
    if constexpr (PLANNED) args.nblk = (a.ntasks + kWaves - 1) / kWaves;  // plan mode: one wavefront per task
    const int64_t nitems = (int64_t)args.nblk * args.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    // Gather depth U: 8 B-row loads in flight per lane group unless the accumulators are
    // already wide (CF = 8) or the caller asks for the shallow form.
    if constexpr (V * S >= 8) {
        hipLaunchKernelGGL((spmm_stream_kernel<V, S, W, VALUED, IDX64, RED, 4, PLANNED>), dim3((unsigned)nitems),
                           dim3(kThreads), 0, st, args);
    } else {
        if (a.flags & kFlagShallowUnroll)
            hipLaunchKernelGGL((spmm_stream_kernel<V, S, W, VALUED, IDX64, RED, 4, PLANNED>), dim3((unsigned)nitems),
                               dim3(kThreads), 0, st, args);
        else
            hipLaunchKernelGGL((spmm_stream_kernel<V, S, W, VALUED, IDX64, RED, 8, PLANNED>), dim3((unsigned)nitems),
                               dim3(kThreads), 0, st, args);
    }
    return hipGetLastError();
}

template <int V, int S, bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t stream_w(const SpmmArgs& a, int W, int rpw, hipStream_t st) {
    switch (W) {
        case 4: return launch_stream<V, S, 4, VALUED, IDX64, RED, PLANNED>(a, rpw, st);
        case 8: return launch_stream<V, S, 8, VALUED, IDX64, RED, PLANNED>(a, rpw, st);
        case 16: return launch_stream<V, S, 16, VALUED, IDX64, RED, PLANNED>(a, rpw, st);
        case 32: return launch_stream<V, S, 32, VALUED, IDX64, RED, PLANNED>(a, rpw, st);
        case 64: return launch_stream<V, S, 64, VALUED, IDX64, RED, PLANNED>(a, rpw, st);
    }
    return hipErrorInvalidValue;
}

template <bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t stream_vs(const SpmmArgs& a, const Geometry& g, hipStream_t st) {
    if (g.strips == 2) {
        if (g.vec == 4) return stream_w<4, 2, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_wave, st);
        // widths that allow no dwordx4 (odd N, N = 2 mod 4) beyond one 64-lane tile: two strips, W = 64 only
        if (g.group != 64) return hipErrorInvalidValue;
        if (g.vec == 2) return launch_stream<2, 2, 64, VALUED, IDX64, RED, PLANNED>(a, g.rows_per_wave, st);
        return launch_stream<1, 2, 64, VALUED, IDX64, RED, PLANNED>(a, g.rows_per_wave, st);
    }
    switch (g.vec) {
        case 1: return stream_w<1, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_wave, st);
        case 2: return stream_w<2, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_wave, st);
        case 4: return stream_w<4, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_wave, st);
    }
    return hipErrorInvalidValue;
}

template <bool PLANNED>
static hipError_t launch_spmm_stream_impl(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    const bool valued = a.val != nullptr;
    if (geo.reduce == kReduceMax) {
        if (valued) return hipErrorInvalidValue;  // max reducer is unweighted (binary_reduce_max.cu)
        return geo.idx64 ? stream_vs<false, true, kReduceMax, PLANNED>(a, geo, st) : stream_vs<false, false, kReduceMax, PLANNED>(a, geo, st);
    }
    if (valued) return geo.idx64 ? stream_vs<true, true, kReduceSum, PLANNED>(a, geo, st) : stream_vs<true, false, kReduceSum, PLANNED>(a, geo, st);
    return geo.idx64 ? stream_vs<false, true, kReduceSum, PLANNED>(a, geo, st) : stream_vs<false, false, kReduceSum, PLANNED>(a, geo, st);
}

template <int V, int S, int W, bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t launch_segstream(const SpmmArgs& a, int rpg, hipStream_t st) {
    constexpr int G = 64 / W;
    SpmmArgs args = a;
    if (rpg < 1) rpg = 1;
    if (rpg > kMaxRowsPerWave) rpg = kMaxRowsPerWave;
    args.rpw = rpg;
    args.nblk = (int)(((int64_t)a.M + (int64_t)kWaves * G * rpg - 1) / ((int64_t)kWaves * G * rpg));
    if constexpr (PLANNED) args.nblk = (a.ngtasks + kWaves * G - 1) / (kWaves * G);  // plan mode: one lane group per task
    args.ntile = (a.N + W * V * S - 1) / (W * V * S);
    const int64_t nitems = (int64_t)args.nblk * args.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    if constexpr (V * S >= 8) {
        hipLaunchKernelGGL((spmm_segstream_kernel<V, S, W, VALUED, IDX64, RED, 4, PLANNED>), dim3((unsigned)nitems),
                           dim3(kThreads), 0, st, args);
    } else {
        if (a.flags & kFlagShallowUnroll)
            hipLaunchKernelGGL((spmm_segstream_kernel<V, S, W, VALUED, IDX64, RED, 4, PLANNED>), dim3((unsigned)nitems),
                               dim3(kThreads), 0, st, args);
        else
            hipLaunchKernelGGL((spmm_segstream_kernel<V, S, W, VALUED, IDX64, RED, 8, PLANNED>), dim3((unsigned)nitems),
                               dim3(kThreads), 0, st, args);
    }
    return hipGetLastError();
}

template <int V, int S, bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t segstream_w(const SpmmArgs& a, int W, int rpg, hipStream_t st) {
    switch (W) {
        case 4: return launch_segstream<V, S, 4, VALUED, IDX64, RED, PLANNED>(a, rpg, st);
        case 8: return launch_segstream<V, S, 8, VALUED, IDX64, RED, PLANNED>(a, rpg, st);
        case 16: return launch_segstream<V, S, 16, VALUED, IDX64, RED, PLANNED>(a, rpg, st);
        case 32: return launch_segstream<V, S, 32, VALUED, IDX64, RED, PLANNED>(a, rpg, st);
        case 64: return launch_segstream<V, S, 64, VALUED, IDX64, RED, PLANNED>(a, rpg, st);
    }
    return hipErrorInvalidValue;
}

template <bool VALUED, bool IDX64, int RED, bool PLANNED>
static hipError_t segstream_vs(const SpmmArgs& a, const Geometry& g, hipStream_t st) {
    if (g.strips == 2) {
        if (g.vec == 4) return segstream_w<4, 2, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_group, st);
        return hipErrorInvalidValue;
    }
    switch (g.vec) {
        case 1: return segstream_w<1, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_group, st);
        case 2: return segstream_w<2, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_group, st);
        case 4: return segstream_w<4, 1, VALUED, IDX64, RED, PLANNED>(a, g.group, g.rows_per_group, st);
    }
    return hipErrorInvalidValue;
}

template <bool PLANNED>
static hipError_t launch_spmm_segstream_impl(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    const bool valued = a.val != nullptr;
    if (geo.reduce == kReduceMax) {
        if (valued) return hipErrorInvalidValue;
        return geo.idx64 ? segstream_vs<false, true, kReduceMax, PLANNED>(a, geo, st)
                         : segstream_vs<false, false, kReduceMax, PLANNED>(a, geo, st);
    }
    if (valued)
        return geo.idx64 ? segstream_vs<true, true, kReduceSum, PLANNED>(a, geo, st)
                         : segstream_vs<true, false, kReduceSum, PLANNED>(a, geo, st);
    return geo.idx64 ? segstream_vs<false, true, kReduceSum, PLANNED>(a, geo, st)
                     : segstream_vs<false, false, kReduceSum, PLANNED>(a, geo, st);
}

}  // namespace gespmm
