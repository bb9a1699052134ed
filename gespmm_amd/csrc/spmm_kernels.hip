// spmm_kernels.hip — CSR x dense row-product SpMM for MI355X (gfx950, wave64).
//
// What is computed (reference semantics, spmm_test.cu:64-236 / spmm_kernel.cu:31-379):
//     C[r, c] = sum_{p = rowptr[r]}^{rowptr[r+1]-1} val[p] * B[colind[p], c]
// accumulated in ONE fp32 register per output element, ascending p, one fused
// multiply-add per non-zero (what nvcc emits for `acc += a*b`), rows without
// non-zeros write 0, C fully overwritten.
//
// How it is laid out on CDNA4 (a re-design, not a translation of the reference's
// 32-lane kernels):
//
//  * Row groups. A 64-lane wavefront is split into G = 64/W groups of W lanes; a
//    group works on one row at a time, each lane owns S strips of V CONTIGUOUS output
//    columns (one dwordx{V} load / store per strip). The reference's "coarse-grained
//    warp merging" (CWM, lane owns columns c, c+32, ...) becomes CF = V*S with
//    contiguous vectors, so a group reads/writes W*V*4 contiguous bytes of a B/C row
//    per instruction (N=128: W=32, V=4 -> 512 B per half-wave).
//
//  * Coalesced Row Caching (CRC) in LDS. Consecutive rows are ONE contiguous CSR
//    range, streamed through wavefront-private LDS tiles with coalesced loads (column
//    index pre-scaled to a byte offset into B, as the reference pre-multiplies by N
//    at spmm_test.cu:124); the next tile is always prefetched in registers.
//
//  * Memory-level parallelism. U (= 8) LDS reads, U independent B-row gathers, then
//    U FMAs in CSR order — the order of the additions never changes, so all variants
//    produce the same bits. Gathers are never issued under divergent branches inside
//    a loop (hipcc then serialises them with s_waitcnt vmcnt(0)).
//
//  * XCD-aware mapping. The hardware deals workgroup b to XCD b % 8; ids are remapped
//    so each XCD (private 4 MiB L2) sweeps one contiguous eighth of the rows and the
//    column tiles of a row block run back-to-back on the same XCD.
//
// Kernels in this file (selection: select.cpp, measurements: DESIGN.md §3.2):
//   spmm_naive_kernel      variant 0: no LDS staging
//   spmm_stream_kernel     batch-stream (default): rows walked G at a time over a wave-wide tile,
//                          small tasks (~12 KB of gathered B per wavefront)
//   spmm_segstream_kernel  segmented-stream: one continuous gather stream per lane group
//                          (short rows with B resident in L2; otherwise opt-in)
//   spmm_longrow_{chunk,combine}_kernel   hub rows of skewed graphs in 2048-entry chunks
//                          spread over the chip, ordered combine (deterministic)
//   spmm_slabplan_kernel / spmm_slab_kernel   cache blocking for dense graphs
//   spmm_parreduce_kernel  variant 5: lanes over nnz, reduce-scatter + xor butterfly
//
// No MFMA: the inner product is a gather, not a dense contraction.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "spmm_kernels.h"
#include "spmm_stream.h"
#include "workspace.h"

namespace gespmm {

// ----------------------------------------------------------------------------- naive kernel (variant 0)
//
// The reference's method 0 (spmm_test.cu:64-95) on the wave64 geometry: one row per lane
// group, every lane reads colind/val of its row itself (lanes of a group hit the same
// address, one broadcast transaction), no LDS staging. Kept as the baseline variant.

template <int V, int S, int W, bool VALUED, bool IDX64>
__global__ __launch_bounds__(kThreads) void spmm_naive_kernel(SpmmArgs a) {
    constexpr int G = 64 / W;
    constexpr int U = (V * S >= 8) ? 2 : ((V * S >= 4) ? 4 : 8);
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const int nitems = a.nblk * a.ntile;
    const int item = (a.flags & kFlagNoXcdRemap) ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, nitems);
    const int tile = item % a.ntile;
    const int rb = item / a.ntile;
    const int row0 = (rb * kWaves + wave) * G;
    if (row0 >= a.M) return;
    const int row = row0 + g;
    const int col0 = tile * (W * V * S) + l * V;
    const bool rowok = row < a.M;
    int lb = 0, hb = 0;
    if (rowok) {
        lb = a.rowptr[row];
        hb = a.rowptr[row + 1];
    }
    bool colok[S];
    off_t cbytes[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        colok[s] = (col0 + s * W * V) < a.N;
        cbytes[s] = colok[s] ? (off_t)(col0 + s * W * V) * 4u : (off_t)0;
    }
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const off_t rowbytes = (off_t)a.N * 4u;
    float acc[S][V];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[s][i] = 0.0f;

    int k = lb;
    for (; k + U <= hb; k += U) {
        off_t off[U];
        float v[U];
        float b[U][S][V];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            off[j] = (off_t)(uint32_t)a.colind[k + j] * rowbytes;
            v[j] = VALUED ? a.val[k + j] : 1.0f;
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s) load_vec<V>(b[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int i = 0; i < V; ++i) acc[s][i] = combine<kReduceSum, VALUED>(acc[s][i], v[j], b[j][s][i]);
    }
    for (; k < hb; ++k) {
        const off_t off = (off_t)(uint32_t)a.colind[k] * rowbytes;
        const float v = VALUED ? a.val[k] : 1.0f;
        float b[S][V];
#pragma unroll
        for (int s = 0; s < S; ++s) load_vec<V>(b[s], Bbase + (off_t)(off + cbytes[s]));
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[s][i] = combine<kReduceSum, VALUED>(acc[s][i], v, b[s][i]);
    }
    if (rowok) {
        float* Crow = a.C + (size_t)row * (size_t)a.N + col0;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (colok[s]) store_vec<V, false>(Crow + s * (W * V), acc[s]);
    }
}

// (the batch-stream and segmented-stream kernels: spmm_stream.h)

// ----------------------------------------------------------------------------- long-row pass
//
// Load balance for skewed graphs (RMAT hubs): a row of 10^4..10^6 non-zeros walked by ONE
// lane group is a serial chain that outlasts the rest of the launch, and even a whole
// workgroup per row is too little for the biggest hubs (a workgroup keeps ~32 KB of
// gathers in flight: tens of GB/s). Rows longer than `long_row` entries are therefore
// skipped by the main kernel and done here in CHUNKS of kLongRowChunk entries:
//
//   1. (in the main kernel)       the lane group that meets a long row skips it and registers it:
//                                 ceil(len/chunk) consecutive chunk slots (one atomic counter) and one
//                                 entry of the long-row list {row, first slot, #chunks};
//   2. spmm_longrow_chunk_kernel  workgroups walk the chunk list grid-stride; the NG = 4*G
//                                 lane groups of a workgroup take the chunk's 64-entry tiles
//                                 round-robin (group q: tiles q, q+NG, ...), each keeps ONE
//                                 accumulator chain over its tiles in ascending order, the NG
//                                 partial rows are added in fixed order q = 0..NG-1 through
//                                 LDS and written to partial[slot][0..N);
//   3. spmm_longrow_combine_kernel  per long row: C[row] = partial[first] + partial[first+1]
//                                 + ... in chunk order.
//
// Which slot a row gets depends on scheduling, its value does not: the result is
// bit-reproducible run to run, but it is a re-association of the strict CSR-order sum, so
// these rows are checked to north_star's 1e-4 tolerance instead of bit-for-bit.
// GESPMM_FLAG_STRICT_ORDER turns the split off. RMAT-20, N=128: the previous form (one
// workgroup per 4096-row slice serving the long rows it finds) took 1.11 ms of a 2.0 ms
// launch pair because the hubs cluster in a few slices.

struct LongRowHeader {
    int nchunks;
    int nrows;
    int pad[2];
};

// The header is zeroed by a KERNEL, not by hipMemsetAsync: captured into a HIP graph (a caller's workspace under torch.cuda.graph) the
// memset node did not reliably precede the kernels behind it on this runtime — about every second process replayed its graph with the
// workspace's stale header, and the combine kernel stored the (empty) partial sums of "row 0" over a finished row
// (profiles/r06/capture_flake_probe.log: row 0 of the second product zero in every replay of such a process, never without the
// long-row pass). A kernel node keeps its place in the stream's order, captured or not.
__global__ void spmm_longrow_reset_kernel(LongRowHeader* hdr) {
    if (threadIdx.x == 0) {
        hdr->nchunks = 0;
        hdr->nrows = 0;
        hdr->pad[0] = hdr->pad[1] = 0;
    }
}

template <int V, int S, int W, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_longrow_chunk_kernel(SpmmArgs a, int chunk, int max_chunks,
                                                                       const LongRowHeader* __restrict__ hdr,
                                                                       const int2* __restrict__ chunklist,
                                                                       float* __restrict__ partial) {
    constexpr int G = 64 / W;
    constexpr int NG = kWaves * G;
    constexpr int T = 64;  // entries per tile
    constexpr int E = T / W;
    constexpr int U = (V * S >= 8) ? 4 : 8;
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ off_t s_off[kWaves][G][T];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? G : 1][VALUED ? T : 1];
    __shared__ float s_part[NG][W * V * S];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const int q = wave * G + g;  // group id inside the workgroup

    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const off_t rowbytes = (off_t)a.N * 4u;
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;
    const int ntile = (a.N + W * V * S - 1) / (W * V * S);
    int nchunks = hdr->nchunks;
    if (nchunks > max_chunks) nchunks = max_chunks;

    for (int slot = blockIdx.x; slot < nchunks; slot += gridDim.x) {
        const int2 job = chunklist[slot];
        const int lb = a.rowptr[job.x] + job.y * chunk;
        const int rend = a.rowptr[job.x + 1];
        const int hb = (lb + chunk < rend) ? lb + chunk : rend;
        const int ntiles_row = (hb - lb + T - 1) / T;
        float* dst = partial + (size_t)slot * (size_t)a.N;
        for (int ct = 0; ct < ntile; ++ct) {
            const int col0 = ct * (W * V * S) + l * V;
            bool colok[S];
            off_t cbytes[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                colok[s] = (col0 + s * W * V) < a.N;
                cbytes[s] = colok[s] ? (off_t)(col0 + s * W * V) * 4u : (off_t)0;
            }
            float acc[S][V];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int k2 = 0; k2 < V; ++k2) acc[s][k2] = init;

            // group q walks tiles q, q+NG, ... ; the W lanes of the group stage T/W entries each
            int pc[E];
            float pv[E];
            auto fetch = [&](int t) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = lb + t * T + l * E + e;
                    pc[e] = 0;
                    pv[e] = 0.0f;
                    if (t < ntiles_row && p < hb) {
                        pc[e] = load_csr(a.colind + p);
                        if constexpr (VALUED) pv[e] = load_csr(a.val + p);
                    }
                }
            };
            fetch(q);
            for (int t = q; t < ntiles_row; t += NG) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    s_off[wave][g][l * E + e] = (off_t)(uint32_t)pc[e] * rowbytes;
                    if constexpr (VALUED) s_val[wave][g][l * E + e] = pv[e];
                }
                fetch(t + NG);
                wave_lds_sync();
                const int cnt_tile = (hb - (lb + t * T) < T) ? hb - (lb + t * T) : T;
                for (int k = 0; k < cnt_tile; k += U) {
                    const int cnt = cnt_tile - k;
                    off_t off[U];
                    float v[U];
                    float bv[U][S][V];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int kj = k + ((j < cnt) ? j : cnt - 1);
                        off[j] = s_off[wave][g][kj];
                        if constexpr (VALUED) v[j] = s_val[wave][g][kj];
                        else v[j] = 1.0f;
#pragma unroll
                        for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        if (j < cnt) {
#pragma unroll
                            for (int s = 0; s < S; ++s)
#pragma unroll
                                for (int k2 = 0; k2 < V; ++k2)
                                    acc[s][k2] = combine<RED, VALUED>(acc[s][k2], v[j], bv[j][s][k2]);
                        }
                    }
                }
                wave_lds_sync();
            }
            // ---- fixed-order combine of the NG partial rows
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int k2 = 0; k2 < V; ++k2) s_part[q][(s * W + l) * V + k2] = acc[s][k2];
            __syncthreads();
            if (q == 0) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int k2 = 0; k2 < V; ++k2) {
                        float t2 = s_part[0][(s * W + l) * V + k2];
                        for (int qq = 1; qq < NG; ++qq) {
                            const float pq = s_part[qq][(s * W + l) * V + k2];
                            if constexpr (RED == kReduceMax) t2 = fmaxf(t2, pq);
                            else t2 = t2 + pq;
                        }
                        if (colok[s]) dst[col0 + s * W * V + k2] = t2;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int RED>
__global__ __launch_bounds__(kThreads) void spmm_longrow_combine_kernel(float* __restrict__ C, int N, int max_rows,
                                                                         const LongRowHeader* __restrict__ hdr,
                                                                         const int4* __restrict__ rowlist,
                                                                         const float* __restrict__ partial,
                                                                         const int32_t* __restrict__ perm) {
    int nrows = hdr->nrows;
    if (nrows > max_rows) nrows = max_rows;
    for (int j = blockIdx.x; j < nrows; j += gridDim.x) {
        const int4 e = rowlist[j];  // {row, first slot, #chunks}
        const float* src = partial + (size_t)e.y * (size_t)N;
        float* dst = C + (size_t)(perm ? perm[e.x] : e.x) * (size_t)N;  // plan mode: e.x is a permuted row id
        for (int col = threadIdx.x; col < N; col += kThreads) {
            float acc = src[col];
            for (int c = 1; c < e.z; ++c) {
                const float pq = src[(size_t)c * (size_t)N + col];
                if constexpr (RED == kReduceMax) acc = fmaxf(acc, pq);
                else acc = acc + pq;
            }
            dst[col] = acc;
        }
    }
}


// ----------------------------------------------------------------------------- slab-blocked path (dense graphs)
//
// Cache blocking for graphs whose B rows are reused hundreds of times (reddit-like:
// mean degree ~500) while B (119 MB at N=128) is far larger than a 4 MiB L2: with the
// streaming kernels 98 % of the 58.7 GB of gathers miss L2. The columns of A (= rows
// of B) are cut into slabs of `slab_rows` rows (~4 MB of B); the product is computed
// slab by slab, ONE LAUNCH PER SLAB, every launch adding each row's entries of that
// slab to C. A kernel boundary is the cheapest chip-wide barrier there is (~2 us):
// it keeps every workgroup on the same slab, so the slab is fetched once per XCD and
// then served from L2, and the hardware's workgroup scheduler balances the rows inside
// a slab. (A single persistent launch that sweeps the slabs was tried and measured:
// without a barrier the lane groups drift apart within a few slabs and the L2 benefit
// is gone — 8.2-9.2 ms vs 8.2 ms streaming on reddit-like.)
//
//   spmm_slabplan_kernel   per row: split points p_s = first CSR position (scanning
//                          forward from p_{s-1}) whose column is >= the end of slab s.
//                          For sorted rows that is the usual partition; for unsorted
//                          rows it is still a forward scan, so entries are consumed in
//                          CSR order either way and only the cache benefit shrinks.
//   spmm_slab_kernel       a lane group walks 8 consecutive rows; per row: acc = (slab > 0
//                          ? C[row] : 0), the row's entries in [p_s, p_{s+1}) as a U-deep
//                          gather stream through a per-group LDS tile, C[row] = acc. The
//                          next row's first tile and C row are prefetched meanwhile.
//
// Each output element is still ONE fp32 chain over the row's entries in CSR order (a
// value stored to C and reloaded is the same value), so the result is bit-identical
// to the other variants. The split points live in a stream-ordered temporary
// (hipMallocAsync) — no plan object, no API change.

// Slab id of a column: min(col / slab_rows, nslab - 1). The quotient is estimated with one fp32
// multiply by the reciprocal and fixed up: nslab <= 4096 keeps the estimate within one of the
// true quotient (relative error ~2^-23), and an integer division per CSR entry would cost more
// than the rest of the scan.
__device__ __forceinline__ int slab_of(int col, int slab_rows, float inv, int nslab) {
    int q = (int)((float)col * inv);
    if (q > nslab - 1) q = nslab - 1;
    const uint32_t lo = (uint32_t)q * (uint32_t)slab_rows;  // <= K + slab_rows < 2^32
    if (lo > (uint32_t)col) --q;
    else if ((uint32_t)col - lo >= (uint32_t)slab_rows && q < nslab - 1) ++q;
    return q;
}

__global__ __launch_bounds__(kThreads) void spmm_slabplan_kernel(const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ colind,
                                                                  int32_t* __restrict__ split, int M, int nslab,
                                                                  int slab_rows, float inv) {
    // split is [nslab + 1][M]. One wavefront per row, 64 entries per coalesced load, four
    // loads in flight. The slab id of the forward scan at entry p is the running maximum of
    // slab_of(col) over the row's entries up to p; wherever it steps up from m' to m,
    // split[m'+1..m] = p. Chunks whose ids are already non-decreasing (ascending columns, the
    // usual case — one wave-wide vote) skip the 6-step shuffle prefix maximum. (Deriving the
    // boundaries from ballots instead of per-entry ids was tried: slower, 302 vs 251 us on
    // reddit-like; the kernel is bound by its dependent rowptr -> colind -> store chain.)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * kWaves + (threadIdx.x >> 6);
    if (r >= M) return;
    const int lb = rowptr[r], hb = rowptr[r + 1];
    if (lane == 0) split[r] = lb;
    int run = 0;  // running maximum before the current chunk (wave-uniform)
    constexpr int D = 8;  // chunks (coalesced 256-byte loads) in flight per wavefront
    for (int sbase = lb; sbase < hb; sbase += 64 * D) {
        int c[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {  // clamped, unconditional: the D loads issue back to back
            const int p = sbase + 64 * k + lane;
            c[k] = load_csr(colind + (p < hb ? p : hb - 1));
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int base = sbase + 64 * k;
            if (base >= hb) break;  // wave-uniform
            const int p = base + lane;
            const bool valid = p < hb;
            int m = valid ? slab_of(c[k], slab_rows, inv, nslab) : 0;
            if (m < run) m = run;
            int prev = __shfl_up(m, 1, 64);
            if (lane == 0) prev = run;
            if (__any(valid && m < prev)) {  // columns not ascending here: prefix maximum
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(m, d, 64);
                    if (lane >= d && t > m) m = t;
                }
                prev = __shfl_up(m, 1, 64);
                if (lane == 0) prev = run;
            }
            if (valid)
                for (int sl = prev + 1; sl <= m; ++sl) split[(size_t)sl * M + r] = p;
            const int nvalid = (hb - base < 64) ? hb - base : 64;
            run = __shfl(m, nvalid - 1, 64);
        }
    }
    for (int sl = run + 1 + lane; sl <= nslab; sl += 64) split[(size_t)sl * M + r] = hb;
}

template <int V, int S, int W, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_slab_kernel(SpmmArgs a) {
    constexpr int G = 64 / W;
    constexpr int T = (W > 32) ? W : 32;
    constexpr int E = T / W;
    constexpr int U = (V * S >= 8) ? 4 : 8;  // U = 4 measured: 6.1 vs 4.8 ms on reddit-like (misses need the depth)
    constexpr int RMAX = kSlabRowsPerGroup;
    const int R = a.rpw;  // consecutive rows one lane group walks per launch (1..RMAX, wave-uniform)
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ off_t s_off[kWaves][G][T];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? G : 1][VALUED ? T : 1];
    __shared__ int s_b[kWaves][G][RMAX];
    __shared__ int s_e[kWaves][G][RMAX];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    int tile = 0, rb = blockIdx.x;
    if (a.ntile > 1) {
        tile = (int)blockIdx.x % a.ntile;
        rb = (int)blockIdx.x / a.ntile;
    }
    const int row0 = ((rb * kWaves + wave) * G + g) * R;
    if (((rb * kWaves + wave) * G) * R >= a.M) return;
    // split points of the group's R rows for this slab: two coalesced loads -> LDS
    // (clamped to the row's own CSR range: split points come from the caller's workspace when it asks
    // to reuse them, and a stale workspace must not turn into out-of-range reads or unbounded loops)
    for (int i = l; i < R; i += W) {
        const bool ok = row0 + i < a.M;
        int sb = 0, se = 0;
        if (ok) {
            const int lb = a.rowptr[row0 + i], hb = a.rowptr[row0 + i + 1];
            sb = load_csr(a.row_begin + row0 + i);
            se = load_csr(a.row_end + row0 + i);
            sb = sb < lb ? lb : (sb > hb ? hb : sb);
            se = se < sb ? sb : (se > hb ? hb : se);
        }
        s_b[wave][g][i] = sb;
        s_e[wave][g][i] = se;
    }
    wave_lds_sync();
    const bool first = a.accumulate == 0;

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

    // prefetch state for the NEXT row: its first tile of CSR entries and its C row
    int qc[E];
    float qv[E];
    float qacc[S][V];
    auto prefetch_row = [&](int i) {
        const int b = (i < R) ? s_b[wave][g][i] : 0;
        const int e = (i < R) ? s_e[wave][g][i] : 0;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const int p = b + l * E + k;
            qc[k] = 0;
            qv[k] = 0.0f;
            if (p < e) {
                qc[k] = load_csr(a.colind + p);
                if constexpr (VALUED) qv[k] = load_csr(a.val + p);
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int k = 0; k < V; ++k) qacc[s][k] = (RED == kReduceMax) ? a.empty : 0.0f;
        if (!first && b < e) {
            const float* Crow = a.C + (size_t)(row0 + i) * (size_t)a.N + col0;
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (colok[s]) load_vec_nt<V>(qacc[s], reinterpret_cast<const char*>(Crow + s * (W * V)));
        }
    };
    prefetch_row(0);

    for (int i = 0; i < R; ++i) {
        const int gb = s_b[wave][g][i];
        const int ge = s_e[wave][g][i];
        const bool rowok = row0 + i < a.M;
        // take over the prefetched tile / accumulator, then start on the next row's
        int pc[E];
        float pv[E];
        float acc[S][V];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            pc[k] = qc[k];
            pv[k] = qv[k];
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[s][k] = qacc[s][k];
        prefetch_row(i + 1);

        int tbase = gb;
        for (int k = gb; k < ge; k += U) {
            if (k == tbase) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    s_off[wave][g][l * E + e] = (off_t)(uint32_t)pc[e] * rowbytes;
                    if constexpr (VALUED) s_val[wave][g][l * E + e] = pv[e];
                }
                if (tbase + T < ge) {  // long segment: next tile of the same row
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int p = tbase + T + l * E + e;
                        if (p < ge) {
                            pc[e] = load_csr(a.colind + p);
                            if constexpr (VALUED) pv[e] = load_csr(a.val + p);
                        }
                    }
                }
                wave_lds_sync();
            }
            const int cnt = ge - k;
            const int t = k - tbase;
            off_t off[U];
            float v[U];
            float bv[U][S][V];
            if (cnt >= U) {  // full step: tile slots at constant offsets, U gathers back to back
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
#pragma unroll
                for (int j = 0; j < U; ++j)
#pragma unroll
                    for (int s = 0; s < S; ++s)
#pragma unroll
                        for (int k2 = 0; k2 < V; ++k2)
                            acc[s][k2] = combine<RED, VALUED>(acc[s][k2], v[j], bv[j][s][k2]);
            } else {  // last step of the segment: only the cnt live gathers are issued (LDS reads unconditional, clamped slot)
#pragma unroll
                for (int j = 0; j < U - 1; ++j) {
                    const int tj = t + ((j < cnt) ? j : cnt - 1);
                    off[j] = s_off[wave][g][tj];
                    if constexpr (VALUED) v[j] = s_val[wave][g][tj];
                    else v[j] = 1.0f;
                }
#pragma unroll
                for (int j = 0; j < U - 1; ++j) {
                    if (j < cnt) {
#pragma unroll
                        for (int s = 0; s < S; ++s) load_vec<V>(bv[j][s], Bbase + (off_t)(off[j] + cbytes[s]));
                    }
                }
#pragma unroll
                for (int j = 0; j < U - 1; ++j) {
                    if (j < cnt) {
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int k2 = 0; k2 < V; ++k2)
                                acc[s][k2] = combine<RED, VALUED>(acc[s][k2], v[j], bv[j][s][k2]);
                    }
                }
            }
            if (k + U >= tbase + T) {
                wave_lds_sync();
                tbase += T;
            }
        }
        wave_lds_sync();
        if (rowok && (first || gb < ge)) {  // later slabs skip rows they do not touch
            float* Crow = a.C + (size_t)(row0 + i) * (size_t)a.N + col0;
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (colok[s]) store_vec_sc1<V>(Crow + s * (W * V), acc[s]);  // written through: the L2 is for the slab (4.34 -> 4.25 ms)
        }
    }
}

// ----------------------------------------------------------------------------- parallel-reduction kernel
//
// Variant 5: the lanes of a W-wide group stride over ONE row's non-zeros (lane l takes
// entries l, l+W, ...), each lane keeps NC partial sums (NC output columns per pass), and
// the group combines them across lanes with ds_bpermute / DPP moves — no LDS storage:
//   1. reduce-scatter butterfly: log2(NC) xor steps, each lane passes on the half of its
//      partial sums that its partner is responsible for (NC-1 moves in total, instead of
//      NC * log2(W) for a plain all-reduce); afterwards lane l holds the column whose
//      index is the bit-reversal of its low log2(NC) bits;
//   2. xor butterfly over the remaining log2(W/NC) lane bits on that single value;
//   3. the NC low lanes store their column.
// Meant for narrow N (class logits, N = 1..8) and long rows, where a row-per-group
// kernel leaves most lanes idle: reddit-like N = 1: 0.7 ms vs 1.4 ms. The summation
// order differs from the reference's chain, so this variant is tolerance-checked
// (|delta| <= 1e-4 * max(|ref|, sum|a*b|)) and never chosen automatically.

template <int W, int NC, bool VALUED, bool IDX64>
__global__ __launch_bounds__(kThreads) void spmm_parreduce_kernel(SpmmArgs a) {
    static_assert(NC <= W && NC <= 16, "columns per pass");
    constexpr int G = 64 / W;
    constexpr int LOGNC = (NC == 1) ? 0 : (NC == 2) ? 1 : (NC == 4) ? 2 : (NC == 8) ? 3 : 4;
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const int item = (a.flags & kFlagNoXcdRemap) ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, a.nblk);
    const int row = (item * kWaves + wave) * G + g;
    const bool rowok = row < a.M;
    int lb = 0, hb = 0;
    if (rowok) {
        lb = a.rowptr[row];
        hb = a.rowptr[row + 1];
    }
    const bool vec4 = (NC % 4 == 0) && (a.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
    const bool vec2 = (NC == 2) && (a.N % 2 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 7) == 0);
    // column this lane ends up owning after the reduce-scatter: bit-reversed low bits
    int mycol = 0;
#pragma unroll
    for (int sft = 0; sft < LOGNC; ++sft) mycol |= ((l >> sft) & 1) << (LOGNC - 1 - sft);

    for (int c0 = 0; c0 < a.N; c0 += NC) {
        float part[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) part[i] = 0.0f;
        auto accumulate = [&](int k) {
            const off_t off = (off_t)(uint32_t)a.colind[k] * (off_t)a.N + (off_t)c0;
            const float v = VALUED ? a.val[k] : 1.0f;
            const float* brow = a.B + off;
            if (vec4 && c0 + NC <= a.N) {
#pragma unroll
                for (int i = 0; i < NC; i += 4) {
                    const VecT<4>::type q = *reinterpret_cast<const VecT<4>::type*>(brow + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) part[(i + j) % NC] = __builtin_fmaf(v, q[j], part[(i + j) % NC]);
                }
            } else if (vec2) {
                const VecT<2>::type q = *reinterpret_cast<const VecT<2>::type*>(brow);
                part[0] = __builtin_fmaf(v, q[0], part[0]);
                part[1 % NC] = __builtin_fmaf(v, q[1], part[1 % NC]);
            } else {
#pragma unroll
                for (int i = 0; i < NC; ++i)
                    if (c0 + i < a.N) part[i] = __builtin_fmaf(v, brow[i], part[i]);
            }
        };
        int k = lb + l;
        for (; k + W < hb; k += 2 * W) {  // two entries per lane in flight
            accumulate(k);
            accumulate(k + W);
        }
        if (k < hb) accumulate(k);

        // 1. reduce-scatter over the low log2(NC) lane bits
#pragma unroll
        for (int sft = 0; sft < LOGNC; ++sft) {
            const int half = NC >> (sft + 1);
            const bool upper = (l >> sft) & 1;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float keep = upper ? part[half + i] : part[i];
                const float give = upper ? part[i] : part[half + i];
                part[i] = keep + __shfl_xor(give, 1 << sft, 64);
            }
        }
        // 2. all-reduce of the remaining single value over the other lane bits
        float total = part[0];
#pragma unroll
        for (int m = NC; m < W; m <<= 1) total += __shfl_xor(total, m, 64);
        // 3. NC low lanes write their column
        if (rowok && l < NC && c0 + mycol < a.N) a.C[(size_t)row * (size_t)a.N + c0 + mycol] = total;
    }
}

// ----------------------------------------------------------------------------- host-side launch table

template <int V, int S, int W, bool VALUED, bool IDX64>
static hipError_t launch_naive(const SpmmArgs& a, hipStream_t st) {
    constexpr int G = 64 / W;
    SpmmArgs args = a;
    args.nblk = (int)(((int64_t)a.M + kWaves * G - 1) / (kWaves * G));
    args.ntile = (a.N + W * V * S - 1) / (W * V * S);
    const int64_t nitems = (int64_t)args.nblk * args.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((spmm_naive_kernel<V, S, W, VALUED, IDX64>), dim3((unsigned)nitems), dim3(kThreads), 0, st,
                       args);
    return hipGetLastError();
}

template <int V, int S, bool VALUED, bool IDX64>
static hipError_t naive_w(const SpmmArgs& a, int W, hipStream_t st) {
    switch (W) {
        case 4: return launch_naive<V, S, 4, VALUED, IDX64>(a, st);
        case 8: return launch_naive<V, S, 8, VALUED, IDX64>(a, st);
        case 16: return launch_naive<V, S, 16, VALUED, IDX64>(a, st);
        case 32: return launch_naive<V, S, 32, VALUED, IDX64>(a, st);
        case 64: return launch_naive<V, S, 64, VALUED, IDX64>(a, st);
    }
    return hipErrorInvalidValue;
}

template <bool VALUED, bool IDX64>
static hipError_t naive_vs(const SpmmArgs& a, const Geometry& g, hipStream_t st) {
    if (g.strips == 2) {
        if (g.vec == 4) return naive_w<4, 2, VALUED, IDX64>(a, g.group, st);
        return hipErrorInvalidValue;
    }
    switch (g.vec) {
        case 1: return naive_w<1, 1, VALUED, IDX64>(a, g.group, st);
        case 2: return naive_w<2, 1, VALUED, IDX64>(a, g.group, st);
        case 4: return naive_w<4, 1, VALUED, IDX64>(a, g.group, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_spmm_naive(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    if (geo.reduce != kReduceSum) return hipErrorInvalidValue;
    const bool valued = a.val != nullptr;
    if (valued) return geo.idx64 ? naive_vs<true, true>(a, geo, st) : naive_vs<true, false>(a, geo, st);
    return geo.idx64 ? naive_vs<false, true>(a, geo, st) : naive_vs<false, false>(a, geo, st);
}

// The plain instantiations live here, the plan-mode ones in spmm_stream_plan.hip.
hipError_t launch_spmm_stream(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    if (a.tasks) return launch_spmm_stream_planned(a, geo, st);
    return launch_spmm_stream_impl<false>(a, geo, st);
}

hipError_t launch_spmm_segstream(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    if (a.gtasks) return launch_spmm_segstream_planned(a, geo, st);
    return launch_spmm_segstream_impl<false>(a, geo, st);
}


// Workspace of one long-row pass: header, long-row list, chunk list, partial rows — one
// stream-ordered allocation sized by upper bounds (#long rows < nnz / long_row, #chunks <=
// nnz / chunk + #long rows), zeroed header, freed stream-ordered after the combine kernel.
struct LongRowWs {
    LongRowHeader* hdr;
    int4* rowlist;
    int2* chunklist;
    float* partial;
    int max_rows, max_chunks, chunk;
};

template <int V, int S, int W, bool VALUED, bool IDX64, int RED>
static hipError_t launch_longrow(const SpmmArgs& a, const LongRowWs& ws, hipStream_t st) {
    // at most 8 workgroups per CU; the chunk list is walked grid-stride
    int blocks = ws.max_chunks < 2048 ? ws.max_chunks : 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((spmm_longrow_chunk_kernel<V, S, W, VALUED, IDX64, RED>), dim3((unsigned)blocks),
                       dim3(kThreads), 0, st, a, ws.chunk, ws.max_chunks, ws.hdr, ws.chunklist, ws.partial);
    return hipGetLastError();
}

template <int V, int S, bool VALUED, bool IDX64, int RED>
static hipError_t longrow_w(const SpmmArgs& a, const LongRowWs& ws, int W, hipStream_t st) {
    switch (W) {
        case 4: return launch_longrow<V, S, 4, VALUED, IDX64, RED>(a, ws, st);
        case 8: return launch_longrow<V, S, 8, VALUED, IDX64, RED>(a, ws, st);
        case 16: return launch_longrow<V, S, 16, VALUED, IDX64, RED>(a, ws, st);
        case 32: return launch_longrow<V, S, 32, VALUED, IDX64, RED>(a, ws, st);
        case 64: return launch_longrow<V, S, 64, VALUED, IDX64, RED>(a, ws, st);
    }
    return hipErrorInvalidValue;
}

template <bool VALUED, bool IDX64, int RED>
static hipError_t longrow_vs(const SpmmArgs& a, const LongRowWs& ws, const Geometry& g, hipStream_t st) {
    if (g.strips == 2) {
        if (g.vec == 4) return longrow_w<4, 2, VALUED, IDX64, RED>(a, ws, g.group, st);
        if (g.group != 64) return hipErrorInvalidValue;
        if (g.vec == 2) return launch_longrow<2, 2, 64, VALUED, IDX64, RED>(a, ws, st);
        return launch_longrow<1, 2, 64, VALUED, IDX64, RED>(a, ws, st);
    }
    switch (g.vec) {
        case 1: return longrow_w<1, 1, VALUED, IDX64, RED>(a, ws, g.group, st);
        case 2: return longrow_w<2, 1, VALUED, IDX64, RED>(a, ws, g.group, st);
        case 4: return longrow_w<4, 1, VALUED, IDX64, RED>(a, ws, g.group, st);
    }
    return hipErrorInvalidValue;
}

size_t longrows_workspace_bytes(int64_t nnz, int64_t N, int long_row) {
    if (long_row <= 0 || nnz <= long_row) return 0;
    const int64_t max_rows = nnz / long_row + 1;
    const int64_t max_chunks = nnz / kLongRowChunk + max_rows;
    size_t off = sizeof(LongRowHeader) + (size_t)max_rows * sizeof(int4) + (size_t)max_chunks * sizeof(int2);
    off = (off + 255) & ~(size_t)255;
    return off + (size_t)max_chunks * (size_t)N * sizeof(float);
}

// Long-row pass around the main kernel: `begin` sizes and zeroes the workspace and points the main
// kernel's arguments at it (the main kernel registers every row it skips: chunk slots from one atomic
// counter, one list entry); `finish` runs the chunk and combine kernels and releases the workspace.
hipError_t longrows_begin(SpmmArgs& a, int64_t nnz, void* ext_ws, size_t ext_bytes, LongRowWs& ws, bool& own,
                          hipStream_t st) {
    a.lr_hdr = nullptr;
    own = false;
    ws.hdr = nullptr;
    if (a.long_row <= 0 || nnz <= a.long_row) return hipSuccess;  // no row can be long
    ws.chunk = kLongRowChunk;
    const int64_t max_rows = nnz / a.long_row + 1;
    const int64_t max_chunks = nnz / ws.chunk + max_rows;
    if (max_chunks > 0x7fffffffLL) return hipErrorInvalidConfiguration;
    ws.max_rows = (int)max_rows;
    ws.max_chunks = (int)max_chunks;
    const size_t off_rows = sizeof(LongRowHeader);
    const size_t off_chunks = off_rows + (size_t)max_rows * sizeof(int4);
    size_t off_partial = off_chunks + (size_t)max_chunks * sizeof(int2);
    off_partial = (off_partial + 255) & ~(size_t)255;
    const size_t bytes = off_partial + (size_t)max_chunks * (size_t)a.N * sizeof(float);
    char* base = nullptr;
    hipError_t e = hipSuccess;
    own = !(ext_ws && ext_bytes >= bytes && (reinterpret_cast<uintptr_t>(ext_ws) & 15) == 0);
    if (own) e = workspace_alloc(reinterpret_cast<void**>(&base), bytes, st);
    else base = static_cast<char*>(ext_ws);
    if (e != hipSuccess) return e;
    ws.hdr = reinterpret_cast<LongRowHeader*>(base);
    ws.rowlist = reinterpret_cast<int4*>(base + off_rows);
    ws.chunklist = reinterpret_cast<int2*>(base + off_chunks);
    ws.partial = reinterpret_cast<float*>(base + off_partial);
    hipLaunchKernelGGL(spmm_longrow_reset_kernel, dim3(1), dim3(64), 0, st, ws.hdr);
    e = hipGetLastError();
    a.lr_hdr = reinterpret_cast<int32_t*>(ws.hdr);
    a.lr_rows = reinterpret_cast<int32_t*>(ws.rowlist);
    a.lr_chunks = reinterpret_cast<int32_t*>(ws.chunklist);
    a.lr_chunk = ws.chunk;
    a.lr_max_rows = ws.max_rows;
    a.lr_max_chunks = ws.max_chunks;
    return e;
}

hipError_t longrows_finish(const SpmmArgs& a, const Geometry& geo, const LongRowWs& ws, bool own, hipError_t e,
                           hipStream_t st) {
    if (!ws.hdr) return e;
    const bool valued = a.val != nullptr;
    if (e == hipSuccess && geo.reduce == kReduceMax && valued) e = hipErrorInvalidValue;
    if (e == hipSuccess) {
        if (geo.reduce == kReduceMax)
            e = geo.idx64 ? longrow_vs<false, true, kReduceMax>(a, ws, geo, st) : longrow_vs<false, false, kReduceMax>(a, ws, geo, st);
        else if (valued)
            e = geo.idx64 ? longrow_vs<true, true, kReduceSum>(a, ws, geo, st) : longrow_vs<true, false, kReduceSum>(a, ws, geo, st);
        else
            e = geo.idx64 ? longrow_vs<false, true, kReduceSum>(a, ws, geo, st) : longrow_vs<false, false, kReduceSum>(a, ws, geo, st);
    }
    if (e == hipSuccess) {
        const int blocks = ws.max_rows < 256 ? ws.max_rows : 256;
        if (geo.reduce == kReduceMax)
            hipLaunchKernelGGL(spmm_longrow_combine_kernel<kReduceMax>, dim3((unsigned)blocks), dim3(kThreads), 0, st, a.C,
                               a.N, ws.max_rows, ws.hdr, ws.rowlist, ws.partial, a.perm);
        else
            hipLaunchKernelGGL(spmm_longrow_combine_kernel<kReduceSum>, dim3((unsigned)blocks), dim3(kThreads), 0, st, a.C,
                               a.N, ws.max_rows, ws.hdr, ws.rowlist, ws.partial, a.perm);
        e = hipGetLastError();
    }
    const hipError_t ef = own ? workspace_free(ws.hdr, st) : hipSuccess;
    return e != hipSuccess ? e : ef;
}

// main streaming kernel + long-row pass
hipError_t launch_spmm_stream_with_longrows(SpmmArgs a, const Geometry& geo, int64_t nnz, void* ext_ws, size_t ext_bytes,
                                            hipStream_t st) {
    LongRowWs ws;
    bool own = false;
    hipError_t e = longrows_begin(a, nnz, ext_ws, ext_bytes, ws, own, st);
    if (e == hipSuccess) e = launch_spmm_stream(a, geo, st);
    return longrows_finish(a, geo, ws, own, e, st);
}


template <int V, int S, int W, bool VALUED, bool IDX64, int RED>
static hipError_t launch_slab(const SpmmArgs& a, hipStream_t st) {
    constexpr int G = 64 / W;
    SpmmArgs args = a;
    int R = a.rpw;  // rows per lane group (slab path)
    if (R < 1) R = 1;
    if (R > kSlabRowsPerGroup) R = kSlabRowsPerGroup;
    args.rpw = R;
    args.nblk = (int)(((int64_t)a.M + kWaves * G * R - 1) / (kWaves * G * R));
    args.ntile = (a.N + W * V * S - 1) / (W * V * S);
    const int64_t nitems = (int64_t)args.nblk * args.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((spmm_slab_kernel<V, S, W, VALUED, IDX64, RED>), dim3((unsigned)nitems), dim3(kThreads), 0, st,
                       args);
    return hipGetLastError();
}

template <int V, int S, bool VALUED, bool IDX64, int RED>
static hipError_t slab_w(const SpmmArgs& a, int W, hipStream_t st) {
    switch (W) {
        case 4: return launch_slab<V, S, 4, VALUED, IDX64, RED>(a, st);
        case 8: return launch_slab<V, S, 8, VALUED, IDX64, RED>(a, st);
        case 16: return launch_slab<V, S, 16, VALUED, IDX64, RED>(a, st);
        case 32: return launch_slab<V, S, 32, VALUED, IDX64, RED>(a, st);
        case 64: return launch_slab<V, S, 64, VALUED, IDX64, RED>(a, st);
    }
    return hipErrorInvalidValue;
}

template <bool VALUED, bool IDX64, int RED>
static hipError_t slab_vs(const SpmmArgs& a, const Geometry& g, hipStream_t st) {
    if (g.strips == 2) {
        if (g.vec == 4) return slab_w<4, 2, VALUED, IDX64, RED>(a, g.group, st);
        if (g.group != 64) return hipErrorInvalidValue;
        if (g.vec == 2) return launch_slab<2, 2, 64, VALUED, IDX64, RED>(a, st);
        return launch_slab<1, 2, 64, VALUED, IDX64, RED>(a, st);
    }
    switch (g.vec) {
        case 1: return slab_w<1, 1, VALUED, IDX64, RED>(a, g.group, st);
        case 2: return slab_w<2, 1, VALUED, IDX64, RED>(a, g.group, st);
        case 4: return slab_w<4, 1, VALUED, IDX64, RED>(a, g.group, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_slabplan(const int32_t* rowptr, const int32_t* colind, int32_t* split, int M, int nslab,
                           int slab_rows, hipStream_t st) {
    hipLaunchKernelGGL(spmm_slabplan_kernel, dim3((M + kWaves - 1) / kWaves), dim3(kThreads), 0, st, rowptr, colind,
                       split, M, nslab, slab_rows, 1.0f / (float)slab_rows);
    return hipGetLastError();
}

size_t slabblocked_workspace_bytes(int64_t M, const Geometry& geo) {
    const int64_t nslab = ((int64_t)geo.K + geo.slab_rows - 1) / geo.slab_rows;
    return (size_t)(nslab + 1) * (size_t)M * 4;
}

hipError_t launch_spmm_slabblocked(const SpmmArgs& a0, const Geometry& geo, void* ext_ws, size_t ext_bytes,
                                   hipStream_t st) {
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
    // a caller that keeps its workspace across calls on an unchanged graph may skip the scan
    if (!(!own && (a0.flags & kFlagReuseSplit)))
        e = launch_slabplan(a0.rowptr, a0.colind, split, M, nslab, geo.slab_rows, st);
    const bool valued = a0.val != nullptr;
    for (int sl = 0; sl < nslab && e == hipSuccess; ++sl) {
        SpmmArgs a = a0;
        a.row_begin = split + (size_t)sl * M;
        a.row_end = split + (size_t)(sl + 1) * M;
        a.accumulate = sl > 0 ? 1 : 0;
        if (geo.reduce == kReduceMax)
            e = geo.idx64 ? slab_vs<false, true, kReduceMax>(a, geo, st) : slab_vs<false, false, kReduceMax>(a, geo, st);
        else if (valued)
            e = geo.idx64 ? slab_vs<true, true, kReduceSum>(a, geo, st) : slab_vs<true, false, kReduceSum>(a, geo, st);
        else
            e = geo.idx64 ? slab_vs<false, true, kReduceSum>(a, geo, st) : slab_vs<false, false, kReduceSum>(a, geo, st);
    }
    const hipError_t ef = own ? workspace_free(split, st) : hipSuccess;
    return e != hipSuccess ? e : ef;
}

template <int W, int NC, bool VALUED, bool IDX64>
static hipError_t launch_parreduce_nc(const SpmmArgs& a, hipStream_t st) {
    constexpr int G = 64 / W;
    SpmmArgs args = a;
    args.nblk = (int)(((int64_t)a.M + kWaves * G - 1) / (kWaves * G));
    args.ntile = 1;
    if (args.nblk <= 0) return hipSuccess;
    hipLaunchKernelGGL((spmm_parreduce_kernel<W, NC, VALUED, IDX64>), dim3(args.nblk), dim3(kThreads), 0, st, args);
    return hipGetLastError();
}

template <int W, bool VALUED, bool IDX64>
static hipError_t launch_parreduce_w(const SpmmArgs& a, hipStream_t st) {
    // columns per pass: the smallest power of two covering N, at most 16 (and at most W)
    if (a.N <= 1) return launch_parreduce_nc<W, 1, VALUED, IDX64>(a, st);
    if (a.N <= 2) return launch_parreduce_nc<W, 2, VALUED, IDX64>(a, st);
    if (a.N <= 4 || W < 8) return launch_parreduce_nc<W, 4, VALUED, IDX64>(a, st);
    if (a.N <= 8 || W < 16) return launch_parreduce_nc<W, (W >= 8 ? 8 : 4), VALUED, IDX64>(a, st);
    return launch_parreduce_nc<W, (W >= 16 ? 16 : 4), VALUED, IDX64>(a, st);
}

hipError_t launch_spmm_parreduce(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    const bool valued = a.val != nullptr;
#define GESPMM_PR(W)                                                                         \
    case W:                                                                                   \
        if (valued) return geo.idx64 ? launch_parreduce_w<W, true, true>(a, st) : launch_parreduce_w<W, true, false>(a, st); \
        return geo.idx64 ? launch_parreduce_w<W, false, true>(a, st) : launch_parreduce_w<W, false, false>(a, st);
    switch (geo.group) {
        GESPMM_PR(4)
        GESPMM_PR(8)
        GESPMM_PR(16)
        GESPMM_PR(32)
        GESPMM_PR(64)
    }
#undef GESPMM_PR
    return hipErrorInvalidValue;
}

}  // namespace gespmm
