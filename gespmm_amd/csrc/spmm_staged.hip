// spmm_staged.hip — the plan's kernel for clustered matrices with rows long enough to share neighbours inside a block:
// scalar-stream walk + the block's most used B rows staged in LDS (round 3).
//
// What bounds the streaming kernels on such a matrix (products-shaped communities, N = 128: 75 % of the gathers hit the L2
// and the product still takes 3.9 ms = 16 TB/s of gathers) is the vector memory path: every wave-level load instruction costs
// the address unit the same ~16 cycles whether it carries 256 bytes or 1 KB, whether it hits or misses — moving hits from L2
// to LDS through masked, flat or out-of-range loads changes nothing (profiles/r03/hotrows_experiment.log). The only gather that
// is free for the address unit is one that is never issued. Hence:
//
//   * ONE ROW PER WAVEFRONT AT A TIME: a B row of N = 64 * VEC floats is one load of VEC dwords per lane. Everything that is the
//     same for the 64 lanes — the CSR stream (code, value), row ends, C row ids — lives in SGPRs and arrives through the scalar
//     cache; "is this entry's B row staged?" is a SCALAR branch around one of {ds_read, global_load}: a staged entry issues no
//     vector memory instruction at all;
//   * a workgroup of 16 wavefronts owns a BLOCK of 96 (N = 128) / 64 (N = 256) consecutive rows of the plan's clustered matrix; the analysis
//     (plan_device.hip: device_build_staging) lists per block the <= H columns used most often inside it (>= 2 uses; H rows =
//     64 KB) and rewrites the block's entries: code >= 0 = column, code < 0 = slot of the staged row. The workgroup copies the
//     listed rows into LDS once, coalesced, then each wavefront walks its share of the block's rows as one stream;
//   * the scalar unit issues one instruction per clock per CU, so the loop is written to need ~4 of them per entry: {code,
//     value} interleaved (one s_load per chunk of 8 entries, the next chunk requested before this one's gathers — the CSR
//     stream is read once, every scalar load goes to memory), ONE vector instruction forms the offset that serves either path
//     (`code << log2(row bytes)` drops the flag bit: LDS address of the staged row or byte offset of the B row, + the lane's
//     offset), compare + branch, and the multiply-adds take the value straight from its SGPR.
//
// Every output element is still ONE fp32 chain over the row's entries in CSR order with one fused multiply-add per entry
// (spmm_test.cu:182-203 semantics; unweighted matrices carry 1.0f: fma(1, b, acc) == acc + b exactly), so the bits are those
// of every other variant. Sum reducer, N = 128 or 256, K * N * 4 < 4 GB (32-bit offsets); everything else stays on the
// streaming kernels. Rows of more than kStagedMaxRow entries never reach this kernel: the plan empties them in the row pointers it
// passes here and runs them through the streaming kernel's long-row pass afterwards (plan.cpp: plan_run).
//
// The gathers are inline assembly: written as C++ the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read (it
// cannot see that the two paths never write the same register in the same pass) and one memory access is in flight at a time.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "spmm_device.h"
#include "spmm_kernels.h"

#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__) || defined(__gfx942__)
// (the inline assembly below spells its memory instructions — sc1 / nt modifiers, SGPR-base global loads — for gfx942 / gfx950)
#else
#error "spmm_staged.hip is written for gfx950 (gfx942 ISA compatible): its inline assembly does not assemble elsewhere"
#endif

// Experiments only: -DGESPMM_STAGED_INSTRUMENT=1 compiles the GESPMM_STAGED_DEBUG knobs in (1 = no staging copy, 2 = every gather from
// LDS — both give WRONG results, they time the skeleton; 4 = per-wavefront phase clocks printed by the launcher). Off: they fold away.
#ifndef GESPMM_STAGED_INSTRUMENT
#define GESPMM_STAGED_INSTRUMENT 0
#endif

namespace gespmm {

namespace {

typedef const __attribute__((address_space(4))) int32_t* cint_ptr;  // constant address space: scalar loads
using f4v = float __attribute__((ext_vector_type(4)));
using i2v = int __attribute__((ext_vector_type(2)));

template <int VEC> struct LaneVec;
template <> struct LaneVec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct LaneVec<4> { using type = float __attribute__((ext_vector_type(4))); };

// Wider matrices are COLUMN-TILED (round 4): N = 64 * VEC << TSHIFT, a workgroup computes ONE tile of 64 * VEC columns of its block
// of rows; tile t is bound to the XCDs whose id is t modulo the tile count (workgroup ids go to XCDs round-robin), so an XCD's L2
// only ever holds its own tile's columns of B, and inside that XCD group the blocks stay contiguous (xcd_contiguous, generalised).
// LDS holds the tile's part of the staged rows: H and the block height are those of the tile width, the tables are shared by
// all tiles. The entry stream is read once per tile.
// PAGE2: B between 4 and 8 GB (products-shaped x 512 columns: 5.0 GB). The lane's 32-bit offset wraps modulo 4 GB by itself —
// `code << log2(row bytes)` drops the bit that says which half — and the base pointer is chosen between B and B + 4 GB by
// that bit of the (scalar) code: two scalar instructions on the memory path, none on the LDS path.
template <int VEC, int U, int TSHIFT, bool PAGE2, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmm_staged_kernel(StagedArgs a) {
    constexpr int kStagedWaves = WAVES;
    constexpr int kStagedLdsBytes = WAVES * kStagedLdsPerWave;
    using vec_t = typename LaneVec<VEC>::type;
    constexpr int kRowBytes = 256 * VEC;          // bytes of a row inside one tile
    constexpr int kRowShift = (VEC == 2) ? 9 : 10;
    constexpr int kGlobalShift = kRowShift + TSHIFT;  // log2(N * 4): row stride of B and C
    constexpr int H = kStagedLdsBytes / kRowBytes;  // staged rows per block
    constexpr int kRowF4 = kRowBytes / 16;
    constexpr int kWin = 64;                        // entries of the stream one wavefront holds in a register pair
    static_assert(kStagedPad >= kWin, "a window is read whole: up to kWin - 1 entries past a task's end");
    static_assert(kWin % U == 0 && (U == 8 || U == 16), "whole chunks per window");
    static_assert(TSHIFT >= 0 && TSHIFT <= 3, "at most one tile per XCD");
    __shared__ f4v s_hot[H * kRowF4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, tile = 0;
    if constexpr (TSHIFT == 0) {
        blk = xcd_contiguous(blockIdx.x, a.nblocks);
    } else {
        constexpr int NX = 8 >> TSHIFT;  // XCDs that serve one tile
        tile = (int)blockIdx.x & ((1 << TSHIFT) - 1);
        const int x = ((int)blockIdx.x & 7) >> TSHIFT, idx = (int)blockIdx.x >> 3;
        const int q = a.nblocks / NX, r = a.nblocks % NX;
        blk = ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
    }
    const int dbg = GESPMM_STAGED_INSTRUMENT ? a.debug : 0;
    const uint64_t t_start = (dbg & 4) ? __builtin_readcyclecounter() : 0;
    const int task = blk * kStagedWaves + wave;
    // Round trip 1 — everything whose address follows from the block id alone: the wavefront's task (scalar) and the block's staged
    // columns (vector; thread t copies the 16-byte pieces t, t + T, t + 2T, t + 3T of the H x row-bytes array, T = threads per block).
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int32_t* hc = a.hot_cols + (size_t)blk * H;
    static_assert(H * kRowF4 == 4 * kStagedWaves * 64, "four pieces per thread");
    int hcol[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) hcol[u] = (dbg & 1) ? -1 : hc[(u * kStagedWaves * 64 + tid) / kRowF4];
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    const float* Bp = a.B + (size_t)tile * (64 * VEC);
    const float* BpHi = Bp + (1ull << 30);  // + 4 GB (PAGE2)
    (void)BpHi;
    const uint32_t loff = (uint32_t)lane * (4u * VEC);

    // Round trip 2 — the staged rows (slots the block does not use hold -1: no load), and the wavefront's own metadata, all of it
    // through the VECTOR path: the first window of the entry stream, and the row ends / C rows of its first 64 rows (lane i: row i).
    // Nothing on the walk below is a scalar MEMORY load — see the header: a scalar load shares its counter with the LDS reads and
    // is waited for with every chunk.
    const f4v* B4 = reinterpret_cast<const f4v*>(a.B);
    f4v stage[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = u * kStagedWaves * 64 + tid;
        stage[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (hcol[u] >= 0) stage[u] = B4[(((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4)];
    }
    const i2v* evv = reinterpret_cast<const i2v*>(a.ev) + wb;
    i2v win = {0, 0};
    int rpv = 0, pmv = 0;
    if (nrows > 0) {
        win = __builtin_nontemporal_load(evv + lane);  // (the stream is padded: a whole window is always readable)
        const int rl = (lane < nrows) ? lane : nrows - 1;  // rows 0 .. 63 of the task (clamped to its last row)
        rpv = __builtin_nontemporal_load(a.rowptr + row_first + 1 + rl);
        pmv = __builtin_nontemporal_load(a.perm + row_first + rl);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (hcol[u] >= 0) s_hot[u * kStagedWaves * 64 + tid] = stage[u];
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start
    uint64_t t_staged = 0;
    if (dbg & 4) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        t_staged = __builtin_readcyclecounter();
    }
    if (nrows == 0) return;
    // one offset serves both paths (LDS address of a staged row / byte offset into B): the staging array must sit at LDS address 0
    // (it is the kernel's only LDS object; a compile-time constant — the check folds away)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot != 0u) __builtin_trap();
    int cur = 0;
    int rend = __builtin_amdgcn_readlane(rpv, 0);
    int crow = __builtin_amdgcn_readlane(pmv, 0);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    auto flush = [&]() {  // row `cur` is complete: store it, step to the next one
        float* Crow = a.C + (((size_t)crow << TSHIFT) + (size_t)tile) * (size_t)(64 * VEC);
        vec_t out;
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = acc[i];
        if constexpr (VEC == 2) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(out), "s"(Crow) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(loff), "v"(out), "s"(Crow) : "memory");
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        ++cur;
        if (cur < nrows) {
            if ((cur & 63) == 0) {
                // a task of more than 64 rows (rare: a block is 24-96 rows for 4-16 wavefronts): the next 64 row ends / C rows. Assembly, so
                // that the wait for these two loads sits in THIS branch — as C++ loads the compiler waits for the vector memory counter
                // at the join, i.e. after every row's store (measured: the walk then takes a store round trip per row)
                const int rl = (cur + lane < nrows) ? cur + lane : nrows - 1;
                const int32_t* rp_src = a.rowptr + row_first + 1 + rl;
                const int32_t* pm_src = a.perm + row_first + rl;
                asm volatile(
                    "global_load_dword %0, %2, off nt\n\t"
                    "global_load_dword %1, %3, off nt\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(rpv), "=&v"(pmv)
                    : "v"(rp_src), "v"(pm_src)
                    : "memory");
            }
            rend = __builtin_amdgcn_readlane(rpv, cur & 63);
            crow = __builtin_amdgcn_readlane(pmv, cur & 63);
        }
    };
    // code: bit 31 = staged (low bits: LDS slot); else the column
    auto gather = [&](int code, vec_t& d) {
        // (the reference to s_hot keeps the staging stores alive: the LDS reads below are invisible to the compiler)
        // (one tile: the same offset serves both paths — LDS address of the staged row / byte offset of the B row — and the
        //  compiler keeps one register; tiled: the strides differ)
        const uint32_t voff_l =
            ((uint32_t)code << kRowShift) + loff + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;
        const uint32_t voff = (TSHIFT == 0) ? voff_l : ((uint32_t)code << kGlobalShift) + loff;
        if constexpr (PAGE2) {
            static_assert(!PAGE2 || VEC == 4, "paged bases exist for the 256-column tiles");
            constexpr int kPageBit = 32 - kGlobalShift;  // bit of the column that selects the 4 GB half
            uint64_t base;  // scratch SGPR pair: the chosen half's base
            asm volatile(
                "s_cmp_lt_i32 %3, 0\n\t"
                "s_cbranch_scc1 1f\n\t"
                "s_bitcmp1_b32 %3, %7\n\t"
                "s_cselect_b64 %1, %6, %4\n\t"
                "global_load_dwordx4 %0, %2, %1\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b128 %0, %5\n"
                "2:"
                : "=&v"(d), "=&s"(base)
                : "v"(voff), "s"(code), "s"(Bp), "v"(voff_l), "s"(BpHi), "n"(kPageBit)
                : "memory", "scc");
        } else if constexpr (VEC == 2)
            asm volatile(
                "s_cmp_lt_i32 %2, 0\n\t"
                "s_cbranch_scc1 1f\n\t"
                "global_load_dwordx2 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b64 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp), "v"(voff_l)
                : "memory", "scc");
        else
            asm volatile(
                "s_cmp_lt_i32 %2, 0\n\t"
                "s_cbranch_scc1 1f\n\t"
                "global_load_dwordx4 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b128 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp), "v"(voff_l)
                : "memory", "scc");
    };
    auto gather_lds = [&](int code, vec_t& d) {  // a chunk whose entries are all staged: no branch, no vector memory
        const uint32_t voff_l =
            ((uint32_t)code << kRowShift) + loff + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;
        if constexpr (VEC == 2) asm volatile("ds_read_b64 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
    };
    auto fma_row = [&](int vbits, const vec_t& b) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(vbits), "v"(b[i]));
    };
    // (Window slots past `we` belong to the next task or the padding: harmless gathers — staged slots are < H, columns are
    // valid — that are not summed.)
    for (int kw = wb; kw < we; kw += kWin) {
        // the next window is requested a whole window (kWin / U chunks) before it is needed; only chunks that gather from memory
        // themselves wait for the vector memory counter, and those wait for their own (younger) loads anyway
        i2v nxt = win;
        if (kw + kWin < we) nxt = __builtin_nontemporal_load(evv + (kw - wb) + kWin + lane);
        const uint64_t gmask = (dbg & 2) ? 0ull : __ballot(win.x >= 0);  // entries of this window whose B row comes from memory
#pragma unroll 1
        for (int c = 0; c < kWin; c += U) {
            const int k = kw + c;
            if (k >= we) break;
            int code[U], vb[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                code[j] = __builtin_amdgcn_readlane(win.x, c + j);
                vb[j] = __builtin_amdgcn_readlane(win.y, c + j);
            }
            vec_t bv[U];
            const uint32_t anymem = (uint32_t)(gmask >> c) & ((1u << U) - 1u);
            if (anymem == 0) {
#pragma unroll
                for (int j = 0; j < U; ++j) gather_lds(code[j], bv[j]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) gather(code[j], bv[j]);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < U; ++j) asm volatile("" : "+v"(bv[j]));  // (uses of bv stay behind the wait)
            if (k + U <= rend) {
#pragma unroll
                for (int j = 0; j < U; ++j) fma_row(vb[j], bv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    if (k + j < we) {
                        while (k + j >= rend) flush();  // rows ending before this entry (incl. empty ones)
                        fma_row(vb[j], bv[j]);
                    }
                }
            }
        }
        win = nxt;
    }
    while (cur < nrows) flush();  // last row and any trailing empty rows
    if (dbg & 4) {
        const uint64_t t_walk = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint64_t t_end = __builtin_readcyclecounter();
        if (lane == 0) {  // one record per wavefront (atomics on four words would serialise the whole launch)
            unsigned long long* rec = a.dbg_clk + (size_t)task * 4;
            rec[0] = t_staged - t_start;
            rec[1] = t_walk - t_staged;
            rec[2] = t_end - t_walk;
            rec[3] = t_start;
        }
    }
}

}  // namespace

// Width of one column tile (128 or 256 columns) and log2 of the tile count, or 0 / -1 when the width is not served:
// N = 128, and N = 256 * 2^t for t = 0..2 (the B / C row stride is formed by a shift).
static int staged_tile_cols(int64_t N, int* tshift) {
    if (N == 128) { *tshift = 0; return 128; }
    for (int t = 0; t <= 2; ++t)
        if (N == (256ll << t)) { *tshift = t; return 256; }
    *tshift = -1;
    return 0;
}

StagedShape staged_shape(int64_t N) {
    // rows per block, measured on the products-shaped community graph with 16 wavefronts (us at N = 128 / 256): 64 rows 3286 / 5584,
    // 80: 3122 / 5596, 96: 3012 / 5834, 112: 3120 / 6264, 128: 3065 / 6440 — about as many rows as LDS holds staged rows for the
    // width's row size (profiles/r03/staged_rows.log): 6 rows per wavefront at 128-column tiles, 4 at 256
    static const int waves_env = getenv("GESPMM_STAGED_WAVES") ? atoi(getenv("GESPMM_STAGED_WAVES")) : 0;
    static const int rows_env = getenv("GESPMM_STAGED_ROWS") ? atoi(getenv("GESPMM_STAGED_ROWS")) : 0;
    int t;
    const int tc = staged_tile_cols(N, &t);
    StagedShape sh = {0, 0, 0};
    if (!tc) return sh;
    sh.waves = (waves_env == 4 || waves_env == 8 || waves_env == 16) ? waves_env : kStagedMaxWaves;
    sh.rows = (tc == 128 ? 6 : 4) * sh.waves;
    if (rows_env > 0) sh.rows = rows_env;
    sh.slots = sh.waves * kStagedLdsPerWave / (tc * 4);
    return sh;
}

// B beyond 4 GB: two 4 GB halves (tiled widths only), up to 8 GB.
bool staged_serves(int64_t K, int64_t N) {
    int t;
    if (!staged_tile_cols(N, &t)) return false;
    const uint64_t bytes = (uint64_t)K * (uint64_t)N * 4ull;
    return bytes < 0xFFFF0000ull || (t >= 1 && bytes < 0x1FFFF0000ull);
}

hipError_t launch_spmm_staged(const StagedArgs& a_in, int64_t K, int64_t N, hipStream_t st) {
    StagedArgs a = a_in;
    static const int dbg_env = (GESPMM_STAGED_INSTRUMENT && getenv("GESPMM_STAGED_DEBUG")) ? atoi(getenv("GESPMM_STAGED_DEBUG")) : 0;
    a.debug = dbg_env;
    static unsigned long long* dbg_buf = nullptr;
    static size_t dbg_cap = 0;
    const size_t dbg_need = (size_t)a.nblocks * (size_t)a.waves * 4;
    if ((dbg_env & 4) && dbg_cap < dbg_need) {
        if (dbg_buf) (void)hipFree(dbg_buf);
        if (hipMalloc(reinterpret_cast<void**>(&dbg_buf), dbg_need * 8) != hipSuccess) return hipErrorOutOfMemory;
        dbg_cap = dbg_need;
    }
    if (dbg_env & 4) (void)hipMemsetAsync(dbg_buf, 0, dbg_need * 8, st);
    a.dbg_clk = dbg_buf;
    if (a.nblocks <= 0) return hipSuccess;
    if (!staged_serves(K, N)) return hipErrorInvalidValue;
    int t;
    const int tc = staged_tile_cols(N, &t);
    const bool paged = (uint64_t)K * (uint64_t)N * 4ull >= 0xFFFF0000ull;
    const dim3 grid((unsigned)a.nblocks << (t > 0 ? t : 0)), block((unsigned)a.waves * 64);
    static const int u_env = getenv("GESPMM_STAGED_U") ? atoi(getenv("GESPMM_STAGED_U")) : 0;  // experiment knob: entries in flight per wavefront
#define GESPMM_STAGED_LAUNCH(VEC, U, TS, PG)                                                                                   \
    do {                                                                                                                       \
        if (a.waves == 16) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 16>), grid, block, 0, st, a);                \
        else if (a.waves == 8) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 8>), grid, block, 0, st, a);             \
        else if (a.waves == 4) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 4>), grid, block, 0, st, a);             \
        else return hipErrorInvalidValue;                                                                                      \
    } while (0)
    if (tc == 128 && u_env == 16) GESPMM_STAGED_LAUNCH(2, 16, 0, false);
    else if (tc == 128) GESPMM_STAGED_LAUNCH(2, 8, 0, false);
    else if (tc == 256 && t == 0) GESPMM_STAGED_LAUNCH(4, 8, 0, false);
    else if (tc == 256 && t == 1 && !paged) GESPMM_STAGED_LAUNCH(4, 8, 1, false);
    else if (tc == 256 && t == 1 && paged) GESPMM_STAGED_LAUNCH(4, 8, 1, true);
    else if (tc == 256 && t == 2 && !paged) GESPMM_STAGED_LAUNCH(4, 8, 2, false);
    else if (tc == 256 && t == 2 && paged) GESPMM_STAGED_LAUNCH(4, 8, 2, true);
    else
        return hipErrorInvalidValue;
#undef GESPMM_STAGED_LAUNCH
    if (dbg_env & 4) {
        static int printed = 0;
        (void)hipStreamSynchronize(st);
        if (printed++ < 6 && t <= 0) {
            const size_t nw = (size_t)a.nblocks * (size_t)a.waves;
            unsigned long long* h = (unsigned long long*)malloc(nw * 32);
            (void)hipMemcpy(h, dbg_buf, nw * 32, hipMemcpyDeviceToHost);
            double sum[3] = {0, 0, 0};
            unsigned long long t0 = ~0ull, t1 = 0;
            size_t n = 0;
            for (size_t i = 0; i < nw; ++i) {
                if (!h[4 * i + 3]) continue;
                ++n;
                for (int q = 0; q < 3; ++q) sum[q] += (double)h[4 * i + q];
                if (h[4 * i + 3] < t0) t0 = h[4 * i + 3];
                const unsigned long long e = h[4 * i + 3] + h[4 * i] + h[4 * i + 1] + h[4 * i + 2];
                if (e > t1) t1 = e;
            }
            if (n)
                fprintf(stderr, "[staged clk] %zu wavefronts: to-staged %.0f walk %.0f drain %.0f cycles per wavefront; first start to last end %llu cycles\n",
                        n, sum[0] / n, sum[1] / n, sum[2] / n, t1 - t0);
            free(h);
        }
    }
    return hipGetLastError();
}

}  // namespace gespmm
