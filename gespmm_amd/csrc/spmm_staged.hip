// spmm_staged.hip — the plan's kernel for clustered matrices with rows long enough to share neighbours inside a block:
// scalar-branch walk over a record stream + the block's most used B rows staged in LDS (round 3; the record stream: round 5).
//
// What bounds the streaming kernels on such a matrix (products-shaped communities, N = 128: 75 % of the gathers hit the L2
// and the product still takes 3.9 ms = 16 TB/s of gathers) is the vector memory path: every wave-level load instruction costs
// the address unit the same ~16 cycles whether it carries 256 bytes or 1 KB, whether it hits or misses — moving hits from L2
// to LDS through masked, flat or out-of-range loads changes nothing (profiles/r03/hotrows_experiment.log). The only gather that
// is free for the address unit is one that is never issued. Hence:
//
//   * ONE ROW PER WAVEFRONT AT A TIME: a B row of N = 64 * VEC floats is one load of VEC dwords per lane. Everything that is the
//     same for the 64 lanes — a record's code and value — is read out of the wavefront's window of the stream into SGPR pairs;
//     "is this entry's B row staged?" is a SCALAR branch around one of {ds_read, global_load}: a staged entry issues no
//     vector memory instruction at all;
//   * a workgroup of 16 wavefronts owns a BLOCK of 96 (N = 128) / 64 (N = 256) consecutive rows of the plan's clustered matrix; the analysis
//     (plan_device.hip: device_build_staging) lists per block the <= H columns used most often inside it (>= 2 uses; H rows =
//     64 KB) and rewrites the block's entries: bit 31 of the code clear = column, set = slot of the staged row. The workgroup copies
//     the listed rows into LDS once, coalesced, then each wavefront walks its share of the block's rows as one stream;
//   * THE RECORD STREAM (round 5): per row its entries {code, value} and then ONE row-end record {kStagedRowEnd, C row} — rows
//     without entries have theirs too. A wavefront needs nothing but a range of that stream: no row pointers, no row ids, no
//     position compares, no row cursor. 64 records at a time sit in a register pair (the next window is requested a window ahead,
//     through the vector path: a scalar load would share its counter with the LDS reads); one ballot per window and kind gives
//     the masks "B row from memory" and "row end", and a chunk of 8 records tests mask BITS: all staged -> 8 LDS reads without a
//     branch; no row end -> 8 packed multiply-add groups and nothing else. Round 4's walk compared every entry's position with its
//     row's end and the task's end: 6.9 scalar + 2.9 branch + 6.4 vector instructions per entry on rows of 12 entries
//     (profiles/r05/staged_issue_counters.log), now 5.2 + 2.1 + 5.3 per record;
//   * two multiply-adds per instruction: v_pk_fma_f32 on accumulator PAIRS with the value broadcast from the record's SGPR pair
//     (op_sel picks its high dword for both halves) — each half is a fused multiply-add rounded like v_fma_f32.
//
// Every output element is still ONE fp32 chain over the row's entries in CSR order with one fused multiply-add per entry
// (spmm_test.cu:182-203 semantics; unweighted matrices carry 1.0f: fma(1, b, acc) == acc + b exactly), so the bits are those
// of every other variant. Sum reducer; N = 128, 256 and — as 256-column tiles bound to XCDs — 512 / 1024; B and C below 4 GB
// (32-bit lane offsets from an SGPR base; the tiled widths reach 8 GB through two bases); everything else stays on the streaming
// kernels. Rows of more than kStagedMaxRow entries never reach this kernel: the plan empties them in the row pointers the tables are
// built from (their row-end record stores zeros) and runs them through the streaming kernel's long-row pass afterwards
// (plan.cpp: plan_run).
//
// The gathers and the row-end chunks are inline assembly: written as C++ the compiler puts `s_waitcnt vmcnt(0)` in front of every
// LDS read (it cannot see that the two paths never write the same register in the same pass), and its structurizer turns "bit set ->
// store and zero, else multiply-add" into ~6 scalar instructions and two register copies per record.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "spmm_device.h"
#include "spmm_kernels.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
// (80 KB of static LDS per workgroup — gfx942 stops at 64 KB — and inline assembly that spells sc1 / nt modifiers and SGPR-base global loads)
#error "spmm_staged.hip is written for gfx950: its LDS shapes and inline assembly do not build elsewhere"
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
using v2f = float __attribute__((ext_vector_type(2)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

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
// ACC (round 6, column-slab tables: plan.cpp build_slab_tables): the launch CONTINUES rows whose first column slabs an earlier launch
// summed — a wavefront's accumulators start as the C row of its first row (task word 0), and a row-end record carries the C row of the
// row behind it in its code's low 30 bits: store this row, load that one. A partial sum stored to C and loaded again is the same value,
// and a row's entries are still added in CSR order (the slabs are ascending column ranges of rows with ascending columns): same bits.
template <int VEC, int U, int TSHIFT, bool PAGE2, int WAVES, int LK, bool ACC = false>
__global__ __launch_bounds__(WAVES * 64) void spmm_staged_kernel(StagedArgs a) {
    static_assert(!ACC || (VEC == 2 && TSHIFT == 0 && !PAGE2), "the continuing form exists for the 128-column shape");
    constexpr int kStagedWaves = WAVES;
    constexpr int kStagedLdsBytes = WAVES * LK * 1024;  // LK KB of staged B rows per wavefront of the block (4: two blocks per CU; 8: one)
    constexpr int P = LK;                                // 16-byte pieces of the staging copy per thread
    using vec_t = typename LaneVec<VEC>::type;
    constexpr int kRowBytes = 256 * VEC;          // bytes of a row inside one tile
    constexpr int kRowShift = (VEC == 2) ? 9 : 10;
    constexpr int kGlobalShift = kRowShift + TSHIFT;  // log2(N * 4): row stride of B and C
    constexpr int H = kStagedLdsBytes / kRowBytes;  // staged rows per block
    constexpr int kRowF4 = kRowBytes / 16;
    constexpr int kWin = 64;                        // entries of the stream one wavefront holds in a register pair
    static_assert(kStagedPad >= kWin, "a window is read whole: up to kWin - 1 entries past a task's end");
    static_assert(kWin % U == 0 && (U == 8 || U == 16), "whole chunks per window, whole groups of four records per chunk");
    static_assert(TSHIFT >= 0 && TSHIFT <= 3, "at most one tile per XCD");
    __shared__ f4v s_hot[H * kRowF4];

    if (a.guard != nullptr && *a.guard != a.guard_want) return;  // (guarded launch: spmm_kernels.h — the whole grid, before any barrier)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, tile = 0;
    if constexpr (TSHIFT == 0) {
        blk = a.blk0 + xcd_contiguous(blockIdx.x, a.nblocks);  // (blk0: the slab's first block — 0 for every other table)
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
    // columns (vector; thread t copies the 16-byte pieces t, t + T, ... (P of them) of the H x row-bytes array, T = threads per block).
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int32_t* hc = a.hot_cols + (size_t)blk * H;
    static_assert(H * kRowF4 == P * kStagedWaves * 64, "P pieces per thread");
    int hcol[P];
#pragma unroll
    for (int u = 0; u < P; ++u) hcol[u] = (dbg & 1) ? -1 : hc[(u * kStagedWaves * 64 + tid) / kRowF4];
    const int wb = tk[2], we = tk[3];  // the wavefront's range of the record stream (entries + one row-end record per row)
    const int crow0 = ACC ? tk[0] : 0;  // (slab tables: C row of the task's first row)
    const float* Bp = a.B + (size_t)tile * (64 * VEC);
    const float* BpHi = Bp + (1ull << 30);  // + 4 GB (PAGE2)
    (void)BpHi;
    const uint32_t loff = (uint32_t)lane * (4u * VEC);

    // Round trip 2 — the staged rows (slots the block does not use hold -1: no load) and the first window of the wavefront's
    // stream, through the VECTOR path. Nothing on the walk below is a scalar MEMORY load — see the header: a scalar load shares its
    // counter with the LDS reads and is waited for with every chunk. Row ends and C rows arrive WITH the stream (row-end records).
    const f4v* B4 = reinterpret_cast<const f4v*>(a.B);
    f4v stage[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int i = u * kStagedWaves * 64 + tid;
        stage[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (hcol[u] >= 0) stage[u] = B4[(((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4)];
    }
    const i2v* evv = reinterpret_cast<const i2v*>(a.ev) + wb;
    i2v win = {0, 0};
    if (we > wb) win = __builtin_nontemporal_load(evv + lane);  // (the stream is padded: a whole window is always readable)
    v2f acc_first = {0.0f, 0.0f};
    if constexpr (ACC) {
        if (we > wb)
            acc_first = *reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(a.C) + (((size_t)(uint32_t)crow0) << kGlobalShift) + loff);
    }
#pragma unroll
    for (int u = 0; u < P; ++u)
        if (hcol[u] >= 0) s_hot[u * kStagedWaves * 64 + tid] = stage[u];
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start
    uint64_t t_staged = 0;
    if (dbg & 4) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        t_staged = __builtin_readcyclecounter();
    }
    if (we <= wb) return;  // (a task without rows)
    // one offset serves both paths (LDS address of a staged row / byte offset into B): the staging array must sit at LDS address 0
    // (it is the kernel's only LDS object; a compile-time constant — the check folds away)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot != 0u) __builtin_trap();
    // Accumulators as PAIRS: one v_pk_fma_f32 per two columns, the value broadcast from the record's SGPR pair {code, value} — both
    // halves take the pair's HIGH dword (op_sel:[1,0,0] op_sel_hi:[1,1,1]). Two fused multiply-adds per instruction, each rounded
    // like v_fma_f32: the bits of every other variant.
    // (256-column tiles: the four accumulators are pinned to v[60:63] — inline assembly cannot name the halves of a register
    //  quadruple, and a row end wants ONE 16-byte store per lane: two 8-byte stores write every line of C in two halves and cost
    //  the short-row graphs a third of their time, profiles/r05/kernel_ab_record_stream.log)
    v2f acc[1] = {acc_first};
    f4v acc4 = {0.0f, 0.0f, 0.0f, 0.0f};
    (void)acc;
    (void)acc4;
    // A row-end record's value word is the C row of the row that ends there (rows without entries have one too: zeros are stored).
    // C rows are addressed like B rows: SGPR base (+ 4 GB for the upper half, PAGE2) and a 32-bit lane offset `crow << log2(row bytes)`.
    float* const Cp = a.C + (size_t)tile * (64 * VEC);
    float* const CpHi = Cp + (1ull << 30);
    (void)CpHi;
    // code: bit 31 = staged (low bits: LDS slot), else the column; kStagedRowEnd (bit 30) = row-end record. Both address shifts drop
    // bits 30 / 31, so a row-end record reads slot 0 in a chunk that gathers from LDS only (harmless, never summed).
    // `mem` = this chunk's share of the window's memory mask, J = the entry's bit in it: a SCALAR bit test decides the path.
    auto gather = [&](int code, uint32_t mem, auto J, vec_t& d) {
        constexpr int kJ = decltype(J)::value;
        const float* const bp = Bp;        // (named here: a generic lambda does not capture through an asm operand)
        const float* const bp_hi = BpHi;
        (void)bp_hi;
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
                "s_bitcmp0_b32 %3, %8\n\t"
                "s_cbranch_scc1 1f\n\t"
                "s_bitcmp1_b32 %9, %7\n\t"
                "s_cselect_b64 %1, %6, %4\n\t"
                "global_load_dwordx4 %0, %2, %1\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b128 %0, %5\n"
                "2:"
                : "=&v"(d), "=&s"(base)
                : "v"(voff), "s"(mem), "s"(bp), "v"(voff_l), "s"(bp_hi), "n"(kPageBit), "n"(kJ), "s"(code)
                : "memory", "scc");
        } else if constexpr (VEC == 2)
            asm volatile(
                "s_bitcmp0_b32 %2, %5\n\t"
                "s_cbranch_scc1 1f\n\t"
                "global_load_dwordx2 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b64 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(mem), "s"(bp), "v"(voff_l), "n"(kJ)
                : "memory", "scc");
        else
            asm volatile(
                "s_bitcmp0_b32 %2, %5\n\t"
                "s_cbranch_scc1 1f\n\t"
                "global_load_dwordx4 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b128 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(mem), "s"(bp), "v"(voff_l), "n"(kJ)
                : "memory", "scc");
    };
    auto gather_lds = [&](int code, vec_t& d) {  // a chunk whose entries are all staged: no branch, no vector memory
        const uint32_t voff_l =
            ((uint32_t)code << kRowShift) + loff + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;
        if constexpr (VEC == 2) asm volatile("ds_read_b64 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
    };
    auto fma_row = [&](uint64_t cv, const vec_t& b) {
        if constexpr (VEC == 2) {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[0]) : "s"(cv), "v"(b));
        } else {
            const v2f blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
            asm("v_pk_fma_f32 v[60:61], %1, %2, v[60:61] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                "v_pk_fma_f32 v[62:63], %1, %3, v[62:63] op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                : "+{v[60:63]}"(acc4)
                : "s"(cv), "v"(blo), "v"(bhi));
        }
    };
    // Four records of a chunk that holds a row end, as ONE block of assembly: per record a scalar bit test of the chunk's row-end
    // mask and a branch that is not taken for an entry (-> its multiply-adds, fall through to the next record); the row-end code sits
    // behind the four (store the accumulators to C row `value word`, zero them, jump back). Written out because the compiler's
    // structurizer turns the same C++ into ~6 scalar instructions and two register copies per record.
#define GESPMM_FMA2(A, C, B) "v_pk_fma_f32 " A ", " C ", " B ", " A " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
    auto consume4 = [&](uint32_t ends, const uint64_t* cv, const vec_t* b) {
        uint32_t t;
        const uint32_t r0 = (uint32_t)(cv[0] >> 32), r1 = (uint32_t)(cv[1] >> 32), r2 = (uint32_t)(cv[2] >> 32), r3 = (uint32_t)(cv[3] >> 32);
        if constexpr (VEC == 2 && ACC) {
            const uint32_t n0 = (uint32_t)cv[0] & 0x3fffffffu, n1 = (uint32_t)cv[1] & 0x3fffffffu, n2 = (uint32_t)cv[2] & 0x3fffffffu,
                           n3 = (uint32_t)cv[3] & 0x3fffffffu;  // C row of the row behind each (possible) row end
            // (the store reads its 8 bytes of data at issue; the load's result arrives long after — and is waited for before the next multiply-add)
#define GESPMM_END2A(R, NX)                                                                                                       \
    "v_lshl_add_u32 %[t], " R ", %[sh], %[lo]\n\tglobal_store_dwordx2 %[t], %[a], %[C] sc1 nt\n\tv_lshl_add_u32 %[t], " NX ", %[sh], %[lo]\n\t" \
    "global_load_dwordx2 %[a], %[t], %[C]\n\ts_waitcnt vmcnt(0)\n\t"
            asm volatile(
                "s_bitcmp1_b32 %[e], 0\n\ts_cbranch_scc1 10f\n\t" GESPMM_FMA2("%[a]", "%[c0]", "%[b0]") "\n11:\n\t"
                "s_bitcmp1_b32 %[e], 1\n\ts_cbranch_scc1 20f\n\t" GESPMM_FMA2("%[a]", "%[c1]", "%[b1]") "\n21:\n\t"
                "s_bitcmp1_b32 %[e], 2\n\ts_cbranch_scc1 30f\n\t" GESPMM_FMA2("%[a]", "%[c2]", "%[b2]") "\n31:\n\t"
                "s_bitcmp1_b32 %[e], 3\n\ts_cbranch_scc1 40f\n\t" GESPMM_FMA2("%[a]", "%[c3]", "%[b3]")
                "s_branch 99f\n"
                "10:\n\t" GESPMM_END2A("%[r0]", "%[n0]") "s_branch 11b\n"
                "20:\n\t" GESPMM_END2A("%[r1]", "%[n1]") "s_branch 21b\n"
                "30:\n\t" GESPMM_END2A("%[r2]", "%[n2]") "s_branch 31b\n"
                "40:\n\t" GESPMM_END2A("%[r3]", "%[n3]") "\n99:"
                : [a] "+v"(acc[0]), [t] "=&v"(t)
                : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),
                  [b3] "v"(b[3]), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [n0] "s"(n0), [n1] "s"(n1), [n2] "s"(n2), [n3] "s"(n3),
                  [sh] "n"(kGlobalShift), [lo] "v"(loff), [C] "s"(Cp)
                : "memory", "scc");
#undef GESPMM_END2A
        } else if constexpr (VEC == 2) {
#define GESPMM_END2(R) "v_lshl_add_u32 %[t], " R ", %[sh], %[lo]\n\tglobal_store_dwordx2 %[t], %[a], %[C] sc1 nt\n\tv_mov_b64 %[a], 0\n\t"
            asm volatile(
                "s_bitcmp1_b32 %[e], 0\n\ts_cbranch_scc1 10f\n\t" GESPMM_FMA2("%[a]", "%[c0]", "%[b0]") "\n11:\n\t"
                "s_bitcmp1_b32 %[e], 1\n\ts_cbranch_scc1 20f\n\t" GESPMM_FMA2("%[a]", "%[c1]", "%[b1]") "\n21:\n\t"
                "s_bitcmp1_b32 %[e], 2\n\ts_cbranch_scc1 30f\n\t" GESPMM_FMA2("%[a]", "%[c2]", "%[b2]") "\n31:\n\t"
                "s_bitcmp1_b32 %[e], 3\n\ts_cbranch_scc1 40f\n\t" GESPMM_FMA2("%[a]", "%[c3]", "%[b3]")
                "s_branch 99f\n"
                "10:\n\t" GESPMM_END2("%[r0]") "s_branch 11b\n"
                "20:\n\t" GESPMM_END2("%[r1]") "s_branch 21b\n"
                "30:\n\t" GESPMM_END2("%[r2]") "s_branch 31b\n"
                "40:\n\t" GESPMM_END2("%[r3]") "\n99:"
                : [a] "+v"(acc[0]), [t] "=&v"(t)
                : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),
                  [b3] "v"(b[3]), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [sh] "n"(kGlobalShift), [lo] "v"(loff), [C] "s"(Cp)
                : "memory", "scc");
#undef GESPMM_END2
        } else {
            v2f bl[4], bh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bl[j] = __builtin_shufflevector(b[j], b[j], 0, 1);
                bh[j] = __builtin_shufflevector(b[j], b[j], 2, 3);
            }
            // (one 16-byte store per lane; TWO wait states before the stored registers are overwritten — a store wider than 8 bytes reads its data late, gfx940+: a single one let the zeros through now and then, cora at N = 1024)
#define GESPMM_STORE4(BASE) "global_store_dwordx4 %[t], v[60:63], " BASE " sc1 nt\n\ts_nop 1\n\tv_mov_b64 v[60:61], 0\n\tv_mov_b64 v[62:63], 0\n\t"
#define GESPMM_ROWS4(END)                                                                                                                  \
    "s_bitcmp1_b32 %[e], 0\n\ts_cbranch_scc1 10f\n\t" GESPMM_FMA2("v[60:61]", "%[c0]", "%[b0]") GESPMM_FMA2("v[62:63]", "%[c0]", "%[h0]") "\n11:\n\t" \
    "s_bitcmp1_b32 %[e], 1\n\ts_cbranch_scc1 20f\n\t" GESPMM_FMA2("v[60:61]", "%[c1]", "%[b1]") GESPMM_FMA2("v[62:63]", "%[c1]", "%[h1]") "\n21:\n\t" \
    "s_bitcmp1_b32 %[e], 2\n\ts_cbranch_scc1 30f\n\t" GESPMM_FMA2("v[60:61]", "%[c2]", "%[b2]") GESPMM_FMA2("v[62:63]", "%[c2]", "%[h2]") "\n31:\n\t" \
    "s_bitcmp1_b32 %[e], 3\n\ts_cbranch_scc1 40f\n\t" GESPMM_FMA2("v[60:61]", "%[c3]", "%[b3]") GESPMM_FMA2("v[62:63]", "%[c3]", "%[h3]")              \
    "s_branch 99f\n"                                                                                                                      \
    "10:\n\t" END("%[r0]") "s_branch 11b\n"                                                                                              \
    "20:\n\t" END("%[r1]") "s_branch 21b\n"                                                                                              \
    "30:\n\t" END("%[r2]") "s_branch 31b\n"                                                                                              \
    "40:\n\t" END("%[r3]") "\n99:"
            if constexpr (!PAGE2) {
#define GESPMM_END4(R) "v_lshl_add_u32 %[t], " R ", %[sh], %[lo]\n\t" GESPMM_STORE4("%[C]")
                asm volatile(GESPMM_ROWS4(GESPMM_END4)
                             : [a4] "+{v[60:63]}"(acc4), [t] "=&v"(t)
                             : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(bl[0]), [b1] "v"(bl[1]),
                               [b2] "v"(bl[2]), [b3] "v"(bl[3]), [h0] "v"(bh[0]), [h1] "v"(bh[1]), [h2] "v"(bh[2]), [h3] "v"(bh[3]), [r0] "s"(r0),
                               [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [sh] "n"(kGlobalShift), [lo] "v"(loff), [C] "s"(Cp)
                             : "memory", "scc");
#undef GESPMM_END4
            } else {
                constexpr int kPageBitC = 32 - kGlobalShift;  // bit of the C row that selects the 4 GB half
                uint64_t base;
#define GESPMM_END4P(R) "s_bitcmp1_b32 " R ", %[pg]\n\ts_cselect_b64 %[cb], %[Chi], %[C]\n\tv_lshl_add_u32 %[t], " R ", %[sh], %[lo]\n\t" GESPMM_STORE4("%[cb]")
                asm volatile(GESPMM_ROWS4(GESPMM_END4P)
                             : [a4] "+{v[60:63]}"(acc4), [t] "=&v"(t), [cb] "=&s"(base)
                             : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(bl[0]), [b1] "v"(bl[1]),
                               [b2] "v"(bl[2]), [b3] "v"(bl[3]), [h0] "v"(bh[0]), [h1] "v"(bh[1]), [h2] "v"(bh[2]), [h3] "v"(bh[3]), [r0] "s"(r0),
                               [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [sh] "n"(kGlobalShift), [lo] "v"(loff), [C] "s"(Cp), [Chi] "s"(CpHi),
                               [pg] "n"(kPageBitC)
                             : "memory", "scc");
#undef GESPMM_END4P
            }
#undef GESPMM_ROWS4
#undef GESPMM_STORE4
        }
    };
#undef GESPMM_FMA2
    // The walk (round 5). One ballot per window and kind turns the codes into two masks — records whose B row comes from memory,
    // row-end records — and a chunk of U records tests mask BITS: a chunk without a row end is U multiply-add groups and nothing else,
    // a chunk that gathers from LDS only has no branch on its gathers. The walk used to compare every entry's position with its row's
    // end and the task's end and to step a row cursor (row pointers and C rows in two more registers): 6.9 scalar + 2.9 branch
    // instructions per entry on a graph with rows of 12 entries, and the scalar unit — one instruction per clock per CU — was the
    // busiest unit of the kernel (profiles/r05/staged_issue_counters.log). A task's last record is a row end, so records behind `we`
    // (the next task's, or the padding: gathered, harmlessly) only ever reach an accumulator that is never stored.
    for (int kw = wb; kw < we; kw += kWin) {
        // the next window is requested a whole window (kWin / U chunks) before it is needed; only chunks that gather from memory
        // themselves wait for the vector memory counter, and those wait for their own (younger) loads anyway
        i2v nxt = win;
        if (kw + kWin < we) nxt = __builtin_nontemporal_load(evv + (kw - wb) + kWin + lane);
        const uint64_t gmask = (dbg & 2) ? 0ull : __ballot((uint32_t)win.x < (uint32_t)kStagedRowEnd);  // B row from memory
        uint64_t lmask = __ballot((win.x & kStagedRowEnd) != 0 && win.x >= 0);                          // row-end records ...
        if (we - kw < kWin) lmask &= (1ull << (we - kw)) - 1ull;                                          // ... of THIS task
#pragma unroll 1
        for (int c = 0; c < kWin; c += U) {
            if (kw + c >= we) break;
            uint64_t cv[U];  // {code, value bits} of the chunk's records: SGPR pairs
            int code[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                code[j] = __builtin_amdgcn_readlane(win.x, c + j);
                cv[j] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(win.y, c + j) << 32) | (uint32_t)code[j];
            }
            vec_t bv[U];
            const uint32_t anymem = (uint32_t)(gmask >> c) & ((1u << U) - 1u);
            if (anymem == 0) {
#pragma unroll
                for (int j = 0; j < U; ++j) gather_lds(code[j], bv[j]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                static_for<U>([&](auto J) { gather(code[decltype(J)::value], anymem, J, bv[decltype(J)::value]); });
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < U; ++j) asm volatile("" : "+v"(bv[j]));  // (uses of bv stay behind the wait)
            const uint32_t ends = (uint32_t)(lmask >> c) & ((1u << U) - 1u);
            if (ends == 0) {
#pragma unroll
                for (int j = 0; j < U; ++j) fma_row(cv[j], bv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; j += 4) consume4(ends >> j, cv + j, bv + j);
            }
        }
        win = nxt;
    }
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
    // LDS per wavefront: 5 KB — two 16-wavefront blocks of 80 KB fill the CU's 160 KB exactly, 160 / 80 staged rows per block instead of
    // 128 / 64 (profiles/r05/staged_lds5.log, 4 -> 5 KB at N = 128 / 256: geometric 202.6 -> 187.6 / 341.6 -> 326.3 us, small-world 350 ->
    // 329 / 618 -> 559, LFR 170 -> 168 / 338 -> 327, products-shaped 2.81 -> 2.77 / 5.27 -> 5.10 ms, com-Amazon-shaped level). 8 KB — ONE
    // block per CU — lost 20-40 % (staged_lds_per_wave.log). GESPMM_STAGED_LDS_KB=4 brings the 64 KB blocks back; smaller blocks keep 4.
    static const int lds_env = getenv("GESPMM_STAGED_LDS_KB") ? atoi(getenv("GESPMM_STAGED_LDS_KB")) : 0;
    const int lds_kb = (sh.waves == 16 && lds_env != 4) ? 5 : 4;
    sh.rows = (tc == 128 ? 6 : 4) * sh.waves;
    if (rows_env > 0) sh.rows = rows_env;
    sh.slots = sh.waves * lds_kb * 1024 / (tc * 4);
    return sh;
}

// B and C rows are addressed by a 32-bit lane offset from an SGPR base: the larger of the two matrices must stay below 4 GB — or, for the
// tiled widths, below 8 GB (two 4 GB halves, the base chosen by one bit of the row index).
bool staged_serves(int64_t M, int64_t K, int64_t N) {
    int t;
    if (!staged_tile_cols(N, &t)) return false;
    const uint64_t bytes = (uint64_t)(M > K ? M : K) * (uint64_t)N * 4ull;
    return bytes < 0xFFFF0000ull || (t >= 1 && bytes < 0x1FFFF0000ull);
}

hipError_t launch_spmm_staged(const StagedArgs& a_in, int64_t M, int64_t K, int64_t N, hipStream_t st) {
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
    if (!staged_serves(M, K, N)) return hipErrorInvalidValue;
    int t;
    const int tc = staged_tile_cols(N, &t);
    const bool paged = (uint64_t)(M > K ? M : K) * (uint64_t)N * 4ull >= 0xFFFF0000ull;
    const dim3 grid((unsigned)a.nblocks << (t > 0 ? t : 0)), block((unsigned)a.waves * 64);
    static const int u_env = getenv("GESPMM_STAGED_U") ? atoi(getenv("GESPMM_STAGED_U")) : 0;  // experiment knob: entries in flight per wavefront
    const int lds_kb = a.slots > 0 ? (int)((int64_t)a.slots * tc * 4 / ((int64_t)a.waves * 1024)) : 4;  // what the tables were built for
    // (8 KB per wavefront — one 16-wavefront block per CU with twice the staged rows — was built and measured: 20-40 % slower on every
    //  graph, products-shaped 3.92 vs 2.79 ms, geometric 240 vs 193 us: two blocks per CU hide each other's staging round trips, one does
    //  not. profiles/r05/staged_lds_per_wave.log; the kernel stays generic in LK, only 4 is instantiated)
    if (lds_kb != 4 && !(lds_kb == 5 && a.waves == 16) && !(lds_kb == 3 && a.waves == 16)) return hipErrorInvalidValue;
    if (lds_kb == 3) {  // (column-slab tables built for THREE 48 KB blocks per CU: 96 staged rows, 8 gathers per chunk — 33 VGPRs)
        if (!(tc == 128 && a.waves == 16)) return hipErrorInvalidValue;
        if (a.acc) hipLaunchKernelGGL((spmm_staged_kernel<2, 8, 0, false, 16, 3, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((spmm_staged_kernel<2, 8, 0, false, 16, 3, false>), grid, block, 0, st, a);
        return hipGetLastError();
    }
    if (a.acc) {  // (column-slab tables, second and later slabs: the 128-column shape only)
        if (!(tc == 128 && a.waves == 16 && lds_kb == 5)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((spmm_staged_kernel<2, 16, 0, false, 16, 5, true>), grid, block, 0, st, a);
        return hipGetLastError();
    }
#define GESPMM_STAGED_LAUNCH(VEC, U, TS, PG)                                                                                   \
    do {                                                                                                                       \
        if (a.waves == 16 && lds_kb == 5) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 16, 5>), grid, block, 0, st, a);   \
        else if (a.waves == 16) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 16, 4>), grid, block, 0, st, a);   \
        else if (a.waves == 8) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 8, 4>), grid, block, 0, st, a);          \
        else if (a.waves == 4) hipLaunchKernelGGL((spmm_staged_kernel<VEC, U, TS, PG, 4, 4>), grid, block, 0, st, a);          \
        else return hipErrorInvalidValue;                                                                                      \
    } while (0)
    // 128 columns: 16 gathers per chunk (57 registers: still eight wavefronts per SIMD). With the retuned blocks, interleaved three times
    // (profiles/r05/staged_u16_lds5.log, 8 -> 16): products-shaped 2.80 -> 2.69 ms, LFR 169 -> 164 us, com-Amazon-shaped 91.2 -> 89.4,
    // small-world 336 -> 330, geometric 188 -> 190; with round 5's first blocks it was level (staged_gathers_per_chunk.log). The
    // 256-column tiles hold four registers per gathered row: 16 rows would not fit 64 registers.
    if (tc == 128 && u_env != 8) GESPMM_STAGED_LAUNCH(2, 16, 0, false);
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
