// spmm_staged_gen.hip — the staged-rows kernel of spmm_staged.hip for ANY width and for the max reducer (round 6).
//
// spmm_staged.hip serves N = 128 and 256 * 2^t: there the row stride of B and C is a shift and every lane of the wavefront owns live
// columns. Real feature widths are not like that — 41 / 47 classes, 100 / 200 / 602 features — and the reference's
// kernels take any N >= 1 (the `nout` guards of spmm_test.cu:206-233, spmm_kernel.cu:186-206), its DGL patch has a max twin
// (binary_reduce_max.cu:26-168). This file is the same design — a block's most used B rows staged in LDS, one row per wavefront at a
// time, the record stream of the plan read into SGPR pairs, scalar branches around {ds_read, global_load} — with what the
// general case needs:
//
//   * a lane owns VEC in {1, 2, 4} contiguous columns (the widest vector that divides N with the fewest column tiles: N = 100 -> 2,
//     N = 200 -> 4, N = 602 -> 2 x 5 tiles, odd N -> 1), a tile is 64 * VEC columns; lanes past N in the last tile gather the tile's
//     first columns (valid memory, summed into accumulators nobody stores) and are masked out of the row-end store through EXEC —
//     the reference's guard, without a compare per entry;
//   * the row stride N * 4 is not a power of two: the byte offset of a B row is one scalar multiply on the record's code (the code
//     is an SGPR), the C row of a row end likewise; LDS slots keep a power-of-two stride (the tile's width);
//   * column tiles of one block run back to back on the same XCD (the record stream and the staging list are re-read from its L2);
//   * RED = max (unweighted: binary_reduce_max.cu:18-24): the multiply-adds become v_max, a row end stores and re-arms the
//     accumulator with `empty`; that walk is plain C++ (the DGL max path is not the benchmarked one), the sum walk keeps the
//     hand-written row-end chunks of spmm_staged.hip.
//
// Same tables as spmm_staged.hip (plan_device.hip: device_build_staging) with H = 80 KB / (tile row bytes) slots per block, 16 tasks per
// block. One fp32 chain per output element in CSR order, one fused multiply-add per entry: the bits of every other variant. B and C
// below 4 GB (32-bit lane offsets from an SGPR base); hub rows (> kStagedMaxRow entries) go to the streaming kernel's long-row pass as
// for the other staged kernels (plan.cpp: plan_run).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "spmm_device.h"
#include "spmm_kernels.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "spmm_staged_gen.hip is written for gfx950: 80 KB of LDS per workgroup and its inline assembly do not build elsewhere"
#endif

namespace gespmm {

namespace {

typedef const __attribute__((address_space(4))) int32_t* cint_ptr;  // constant address space: scalar loads
using i2v = int __attribute__((ext_vector_type(2)));
using v2f = float __attribute__((ext_vector_type(2)));
using f4v = float __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for_g(F&& f) {
    if constexpr (N > 0) {
        static_for_g<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int VEC> struct GenVec;
template <> struct GenVec<1> { using type = float; };
template <> struct GenVec<2> { using type = v2f; };
template <> struct GenVec<4> { using type = f4v; };

constexpr int kGenWaves = kStagedMaxWaves;  // 16 wavefronts per block
constexpr int kGenLdsKb = 5;                // 5 KB of staged rows per wavefront: two blocks fill the CU's 160 KB (spmm_staged.hip)

template <int VEC, int RED>
__global__ __launch_bounds__(kGenWaves * 64, 8) void spmm_staged_gen_kernel(StagedArgs a) {  // (8 wavefronts per SIMD = two blocks per CU: <= 64 VGPRs)
    constexpr int TW = 64 * VEC;                          // columns of a tile
    constexpr int kSlotShift = VEC == 1 ? 8 : (VEC == 2 ? 9 : 10);  // log2(bytes of a staged row's part in this tile)
    constexpr int H = kGenWaves * kGenLdsKb * 1024 / (TW * 4);      // staged rows per block: 320 / 160 / 80
    constexpr int PS = H / kGenWaves;                     // slots each wavefront copies: 20 / 10 / 5
    constexpr int U = 8;                                  // records gathered together
    constexpr int kWin = 64;                              // records a wavefront holds in a register pair
    static_assert(kStagedPad >= kWin, "a window is read whole: up to kWin - 1 records past a task's end");
    static_assert(H % kGenWaves == 0, "whole slots per wavefront");
    using vec_t = typename GenVec<VEC>::type;
    __shared__ vec_t s_hot[H * 64];  // slot s, lane l: s_hot[s * 64 + l] (byte address s << kSlotShift | l * VEC * 4)

    if (a.guard != nullptr && *a.guard != a.guard_want) return;  // (guarded launch: spmm_kernels.h — the whole grid, before any barrier)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.n;
    const uint32_t n4 = (uint32_t)N * 4u;  // row stride of B and C in bytes
    // block and column tile of this workgroup: ids go to XCDs round-robin; XCD x sweeps a contiguous eighth of the blocks and runs the
    // tiles of a block back to back
    int blk, tile = 0;
    if (a.ntiles == 1) {
        blk = xcd_contiguous(blockIdx.x, a.nblocks);
    } else {
        const int x = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        tile = q % a.ntiles;
        const int bl = q / a.ntiles;
        const int q0 = a.nblocks >> 3, r = a.nblocks & 7;
        if (bl >= q0 + (x < r ? 1 : 0)) return;  // (the whole workgroup: the grid is padded to whole rounds of eight)
        blk = ((x < r) ? x * (q0 + 1) : r * (q0 + 1) + (x - r) * q0) + bl;
    }
    const int col0 = tile * TW;
    const int ncol = (N - col0 < TW) ? N - col0 : TW;  // live columns of this tile (a multiple of VEC)
    const int nact = ncol / VEC;                       // lanes that own live columns
    const bool active = lane < nact;
    const uint64_t am = nact >= 64 ? ~0ull : ((1ull << nact) - 1ull);  // their EXEC mask (row-end stores)
    const uint32_t loff_l = (uint32_t)lane * (4u * VEC);                                // the lane's bytes inside an LDS slot
    const uint32_t loff_g = ((uint32_t)col0 + (active ? (uint32_t)lane * VEC : 0u)) * 4u;  // ... inside a B / C row (lanes past N: the tile's first columns)

    // Round trip 1: the wavefront's task and the block's staged columns (wavefront w copies slots w, w + 16, ...: one row part per
    // coalesced access). Round trip 2: those rows and the first window of the record stream.
    const int task = blk * kGenWaves + wave;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    // (every lane loads its wavefront's slot ids itself — one address per wavefront and load, PS loads issued back to back; through the
    //  scalar path the compiler waits for each id before the row load it guards: PS dependent round trips in front of the barrier)
    const int32_t* hc = a.hot_cols + (size_t)blk * H + (tid >> 6);
    int hcol[PS];
#pragma unroll
    for (int u = 0; u < PS; ++u) hcol[u] = hc[u * kGenWaves];
    const int wb = tk[2], we = tk[3];
    const char* const Bc = reinterpret_cast<const char*>(a.B);
    // (slots the block does not use hold -1: they are filled with row 0 — one line the whole chip shares — and no code refers to them; a load
    //  under `if (hcol >= 0)` makes the compiler merge a zero with the loaded registers and wait for the load right there, and keeping the
    //  ids alive for a guarded LDS store spills at one float per lane: 20 ids + 20 offsets + 20 rows)
    vec_t stage[PS];
#pragma unroll
    for (int u = 0; u < PS; ++u) {
        const uint32_t hrow = hcol[u] < 0 ? 0u : (uint32_t)hcol[u];
        stage[u] = *reinterpret_cast<const vec_t*>(Bc + (size_t)(uint32_t)(hrow * n4 + loff_g));  // (B is below 4 GB: SGPR base + 32-bit lane offset)
    }
    const i2v* evv = reinterpret_cast<const i2v*>(a.ev) + wb;
    i2v win = {0, 0};
    if (we > wb) win = __builtin_nontemporal_load(evv + lane);  // (the stream is padded: a whole window is always readable)
#pragma unroll
    for (int u = 0; u < PS; ++u) s_hot[(u * kGenWaves + wave) * 64 + lane] = stage[u];  // (unused slots receive row 0: no code refers to them)
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start
    if (we <= wb) return;
    // one base serves both paths: the staging array must sit at LDS address 0 (the kernel's only LDS object; the check folds away)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) vec_t*)s_hot != 0u) __builtin_trap();
    const float* const Bp = a.B;
    float* const Cp = a.C;
    const float empty = a.empty;
    (void)empty;

    vec_t acc;  // (VEC == 4, sum: pinned to v[60:63] by the assembly that touches it — it names the halves)
    auto arm = [&]() {
        const float z = (RED == kReduceMax) ? empty : 0.0f;
        if constexpr (VEC == 1) acc = z;
        else if constexpr (VEC == 2) acc = v2f{z, z};
        else acc = f4v{z, z, z, z};
    };
    arm();

    // code: bit 31 = staged (low bits: LDS slot), else the column; kStagedRowEnd (bit 30) = row-end record (its shift drops the bit:
    // slot 0, read harmlessly, never summed). `mem` = the chunk's share of the window's memory mask: a SCALAR bit test picks the path.
    auto gather = [&](int code, uint32_t mem, auto J, vec_t& d) {
        constexpr int kJ = decltype(J)::value;
        const float* const bp = Bp;
        const uint32_t voff_l = ((uint32_t)code << kSlotShift) + loff_l + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) vec_t*)s_hot;
        const uint32_t voff = (uint32_t)code * n4 + loff_g;  // (the code is an SGPR: one s_mul_i32)
        if constexpr (VEC == 1)
            asm volatile(
                "s_bitcmp0_b32 %2, %5\n\t"
                "s_cbranch_scc1 1f\n\t"
                "global_load_dword %0, %1, %3\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b32 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(mem), "s"(bp), "v"(voff_l), "n"(kJ)
                : "memory", "scc");
        else if constexpr (VEC == 2)
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
    auto gather_lds = [&](int code, vec_t& d) {  // a chunk whose records are all staged: no branch, no vector memory
        const uint32_t voff_l = ((uint32_t)code << kSlotShift) + loff_l + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) vec_t*)s_hot;
        if constexpr (VEC == 1) asm volatile("ds_read_b32 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
        else if constexpr (VEC == 2) asm volatile("ds_read_b64 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
        else asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(voff_l) : "memory");
    };
    // one entry: acc = fma(value, b, acc) per column (sum; the value is the high dword of the record's SGPR pair) / acc = max(acc, b)
    auto fma_row = [&](uint64_t cv, const vec_t& b) {
        if constexpr (RED == kReduceMax) {
            if constexpr (VEC == 1) acc = fmaxf(acc, b);
            else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fmaxf(acc[i], b[i]);
            }
        } else if constexpr (VEC == 1) {
            const uint32_t vbits = (uint32_t)(cv >> 32);
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(vbits), "v"(b));
        } else if constexpr (VEC == 2) {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(cv), "v"(b));
        } else {
            const v2f blo = __builtin_shufflevector(b, b, 0, 1), bhi = __builtin_shufflevector(b, b, 2, 3);
            asm("v_pk_fma_f32 v[60:61], %1, %2, v[60:61] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                "v_pk_fma_f32 v[62:63], %1, %3, v[62:63] op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                : "+{v[60:63]}"(acc)
                : "s"(cv), "v"(blo), "v"(bhi));
        }
    };
    // Four records of a chunk that holds a row end (sum): per record a scalar bit test of the chunk's row-end mask and a branch that is
    // not taken for an entry; the row-end code sits behind the four: C row offset (premultiplied: an SGPR) + the lane's offset, the
    // store under the live lanes' EXEC mask, zero the accumulators, jump back. As in spmm_staged.hip, generalised in the address and
    // the mask.
#define GESPMM_G_FMA2(A, C, B) "v_pk_fma_f32 " A ", " C ", " B ", " A " op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
#define GESPMM_G_ROWS4(FMA0, FMA1, FMA2, FMA3, END)                                          \
    "s_bitcmp1_b32 %[e], 0\n\ts_cbranch_scc1 10f\n\t" FMA0 "\n11:\n\t"                      \
    "s_bitcmp1_b32 %[e], 1\n\ts_cbranch_scc1 20f\n\t" FMA1 "\n21:\n\t"                      \
    "s_bitcmp1_b32 %[e], 2\n\ts_cbranch_scc1 30f\n\t" FMA2 "\n31:\n\t"                      \
    "s_bitcmp1_b32 %[e], 3\n\ts_cbranch_scc1 40f\n\t" FMA3                                  \
    "s_branch 99f\n"                                                                        \
    "10:\n\t" END("%[r0]") "s_branch 11b\n"                                                 \
    "20:\n\t" END("%[r1]") "s_branch 21b\n"                                                 \
    "30:\n\t" END("%[r2]") "s_branch 31b\n"                                                 \
    "40:\n\t" END("%[r3]") "\n99:"
    auto consume4 = [&](uint32_t ends, const uint64_t* cv, const vec_t* b) {
        // (value word of a row-end record = its C row; times the row stride: four s_mul_i32 per group that holds a row end)
        const uint32_t h0 = (uint32_t)(cv[0] >> 32), h1 = (uint32_t)(cv[1] >> 32), h2 = (uint32_t)(cv[2] >> 32), h3 = (uint32_t)(cv[3] >> 32);
        const uint32_t r0 = h0 * n4, r1 = h1 * n4, r2 = h2 * n4, r3 = h3 * n4;
        uint32_t t;
        if constexpr (VEC == 1) {
#define GESPMM_G_END1(R) "v_add_u32 %[t], " R ", %[lo]\n\ts_mov_b64 exec, %[am]\n\tglobal_store_dword %[t], %[a], %[C] sc1 nt\n\ts_mov_b64 exec, -1\n\tv_mov_b32 %[a], 0\n\t"
            asm volatile(GESPMM_G_ROWS4("v_fma_f32 %[a], %[h0], %[b0], %[a]\n\t", "v_fma_f32 %[a], %[h1], %[b1], %[a]\n\t",
                                        "v_fma_f32 %[a], %[h2], %[b2], %[a]\n\t", "v_fma_f32 %[a], %[h3], %[b3], %[a]\n\t", GESPMM_G_END1)
                         : [a] "+v"(acc), [t] "=&v"(t)
                         : [e] "s"(ends), [h0] "s"(h0), [h1] "s"(h1), [h2] "s"(h2), [h3] "s"(h3), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),
                           [b3] "v"(b[3]), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [lo] "v"(loff_g), [C] "s"(Cp), [am] "s"(am)
                         : "memory", "scc");
#undef GESPMM_G_END1
        } else if constexpr (VEC == 2) {
#define GESPMM_G_END2(R) "v_add_u32 %[t], " R ", %[lo]\n\ts_mov_b64 exec, %[am]\n\tglobal_store_dwordx2 %[t], %[a], %[C] sc1 nt\n\ts_mov_b64 exec, -1\n\tv_mov_b64 %[a], 0\n\t"
            asm volatile(GESPMM_G_ROWS4(GESPMM_G_FMA2("%[a]", "%[c0]", "%[b0]"), GESPMM_G_FMA2("%[a]", "%[c1]", "%[b1]"),
                                        GESPMM_G_FMA2("%[a]", "%[c2]", "%[b2]"), GESPMM_G_FMA2("%[a]", "%[c3]", "%[b3]"), GESPMM_G_END2)
                         : [a] "+v"(acc), [t] "=&v"(t)
                         : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(b[0]), [b1] "v"(b[1]),
                           [b2] "v"(b[2]), [b3] "v"(b[3]), [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [lo] "v"(loff_g), [C] "s"(Cp),
                           [am] "s"(am)
                         : "memory", "scc");
#undef GESPMM_G_END2
        } else {
            v2f bl[4], bh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bl[j] = __builtin_shufflevector(b[j], b[j], 0, 1);
                bh[j] = __builtin_shufflevector(b[j], b[j], 2, 3);
            }
            // (one 16-byte store per lane; a store wider than 8 bytes reads its data late: two wait states before the registers are zeroed)
#define GESPMM_G_END4(R)                                                                                                              \
    "v_add_u32 %[t], " R ", %[lo]\n\ts_mov_b64 exec, %[am]\n\tglobal_store_dwordx4 %[t], v[60:63], %[C] sc1 nt\n\ts_mov_b64 exec, -1\n\t" \
    "s_nop 1\n\tv_mov_b64 v[60:61], 0\n\tv_mov_b64 v[62:63], 0\n\t"
            asm volatile(GESPMM_G_ROWS4(GESPMM_G_FMA2("v[60:61]", "%[c0]", "%[b0]") GESPMM_G_FMA2("v[62:63]", "%[c0]", "%[g0]"),
                                        GESPMM_G_FMA2("v[60:61]", "%[c1]", "%[b1]") GESPMM_G_FMA2("v[62:63]", "%[c1]", "%[g1]"),
                                        GESPMM_G_FMA2("v[60:61]", "%[c2]", "%[b2]") GESPMM_G_FMA2("v[62:63]", "%[c2]", "%[g2]"),
                                        GESPMM_G_FMA2("v[60:61]", "%[c3]", "%[b3]") GESPMM_G_FMA2("v[62:63]", "%[c3]", "%[g3]"), GESPMM_G_END4)
                         : [a4] "+{v[60:63]}"(acc), [t] "=&v"(t)
                         : [e] "s"(ends), [c0] "s"(cv[0]), [c1] "s"(cv[1]), [c2] "s"(cv[2]), [c3] "s"(cv[3]), [b0] "v"(bl[0]), [b1] "v"(bl[1]),
                           [b2] "v"(bl[2]), [b3] "v"(bl[3]), [g0] "v"(bh[0]), [g1] "v"(bh[1]), [g2] "v"(bh[2]), [g3] "v"(bh[3]), [r0] "s"(r0),
                           [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [lo] "v"(loff_g), [C] "s"(Cp), [am] "s"(am)
                         : "memory", "scc");
#undef GESPMM_G_END4
        }
    };
#undef GESPMM_G_ROWS4
#undef GESPMM_G_FMA2
    // max reducer: the same records in plain C++ — a row end stores the live lanes' accumulators and re-arms them with `empty`
    auto consume_max = [&](uint32_t ends, const uint64_t* cv, const vec_t* b) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if ((ends >> j) & 1u) {
                const uint32_t crow = (uint32_t)(cv[j] >> 32);
                if (active) __builtin_nontemporal_store(acc, reinterpret_cast<vec_t*>(reinterpret_cast<char*>(Cp) + (size_t)crow * n4 + loff_g));
                arm();
            } else {
                fma_row(cv[j], b[j]);
            }
        }
    };

    // The walk of spmm_staged.hip: one ballot per window and kind gives the masks "B row from memory" and "row end"; a chunk of U records
    // tests mask bits. A task's last record is a row end, so records behind `we` only reach an accumulator that is never stored.
    for (int kw = wb; kw < we; kw += kWin) {
        i2v nxt = win;
        if (kw + kWin < we) nxt = __builtin_nontemporal_load(evv + (kw - wb) + kWin + lane);
        const uint64_t gmask = __ballot((uint32_t)win.x < (uint32_t)kStagedRowEnd);    // B row from memory
        uint64_t lmask = __ballot((win.x & kStagedRowEnd) != 0 && win.x >= 0);          // row-end records ...
        if (we - kw < kWin) lmask &= (1ull << (we - kw)) - 1ull;                          // ... of THIS task
#pragma unroll 1
        for (int c = 0; c < kWin; c += U) {
            if (kw + c >= we) break;
            uint64_t cv[U];
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
                static_for_g<U>([&](auto J) { gather(code[decltype(J)::value], anymem, J, bv[decltype(J)::value]); });
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < U; ++j) asm volatile("" : "+v"(bv[j]));  // (uses of bv stay behind the wait)
            const uint32_t ends = (uint32_t)(lmask >> c) & ((1u << U) - 1u);
            if (ends == 0) {
#pragma unroll
                for (int j = 0; j < U; ++j) fma_row(cv[j], bv[j]);
            } else if constexpr (RED == kReduceMax) {
                consume_max(ends, cv, bv);
            } else {
#pragma unroll
                for (int j = 0; j < U; j += 4) consume4(ends >> j, cv + j, bv + j);
            }
        }
        win = nxt;
    }
}

}  // namespace

// Floats per lane the general kernel takes at width N: the widest vector that divides N among those with the fewest 64-lane tiles
// (N = 100: 2 — one tile either way, more live lanes; N = 200: 4; N = 602: 2; odd N: 1).
int staged_gen_vec(int64_t N) {
    if (N <= 0) return 0;
    int best = 1;
    int64_t best_tiles = (N + 63) / 64;
    for (int v = 2; v <= 4; v *= 2) {
        if (N % v) break;
        const int64_t t = (N + 64 * v - 1) / (64 * v);
        if (t < best_tiles) {
            best_tiles = t;
            best = v;
        }
    }
    return best;
}

StagedShape staged_gen_shape(int64_t N) {
    StagedShape sh = {0, 0, 0};
    const int v = staged_gen_vec(N);
    if (!v) return sh;
    static const int rows_env = getenv("GESPMM_STAGED_GEN_ROWS") ? atoi(getenv("GESPMM_STAGED_GEN_ROWS")) : 0;
    sh.waves = kGenWaves;
    sh.slots = kGenWaves * kGenLdsKb * 1024 / (256 * v);
    // rows per block: those of the tile width's tuned kernel (96 at 128-column tiles, 64 at 256: spmm_staged.hip, plan_policy.cpp
    // staged_rows_for refines them by mean degree); 64-column tiles (odd widths) by the same rule "about as many rows as staged slots / 1.7"
    sh.rows = rows_env > 0 ? rows_env : (v == 1 ? 192 : (v == 2 ? 96 : 64));
    return sh;
}

bool staged_gen_serves(int64_t M, int64_t K, int64_t N) {
    if (N < 1 || N > (1 << 20)) return false;
    return (uint64_t)(M > K ? M : K) * (uint64_t)N * 4ull < 0xFFFF0000ull;
}

hipError_t launch_spmm_staged_gen(const StagedArgs& a_in, int64_t M, int64_t K, int64_t N, int reduce, float empty, hipStream_t st) {
    if (a_in.nblocks <= 0) return hipSuccess;
    if (!staged_gen_serves(M, K, N) || a_in.waves != kGenWaves) return hipErrorInvalidValue;
    const int v = staged_gen_vec(N);
    if (a_in.slots != kGenWaves * kGenLdsKb * 1024 / (256 * v)) return hipErrorInvalidValue;  // tables of another tile width
    if ((reinterpret_cast<uintptr_t>(a_in.B) | reinterpret_cast<uintptr_t>(a_in.C)) & (uintptr_t)(4 * v - 1)) return hipErrorInvalidValue;
    StagedArgs a = a_in;
    a.n = (int32_t)N;
    a.ntiles = (int32_t)((N + 64 * v - 1) / (64 * v));
    a.empty = empty;
    const int64_t rounds = ((int64_t)a.nblocks + 7) / 8;
    const int64_t nwg = a.ntiles == 1 ? (int64_t)a.nblocks : rounds * 8 * a.ntiles;
    if (nwg > 0x7fffffffLL) return hipErrorInvalidConfiguration;
    const dim3 grid((unsigned)nwg), block(kGenWaves * 64);
#define GESPMM_GEN_LAUNCH(V)                                                                                              \
    do {                                                                                                                  \
        if (reduce == kReduceMax) hipLaunchKernelGGL((spmm_staged_gen_kernel<V, kReduceMax>), grid, block, 0, st, a);     \
        else hipLaunchKernelGGL((spmm_staged_gen_kernel<V, kReduceSum>), grid, block, 0, st, a);                          \
    } while (0)
    if (v == 1) GESPMM_GEN_LAUNCH(1);
    else if (v == 2) GESPMM_GEN_LAUNCH(2);
    else GESPMM_GEN_LAUNCH(4);
#undef GESPMM_GEN_LAUNCH
    return hipGetLastError();
}

}  // namespace gespmm
