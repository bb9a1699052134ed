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

#include "spmm_device.h"
#include "spmm_kernels.h"

#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__) || defined(__gfx942__)
// (the inline assembly below spells its memory instructions — sc1 / nt modifiers, SGPR-base global loads — for gfx942 / gfx950)
#else
#error "spmm_staged.hip is written for gfx950 (gfx942 ISA compatible): its inline assembly does not assemble elsewhere"
#endif

namespace gespmm {

namespace {

typedef const __attribute__((address_space(4))) int32_t* cint_ptr;  // constant address space: scalar loads
using f4v = float __attribute__((ext_vector_type(4)));

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
template <int VEC, int U, int TSHIFT, bool PAGE2>
__global__ __launch_bounds__(kStagedWaves * 64) void spmm_staged_kernel(StagedArgs a) {
    using vec_t = typename LaneVec<VEC>::type;
    constexpr int kRowBytes = 256 * VEC;          // bytes of a row inside one tile
    constexpr int kRowShift = (VEC == 2) ? 9 : 10;
    constexpr int kGlobalShift = kRowShift + TSHIFT;  // log2(N * 4): row stride of B and C
    constexpr int H = kStagedLdsBytes / kRowBytes;  // staged rows per block
    constexpr int kRowF4 = kRowBytes / 16;
    static_assert(kStagedPad >= 3 * U, "the stream is over-read by up to three chunks past a task's end");
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
    const int task = blk * kStagedWaves + wave;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr ev = (cint_ptr)(uintptr_t)a.ev + (size_t)wb * 2;  // {code, value bits} per entry
    const float* Bp = a.B + (size_t)tile * (64 * VEC);
    const float* BpHi = Bp + (1ull << 30);  // + 4 GB (PAGE2)
    (void)BpHi;
    const uint32_t loff = (uint32_t)lane * (4u * VEC);
    {
        const int total = ((cint_ptr)(uintptr_t)a.nhot)[blk] * kRowF4;
        const int32_t* hc = a.hot_cols + (size_t)blk * H;
        const f4v* B4 = reinterpret_cast<const f4v*>(a.B);
        for (int i0 = 0; i0 < total; i0 += kStagedWaves * 64 * 4) {
            f4v r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // (clamped, not predicated: the four loads stay in flight together)
                const int i = i0 + u * kStagedWaves * 64 + tid;
                const int ic = i < total ? i : total - 1;
                r[u] = B4[(((size_t)hc[ic / kRowF4] << TSHIFT) + (size_t)tile) * kRowF4 + (ic % kRowF4)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kStagedWaves * 64 + tid;
                if (i < total) s_hot[i] = r[u];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start
    }
    if (nrows == 0) return;
    // one offset serves both paths (LDS address of a staged row / byte offset into B): the staging array must sit at LDS address 0
    // (it is the kernel's only LDS object; a compile-time constant — the check folds away)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot != 0u) __builtin_trap();
    int cur = 0;
    int rend = rowptr[1], rend_next = rowptr[nrows > 1 ? 2 : 1];
    int crow = perm[0], crow_next = perm[nrows > 1 ? 1 : 0];
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    auto flush = [&]() {  // row `cur` is complete: store it, step to the next one (its end and C row were requested a row ago)
        float* Crow = a.C + (((size_t)crow << TSHIFT) + (size_t)tile) * (size_t)(64 * VEC);
        vec_t out;
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = acc[i];
        if constexpr (VEC == 2) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(out), "s"(Crow) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(loff), "v"(out), "s"(Crow) : "memory");
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        ++cur;
        rend = rend_next;
        crow = crow_next;
        const int nx = (cur + 1 < nrows) ? cur + 1 : nrows - 1;
        rend_next = rowptr[nx + 1];
        crow_next = perm[nx];
    };
    // code: bit 31 = staged (low bits: LDS slot); else column, bit 30 = "far": the row lives in a distant part of the clustered order,
    // nobody near this block will ask for it again — gathered with `nt` so that it does not push the neighbourhood's rows out of L2
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
                "s_bitcmp1_b32 %3, 30\n\t"
                "s_cbranch_scc1 3f\n\t"
                "global_load_dwordx4 %0, %2, %1\n\t"
                "s_branch 2f\n"
                "3:\n\t"
                "global_load_dwordx4 %0, %2, %1 nt\n\t"
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
                "s_bitcmp1_b32 %2, 30\n\t"
                "s_cbranch_scc1 3f\n\t"
                "global_load_dwordx2 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "3:\n\t"
                "global_load_dwordx2 %0, %1, %3 nt\n\t"
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
                "s_bitcmp1_b32 %2, 30\n\t"
                "s_cbranch_scc1 3f\n\t"
                "global_load_dwordx4 %0, %1, %3\n\t"
                "s_branch 2f\n"
                "3:\n\t"
                "global_load_dwordx4 %0, %1, %3 nt\n\t"
                "s_branch 2f\n"
                "1:\n\t"
                "ds_read_b128 %0, %4\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp), "v"(voff_l)
                : "memory", "scc");
    };
    auto fma_row = [&](int vbits, const vec_t& b) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(vbits), "v"(b[i]));
    };
    auto wait_all = [&](vec_t (&bv)[U]) {
        static_assert(U == 8, "operand list below");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7])::"memory");
    };
    // (The stream is padded by kStagedPad entries; chunk slots past `we` belong to the next task or the padding: harmless
    // gathers — staged slots are < H, columns are valid — that are not summed.)
    auto process = [&](const int (&e)[2 * U], int k) {
        vec_t bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) gather(e[2 * j], bv[j]);
        wait_all(bv);
        if (k + U <= rend) {
#pragma unroll
            for (int j = 0; j < U; ++j) fma_row(e[2 * j + 1], bv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (k + j < we) {
                    while (k + j >= rend) flush();  // rows ending before this entry (incl. empty ones)
                    fma_row(e[2 * j + 1], bv[j]);
                }
            }
        }
    };
    int eA[2 * U], eB[2 * U];
#pragma unroll
    for (int i = 0; i < 2 * U; ++i) eA[i] = ev[i];
    for (int k = wb; k < we;) {
#pragma unroll
        for (int i = 0; i < 2 * U; ++i) eB[i] = ev[2 * U + i];
        process(eA, k);
        k += U;
        if (k >= we) break;
#pragma unroll
        for (int i = 0; i < 2 * U; ++i) eA[i] = ev[4 * U + i];
        process(eB, k);
        k += U;
        ev += 4 * U;
    }
    while (cur < nrows) flush();  // last row and any trailing empty rows
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

int staged_block_rows(int64_t N) {
    // measured on the products-shaped community graph (us at N = 128 / 256): 64 rows 3286 / 5584, 80: 3122 / 5596, 96: 3012 / 5834,
    // 112: 3120 / 6264, 128: 3065 / 6440 — about as many rows as LDS holds staged rows for the width's row size
    int t;
    const int tc = staged_tile_cols(N, &t);
    return tc == 128 ? 96 : (tc == 256 ? 64 : 0);
}

int staged_rows_per_block_lds(int64_t N) {
    int t;
    const int tc = staged_tile_cols(N, &t);
    return tc ? kStagedLdsBytes / (tc * 4) : 0;
}

// B beyond 4 GB: two 4 GB halves (tiled widths only), up to 8 GB.
bool staged_serves(int64_t K, int64_t N) {
    int t;
    if (!staged_tile_cols(N, &t)) return false;
    const uint64_t bytes = (uint64_t)K * (uint64_t)N * 4ull;
    return bytes < 0xFFFF0000ull || (t >= 1 && bytes < 0x1FFFF0000ull);
}

hipError_t launch_spmm_staged(const StagedArgs& a, int64_t K, int64_t N, hipStream_t st) {
    if (a.nblocks <= 0) return hipSuccess;
    if (!staged_serves(K, N)) return hipErrorInvalidValue;
    int t;
    const int tc = staged_tile_cols(N, &t);
    const bool paged = (uint64_t)K * (uint64_t)N * 4ull >= 0xFFFF0000ull;
    const dim3 grid((unsigned)a.nblocks << (t > 0 ? t : 0)), block(kStagedWaves * 64);
    if (tc == 128) hipLaunchKernelGGL((spmm_staged_kernel<2, 8, 0, false>), grid, block, 0, st, a);
    else if (tc == 256 && t == 0) hipLaunchKernelGGL((spmm_staged_kernel<4, 8, 0, false>), grid, block, 0, st, a);
    else if (tc == 256 && t == 1 && !paged) hipLaunchKernelGGL((spmm_staged_kernel<4, 8, 1, false>), grid, block, 0, st, a);
    else if (tc == 256 && t == 1 && paged) hipLaunchKernelGGL((spmm_staged_kernel<4, 8, 1, true>), grid, block, 0, st, a);
    else if (tc == 256 && t == 2 && !paged) hipLaunchKernelGGL((spmm_staged_kernel<4, 8, 2, false>), grid, block, 0, st, a);
    else if (tc == 256 && t == 2 && paged) hipLaunchKernelGGL((spmm_staged_kernel<4, 8, 2, true>), grid, block, 0, st, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace gespmm
