// spmm_staged_narrow.hip — the staged-rows idea at NARROW widths (N = 16 / 32 / 64): a B row is 64 / 128 / 256 bytes, far less than the
// 1 KB a wavefront-level access carries, so the one-row-per-wavefront walk of spmm_staged.hip (scalar branches, SGPR records) would leave
// 7/8 of every instruction empty. Here a wavefront is G = 64 / W LANE GROUPS of W = N / 4 lanes (4 floats per lane), and every group walks
// ITS OWN range of the plan's record stream (spmm_kernels.h: entries {code, value} + one row-end record {kStagedRowEnd, C row} per row):
//
//   * the block's most used B rows are staged in LDS exactly as in spmm_staged.hip (same tables, H = 64 KB / row bytes = 1024 / 512 / 256
//     slots), so three entries out of four of a clustered graph are one `ds_read_b128` per lane — and ONE LDS instruction serves G rows;
//   * records live in VGPRs (every lane of a group loads its group's next 8 records itself: eight lanes, one cache line), so "staged or
//     memory", "entry or row end" are per-lane predicates: the gather of a step is one LDS read for the lanes whose row is staged and one
//     memory load for the others, issued under complementary EXEC masks into the SAME registers (disjoint lanes; assembly — the compiler
//     would wait for the LDS read before it lets the load overwrite "the same" register), the multiply-adds run under the entry mask, a
//     row end stores the group's accumulators and zeroes them under its mask;
//   * no row pointers, no row ids, no batches of rows that wait for their longest member, no per-row round trips: after the staging copy a
//     group streams its records from the first to the last.
//
// Every output element is still ONE fp32 chain over its row's entries in CSR order, one fused multiply-add per entry: the bits of every
// other variant. Sum reducer; B and C below 4 GB (always true at these widths unless K or M exceed 2^24-2^26 rows). Rows longer than
// kStagedMaxRow entries are handed to the streaming kernel's long-row pass by the plan, as for the wide kernel.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "spmm_device.h"
#include "spmm_kernels.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "spmm_staged_narrow.hip is written for gfx950: 80 KB of LDS per workgroup and its inline assembly do not build elsewhere"
#endif

namespace gespmm {

namespace {

using f4v = float __attribute__((ext_vector_type(4)));
using i4v = int __attribute__((ext_vector_type(4)));
using i2v = int __attribute__((ext_vector_type(2)));

template <int N, typename F>
__device__ __forceinline__ void static_for_n(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (N > 0) {
        static_for_n<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int kNarrowWin = 8;  // records per window = records gathered together (8 x 16 bytes per lane in flight)

// lane `src` (0 .. W-1) of every W-lane group -> all lanes of the group: ds_swizzle in bit mode (new lane = (lane & and) | or inside
// each half-wavefront; groups never straddle one). No LDS memory is touched: the value crosses the LDS crossbar.
template <int W, int SRC>
__device__ __forceinline__ int group_bcast(int v) {
    constexpr int kAnd = 0x1f & ~(W - 1);
    return __builtin_amdgcn_ds_swizzle(v, (SRC << 5) | kAnd);
}

template <int W, int WAVES, int LKB>
__global__ __launch_bounds__(WAVES * 64, 8) void spmm_staged_narrow_kernel(StagedArgs a) {  // (8 wavefronts per SIMD: <= 64 VGPRs)
    constexpr int G = 64 / W;                        // rows (lane groups) per wavefront step
    constexpr int kRowBytes = W * 16;                // N * 4
    constexpr int kRowShift = (W == 4) ? 6 : (W == 8 ? 7 : 8);
    constexpr int kLdsBytes = WAVES * LKB * 1024;    // LKB KB of staged B rows per wavefront (5: two 16-wavefront blocks fill the CU's 160 KB)
    constexpr int H = kLdsBytes / kRowBytes;         // staged rows per block
    constexpr int P = LKB;                           // 16-byte pieces of the staging copy per thread
    constexpr int S = (kNarrowWin + W - 1) / W;      // window registers per lane: record r of a window sits in lane r % W, register r / W
    static_assert(W == 4 || W == 8 || W == 16, "N = 16, 32 or 64");
    static_assert(kStagedPad >= 4 * kNarrowWin, "windows are read whole and three ahead: records past a task's end must be readable");
    __shared__ f4v s_hot[H * W];

    if (a.guard != nullptr && *a.guard != a.guard_want) return;  // (guarded launch: spmm_kernels.h — the whole grid, before any barrier)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane / W, l = lane % W;
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);
    // Round trip 1: the block's staged columns and the group's task
    const int32_t* hc = a.hot_cols + (size_t)blk * H;
    int hcol[P];
#pragma unroll
    for (int u = 0; u < P; ++u) hcol[u] = hc[(u * WAVES * 64 + tid) / W];
    const i4v t = reinterpret_cast<const i4v*>(a.tasks)[((size_t)blk * WAVES + wave) * G + g];
    const int gb = t.z, ge = t.w;  // the group's range of the record stream
    // Round trip 2: the staged rows and the group's first three windows of records (one record per lane and register)
    const f4v* B4 = reinterpret_cast<const f4v*>(a.B);
    f4v stage[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int i = u * WAVES * 64 + tid;
        stage[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (hcol[u] >= 0) stage[u] = B4[(size_t)hcol[u] * W + (i % W)];
    }
    // (plain loads: the compiler must KNOW they are in flight — it waits before the first use, three windows later, and it must never
    //  copy or spill a register whose load has not landed, which it would do to the output of an assembly load)
    auto load_window = [&](int pos, i2v (&w)[S]) {
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2)
            w[s2] = __builtin_nontemporal_load(reinterpret_cast<const i2v*>(a.ev) + (size_t)pos + (size_t)((s2 * W + l) & (kNarrowWin - 1)));
    };
    // FOUR window buffers with fixed roles that rotate by unrolling (below): a register whose load is still in flight must not be
    // copied — the hardware does not wait for it, only s_waitcnt does
    i2v wA[S], wB[S], wC[S], wD[S];
    load_window(gb, wA);
    load_window(gb + kNarrowWin, wB);      // (the stream is padded: always readable)
    load_window(gb + 2 * kNarrowWin, wC);
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) wD[s2] = i2v{0, 0};
#pragma unroll
    for (int u = 0; u < P; ++u)
        if (hcol[u] >= 0) s_hot[u * WAVES * 64 + tid] = stage[u];
    __syncthreads();
    // one offset serves both paths (LDS address of a staged row / byte offset into B): the staging array sits at LDS address 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot != 0u) __builtin_trap();
    const uint32_t loff = (uint32_t)l * 16u;
    const float* const Bp = a.B;
    float* const Cp = a.C;
    // the group's four accumulators are pinned to v[60:63]: the multiply-adds name them one by one, a row end stores them as ONE
    // 16-byte quadruple (inline assembly cannot name the parts of a register tuple)
    f4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
    int pos = gb;
    // one window of 8 records per group: `win` is walked, the window three ahead is requested into `req` (the buffer walked last)
    auto phase = [&](i2v (&win)[S], i2v (&req)[S]) {
        int vb[kNarrowWin];
        uint64_t fm[kNarrowWin], em[kNarrowWin];  // lanes whose record j is an entry (multiply-add) / a row end (store): SGPR masks
        f4v b[kNarrowWin];
        const int left = ge - pos;  // records of the task still ahead (<= 0: the group is done)
        int codes[kNarrowWin];
        // all eight code broadcasts first, ONE wait: the compiler counts the LDS operations it knows — with the assembly LDS reads of the
        // gathers between them its per-broadcast waits would also wait for those (in-order counter) and serialise the chunk. The VALUE
        // broadcasts follow behind the gathers' wait (eight registers less while 32 registers of B rows are in flight: the kernel must
        // not spill — a spilled register may be the target of an assembly load that has not landed; the Makefile checks it)
        static_for_n<kNarrowWin>([&](auto J) {
            constexpr int j = decltype(J)::value;
            codes[j] = group_bcast<W, j % W>(win[j / W].x);
        });
#pragma unroll
        for (int j = 0; j < kNarrowWin; ++j) asm volatile("" : "+v"(codes[j]));  // (all of them are used from here on)
        static_for_n<kNarrowWin>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int code = codes[j];
            // lane masks straight out of the compares (ICMP codes: 32 eq, 36 ult, 40 slt): the record is one of this task's, its B row
            // comes from memory (else: staged slot, or a row end -> slot 0), it ends a row
            const uint64_t m_act = __builtin_amdgcn_sicmp(j, left, 40);
            const uint64_t mm = __builtin_amdgcn_uicmp((uint32_t)code, (uint32_t)kStagedRowEnd, 36) & m_act;
            em[j] = __builtin_amdgcn_uicmp((uint32_t)code & 0xC0000000u, (uint32_t)kStagedRowEnd, 32) & m_act;
            fm[j] = m_act & ~em[j];
            const uint32_t off = (((uint32_t)code) << kRowShift) + loff +
                                 (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;  // LDS address / byte offset into B
            const float* const bp = Bp;  // (named here: a generic lambda does not capture through an asm operand)
            f4v bj;
            // staged lanes read LDS, the others memory: complementary EXEC masks, the same destination registers (disjoint lanes)
            asm volatile(
                "s_andn2_b64 exec, exec, %2\n\t"
                "ds_read_b128 %0, %1\n\t"
                "s_mov_b64 exec, %2\n\t"
                "s_cbranch_execz 1f\n\t"
                "global_load_dwordx4 %0, %1, %3\n"
                "1:\n\t"
                "s_mov_b64 exec, -1"
                : "=&v"(bj)
                : "v"(off), "s"(mm), "s"(bp)
                : "memory", "scc");  // (s_andn2 writes SCC; both blocks rewrite EXEC and rely on EXEC == -1 on entry: the walk has no divergent region)
            b[j] = bj;
        });
        // the window three ahead, requested AFTER this chunk's gathers: `vmcnt(S)` then waits for the gathers (and for everything older —
        // the window requested one chunk ago, which had that chunk's whole round trip to arrive) and leaves the new request in flight
        // (a group that is done keeps stepping with its wavefront: its requests are clamped to its own end — whole windows behind `ge` are
        //  inside the stream's padding, positions a task's length further on are not)
        load_window(pos + 3 * kNarrowWin < ge ? pos + 3 * kNarrowWin : ge, req);
        if constexpr (S == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < kNarrowWin; ++j) asm volatile("" : "+v"(b[j]));  // (uses of b stay behind the wait)
        static_for_n<kNarrowWin>([&](auto J) {
            constexpr int j = decltype(J)::value;
            vb[j] = group_bcast<W, j % W>(win[j / W].y);
        });
#pragma unroll
        for (int j = 0; j < kNarrowWin; ++j) asm volatile("" : "+v"(vb[j]));
#pragma unroll
        for (int j = 0; j < kNarrowWin; ++j) {
            uint32_t t32;
            // entries: acc += value * b under the entry mask; row ends: store acc to C row `value word`, zero it, under the row-end mask
            asm volatile(
                "s_mov_b64 exec, %[fm]\n\t"
                "v_fma_f32 v60, %[v], %[b0], v60\n\t"
                "v_fma_f32 v61, %[v], %[b1], v61\n\t"
                "v_fma_f32 v62, %[v], %[b2], v62\n\t"
                "v_fma_f32 v63, %[v], %[b3], v63\n\t"
                "s_mov_b64 exec, %[em]\n\t"
                "s_cbranch_execz 1f\n\t"
                "v_lshl_add_u32 %[t], %[v], %[sh], %[lo]\n\t"
                "global_store_dwordx4 %[t], v[60:63], %[C] sc1 nt\n\t"
                "s_nop 1\n\t"
                "v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n"
                "1:\n\t"
                "s_mov_b64 exec, -1"
                : [a4] "+{v[60:63]}"(acc), [t] "=&v"(t32)
                : [fm] "s"(fm[j]), [em] "s"(em[j]), [v] "v"(vb[j]), [b0] "v"(b[j][0]), [b1] "v"(b[j][1]), [b2] "v"(b[j][2]), [b3] "v"(b[j][3]),
                  [sh] "n"(kRowShift), [lo] "v"(loff), [C] "s"(Cp)
                : "memory", "scc");
        }
        pos += kNarrowWin;  // (every group steps; a finished one stays finished)
    };
    for (;;) {  // (wave-uniform exits: groups that are done idle through the masks)
        if (!__any(pos < ge)) break;
        phase(wA, wD);
        if (!__any(pos < ge)) break;
        phase(wB, wA);
        if (!__any(pos < ge)) break;
        phase(wC, wB);
        if (!__any(pos < ge)) break;
        phase(wD, wC);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last requests land before the wavefront's registers are released)
}

}  // namespace

// Widths this kernel serves (0 = not served) and the lanes per row.
static int narrow_lanes(int64_t N) { return N == 16 ? 4 : (N == 32 ? 8 : (N == 64 ? 16 : 0)); }

StagedShape staged_narrow_shape(int64_t N) {
    static const int rows_env = getenv("GESPMM_STAGED_NARROW_ROWS") ? atoi(getenv("GESPMM_STAGED_NARROW_ROWS")) : 0;
    StagedShape sh = {0, 0, 0};
    const int W = narrow_lanes(N);
    if (!W) return sh;
    static const int waves_env = getenv("GESPMM_STAGED_NARROW_WAVES") ? atoi(getenv("GESPMM_STAGED_NARROW_WAVES")) : 0;
    sh.waves = waves_env == 8 ? 8 : kStagedMaxWaves;
    static const int lds_env = getenv("GESPMM_STAGED_LDS_KB") ? atoi(getenv("GESPMM_STAGED_LDS_KB")) : 0;
    const int lkb = (sh.waves == kStagedMaxWaves && lds_env != 4) ? 5 : 4;  // (as for the wide kernel: spmm_staged.hip, staged_shape)
    sh.slots = sh.waves * lkb * 1024 / (W * 16);
    // as many rows per block as staged slots (N = 32: 384 / 512 / 640 / 768 / 1024 rows -> geometric 80.7 / 76.4 / 77.9 / 89.0 / 94.7 us,
    // products-shaped 1214 / 1185 / 1157 / 1188 / 1269; N = 64: 192 / 256 / 384 / 512 -> 127.6 / 124.5 / 132.1 / 139.9 us: profiles/r05/narrow_shapes.log)
    sh.rows = rows_env > 0 ? rows_env : sh.waves * 4096 / (W * 16);  // (512 / 256 rows at N = 32 / 64: what the sweep above found)
    return sh;
}

int staged_narrow_groups(int64_t N) {
    const int W = narrow_lanes(N);
    return W ? 64 / W : 0;
}

bool staged_narrow_serves(int64_t M, int64_t K, int64_t N) {
    if (!narrow_lanes(N)) return false;
    return (uint64_t)(M > K ? M : K) * (uint64_t)N * 4ull < 0xFFFF0000ull;
}

hipError_t launch_spmm_staged_narrow(const StagedArgs& a, int64_t M, int64_t K, int64_t N, hipStream_t st) {
    if (a.nblocks <= 0) return hipSuccess;
    if (!staged_narrow_serves(M, K, N) || (a.waves != kStagedMaxWaves && a.waves != 8)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)a.nblocks), block((unsigned)a.waves * 64);
    const int W = narrow_lanes(N);
    const int lkb = a.slots > 0 ? (int)((int64_t)a.slots * W * 16 / ((int64_t)a.waves * 1024)) : 4;  // what the tables were built for
    if (a.waves == 8) {  // (experiments: half-size blocks, GESPMM_STAGED_NARROW_WAVES=8)
        if (lkb != 4) return hipErrorInvalidValue;
        if (W == 8) hipLaunchKernelGGL((spmm_staged_narrow_kernel<8, 8, 4>), grid, block, 0, st, a);
        else if (W == 16) hipLaunchKernelGGL((spmm_staged_narrow_kernel<16, 8, 4>), grid, block, 0, st, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (lkb == 5) {
        switch (W) {
            case 4: hipLaunchKernelGGL((spmm_staged_narrow_kernel<4, kStagedMaxWaves, 5>), grid, block, 0, st, a); break;
            case 8: hipLaunchKernelGGL((spmm_staged_narrow_kernel<8, kStagedMaxWaves, 5>), grid, block, 0, st, a); break;
            case 16: hipLaunchKernelGGL((spmm_staged_narrow_kernel<16, kStagedMaxWaves, 5>), grid, block, 0, st, a); break;
            default: return hipErrorInvalidValue;
        }
    } else if (lkb == 4) {
        switch (W) {
            case 4: hipLaunchKernelGGL((spmm_staged_narrow_kernel<4, kStagedMaxWaves, 4>), grid, block, 0, st, a); break;
            case 8: hipLaunchKernelGGL((spmm_staged_narrow_kernel<8, kStagedMaxWaves, 4>), grid, block, 0, st, a); break;
            case 16: hipLaunchKernelGGL((spmm_staged_narrow_kernel<16, kStagedMaxWaves, 4>), grid, block, 0, st, a); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gespmm
