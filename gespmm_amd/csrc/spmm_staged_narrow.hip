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

#include "spmm_device.h"
#include "spmm_kernels.h"

#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__) || defined(__gfx942__)
#else
#error "spmm_staged_narrow.hip is written for gfx950 (gfx942 ISA compatible): its inline assembly does not assemble elsewhere"
#endif

namespace gespmm {

namespace {

using f4v = float __attribute__((ext_vector_type(4)));
using i4v = int __attribute__((ext_vector_type(4)));

constexpr int kNarrowWin = 4;    // records of its stream a lane group holds in registers (8: 16 registers spilled at 64 VGPRs)
constexpr int kNarrowChunk = 4;  // records gathered together (4 x 16 bytes per lane in flight)

template <int W, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 8) void spmm_staged_narrow_kernel(StagedArgs a) {  // (8 wavefronts per SIMD = two blocks per CU: <= 64 VGPRs)
    constexpr int G = 64 / W;                        // rows (lane groups) per wavefront step
    constexpr int kRowBytes = W * 16;                // N * 4
    constexpr int kRowShift = (W == 4) ? 6 : (W == 8 ? 7 : 8);
    constexpr int kLdsBytes = WAVES * kStagedLdsPerWave;
    constexpr int H = kLdsBytes / kRowBytes;         // staged rows per block
    constexpr int P = kStagedLdsPerWave / 1024;      // 16-byte pieces of the staging copy per thread
    static_assert(W == 4 || W == 8 || W == 16, "N = 16, 32 or 64");
    static_assert(kStagedPad >= kNarrowWin, "a window is read whole: up to kNarrowWin - 1 records past a task's end");
    __shared__ f4v s_hot[H * W];

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
    // Round trip 2: the staged rows and the group's first window of records
    const f4v* B4 = reinterpret_cast<const f4v*>(a.B);
    f4v stage[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int i = u * WAVES * 64 + tid;
        stage[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (hcol[u] >= 0) stage[u] = B4[(size_t)hcol[u] * W + (i % W)];
    }
    const i4v* ev4 = reinterpret_cast<const i4v*>(a.ev);  // two records per 16 bytes (record positions need not be even: dword-aligned loads)
    auto load_window = [&](int pos, i4v (&w)[kNarrowWin / 2]) {
        const int32_t* p = a.ev + 2 * (size_t)pos;
#pragma unroll
        for (int j = 0; j < kNarrowWin / 2; ++j) w[j] = *reinterpret_cast<const i4v*>(p + 4 * j);  // (the stream is padded)
    };
    (void)ev4;
    i4v win[kNarrowWin / 2];
    load_window(gb, win);
#pragma unroll
    for (int u = 0; u < P; ++u)
        if (hcol[u] >= 0) s_hot[u * WAVES * 64 + tid] = stage[u];
    __syncthreads();
    // one offset serves both paths (LDS address of a staged row / byte offset into B): the staging array sits at LDS address 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot != 0u) __builtin_trap();
    const uint32_t loff = (uint32_t)l * 16u;
    const float* const Bp = a.B;
    float* const Cp = a.C;
    f4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
    int pos = gb;
    while (__any(pos < ge)) {  // (wave-uniform: groups that are done idle through the masks)
        i4v nxt[kNarrowWin / 2];
        load_window(pos + kNarrowWin < ge ? pos + kNarrowWin : pos, nxt);  // (a finished group re-reads its last window: valid addresses, no use)
#pragma unroll
        for (int c = 0; c < kNarrowWin; c += kNarrowChunk) {
            int code[kNarrowChunk], vb[kNarrowChunk];
            f4v b[kNarrowChunk];
#pragma unroll
            for (int j = 0; j < kNarrowChunk; ++j) {
                const i4v r = win[(c + j) / 2];
                code[j] = ((c + j) & 1) ? r.z : r.x;
                vb[j] = ((c + j) & 1) ? r.w : r.y;
            }
#pragma unroll
            for (int j = 0; j < kNarrowChunk; ++j) {
                const bool act = pos + c + j < ge;
                const bool mem = act && (uint32_t)code[j] < (uint32_t)kStagedRowEnd;  // B row from memory (else: staged slot, or a row end -> slot 0)
                const uint32_t off = (((uint32_t)code[j]) << kRowShift) + loff +
                                     (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;  // LDS address / byte offset into B
                const uint64_t mm = __ballot(mem);
                // staged lanes read LDS, the others memory: complementary EXEC masks, the same destination registers (disjoint lanes)
                asm volatile(
                    "s_andn2_b64 exec, exec, %2\n\t"
                    "ds_read_b128 %0, %1\n\t"
                    "s_mov_b64 exec, %2\n\t"
                    "s_cbranch_execz 1f\n\t"
                    "global_load_dwordx4 %0, %1, %3\n"
                    "1:\n\t"
                    "s_mov_b64 exec, -1"
                    : "=&v"(b[j])
                    : "v"(off), "s"(mm), "s"(Bp)
                    : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < kNarrowChunk; ++j) asm volatile("" : "+v"(b[j]));  // (uses of b stay behind the wait)
#pragma unroll
            for (int j = 0; j < kNarrowChunk; ++j) {
                const bool act = pos + c + j < ge;
                const bool end = act && (code[j] & kStagedRowEnd) != 0 && code[j] >= 0;
                if (act && !end) {
                    const float v = __int_as_float(vb[j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(v, b[j][i], acc[i]);
                }
                if (end) {  // the value word of a row-end record is the C row
                    *reinterpret_cast<f4v*>(reinterpret_cast<char*>(Cp) + (((size_t)(uint32_t)vb[j]) << kRowShift) + loff) = acc;
                    acc = f4v{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
        }
        if (pos < ge) pos += kNarrowWin;
#pragma unroll
        for (int j = 0; j < kNarrowWin / 2; ++j) win[j] = nxt[j];
    }
}

}  // namespace

// Widths this kernel serves (0 = not served) and the lanes per row.
static int narrow_lanes(int64_t N) { return N == 16 ? 4 : (N == 32 ? 8 : (N == 64 ? 16 : 0)); }

StagedShape staged_narrow_shape(int64_t N) {
    static const int rows_env = getenv("GESPMM_STAGED_NARROW_ROWS") ? atoi(getenv("GESPMM_STAGED_NARROW_ROWS")) : 0;
    StagedShape sh = {0, 0, 0};
    const int W = narrow_lanes(N);
    if (!W) return sh;
    sh.waves = kStagedMaxWaves;
    sh.slots = sh.waves * kStagedLdsPerWave / (W * 16);
    sh.rows = rows_env > 0 ? rows_env : sh.slots * 3 / 4;  // as many rows as the wide kernel takes per staged slot (96 : 128)
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
    if (!staged_narrow_serves(M, K, N) || a.waves != kStagedMaxWaves) return hipErrorInvalidValue;
    const dim3 grid((unsigned)a.nblocks), block((unsigned)a.waves * 64);
    switch (narrow_lanes(N)) {
        case 4: hipLaunchKernelGGL((spmm_staged_narrow_kernel<4, kStagedMaxWaves>), grid, block, 0, st, a); break;
        case 8: hipLaunchKernelGGL((spmm_staged_narrow_kernel<8, kStagedMaxWaves>), grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL((spmm_staged_narrow_kernel<16, kStagedMaxWaves>), grid, block, 0, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gespmm
