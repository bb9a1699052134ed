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

namespace gespmm {

namespace {

typedef const __attribute__((address_space(4))) int32_t* cint_ptr;  // constant address space: scalar loads
using f4v = float __attribute__((ext_vector_type(4)));

template <int VEC> struct LaneVec;
template <> struct LaneVec<2> { using type = float __attribute__((ext_vector_type(2))); };
template <> struct LaneVec<4> { using type = float __attribute__((ext_vector_type(4))); };

template <int VEC, int U>
__global__ __launch_bounds__(kStagedWaves * 64) void spmm_staged_kernel(StagedArgs a) {
    using vec_t = typename LaneVec<VEC>::type;
    constexpr int kRowBytes = 256 * VEC;          // N * 4
    constexpr int kRowShift = (VEC == 2) ? 9 : 10;
    constexpr int H = kStagedLdsBytes / kRowBytes;  // staged rows per block
    constexpr int kRowF4 = kRowBytes / 16;
    __shared__ f4v s_hot[H * kRowF4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);
    const int task = blk * kStagedWaves + wave;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr ev = (cint_ptr)(uintptr_t)a.ev + (size_t)wb * 2;  // {code, value bits} per entry
    const float* Bp = a.B;
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
                r[u] = B4[(size_t)hc[ic / kRowF4] * kRowF4 + (ic % kRowF4)];
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
        float* Crow = a.C + (size_t)crow * (size_t)(64 * VEC);
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
        const uint32_t voff =
            ((uint32_t)code << kRowShift) + loff + (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f4v*)s_hot;
        if constexpr (VEC == 2)
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
                "ds_read_b64 %0, %1\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp)
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
                "ds_read_b128 %0, %1\n"
                "2:"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp)
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

int staged_block_rows(int64_t N) {
    // measured on the products-shaped community graph (us at N = 128 / 256): 64 rows 3286 / 5584, 80: 3122 / 5596, 96: 3012 / 5834,
    // 112: 3120 / 6264, 128: 3065 / 6440 — about as many rows as LDS holds staged rows for the width's row size
    if (N == 128) return 96;
    if (N == 256) return 64;
    return 0;
}

int staged_rows_per_block_lds(int64_t N) {
    if (N == 128) return kStagedLdsBytes / 512;
    if (N == 256) return kStagedLdsBytes / 1024;
    return 0;
}

hipError_t launch_spmm_staged(const StagedArgs& a, int64_t N, hipStream_t st) {
    if (a.nblocks <= 0) return hipSuccess;
    if (N == 128)
        hipLaunchKernelGGL((spmm_staged_kernel<2, 8>), dim3((unsigned)a.nblocks), dim3(kStagedWaves * 64), 0, st, a);
    else if (N == 256)
        hipLaunchKernelGGL((spmm_staged_kernel<4, 8>), dim3((unsigned)a.nblocks), dim3(kStagedWaves * 64), 0, st, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace gespmm
