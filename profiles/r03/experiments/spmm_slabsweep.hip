// spmm_slabsweep.hip — EXPERIMENT (round 3): the cooperative slab sweep for dense graphs that the round-2 review asked for.
//
// The shipped cache-blocked path (spmm_slab_kernel) makes one launch per ~6 MB column slab of B and read-modify-writes C in
// every launch: on the reddit-shaped graph 23.5 GB of slab re-fetches (a 6 MB slab does not fit a 4 MiB L2 beside the streams)
// and 4.5 GB of C traffic for 1.16 GB of algorithmic bytes. Here ONE cooperative launch does the whole product:
//   * slabs small enough to stay in L2 (3 MB);
//   * every lane group keeps the accumulators of its R = 8 rows in REGISTERS across all slabs (32 VGPRs at 128 columns):
//     C is written once, never read;
//   * a grid-wide barrier after every slab keeps all workgroups on the same slab (without it they drift apart within a few
//     slabs: profiles/r02/slab_resident_experiment.log). The spin is bounded: a barrier that does not complete sets an error
//     flag and the kernel runs to its end instead of hanging the device.
// Each output element is still one fp32 chain over the row's entries in CSR order (the per-slab ranges of a row are
// consecutive CSR ranges), so the result has the same bits as every other variant.
// Reached only through gespmm_debug_slabsweep_f32 (tests / measurements); see profiles/r03/slab_sweep_cooperative.log.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gespmm.h"
#include "spmm_device.h"
#include "spmm_kernels.h"

namespace gespmm {

namespace {

constexpr int kSweepRows = 8;  // rows per lane group (accumulators in registers)
constexpr int kSweepU = 8;     // gathers in flight per lane group

struct SweepArgs {
    const int32_t* rowptr;
    const int32_t* colind;
    const float* val;
    const float* B;
    float* C;
    const int32_t* split;  // [(nslab + 1)][M]
    int32_t M, N, nslab;
    unsigned* bar;         // {count, generation, error}
};

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nwg, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&bar[0], 1u) == nwg - 1) {
            atomicExch(&bar[0], 0u);
            __threadfence();
            atomicAdd(&bar[1], 1u);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 22)) {  // ~ a second: something is wrong — flag it and go on rather than hang
                    atomicExch(&bar[2], 1u);
                    break;
                }
            }
        }
    }
    ++gen;
    __syncthreads();
}

template <bool VALUED>
__global__ __launch_bounds__(kThreads) void spmm_slabsweep_kernel(SweepArgs a) {
    constexpr int W = 32, V = 4, R = kSweepRows, U = kSweepU;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const unsigned nwg = gridDim.x;
    const int groups_per_wg = kWaves * 2;
    const long long total_groups = (long long)nwg * groups_per_wg;
    const long long my_group = (long long)blockIdx.x * groups_per_wg + wave * 2 + g;
    const int col0 = l * V;
    const bool colok = col0 < a.N;
    const uint32_t cbytes = colok ? (uint32_t)col0 * 4u : 0u;
    const uint32_t rowbytes = (uint32_t)a.N * 4u;
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    unsigned gen = 0;
    const long long rows_per_round = total_groups * R;
    const int rounds = (int)((a.M + rows_per_round - 1) / rows_per_round);

    for (int round = 0; round < rounds; ++round) {
        const long long row0 = (long long)round * rows_per_round + my_group * R;
        float acc[R][V];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[i][k] = 0.0f;

        for (int s = 0; s < a.nslab; ++s) {
            const int32_t* sb = a.split + (size_t)s * a.M;
            const int32_t* se = a.split + (size_t)(s + 1) * a.M;
            // split points of the group's R rows: ONE pair of loads (lane i < R takes row i), broadcast by shuffles
            int myb = 0, mye = 0;
            if (l < R && row0 + l < a.M) {
                myb = sb[row0 + l];
                mye = se[row0 + l];
            }
            // software pipeline over the rows: the first tile of row i + 1 is requested before row i's gathers
            int nb = __shfl(myb, 0, W), ne = __shfl(mye, 0, W);
            int pc = 0;
            float pv = 0.0f;
            if (nb + l < ne) {
                pc = load_csr(a.colind + nb + l);
                if constexpr (VALUED) pv = load_csr(a.val + nb + l);
                else pv = 1.0f;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int b = nb, e = ne;
                int c = pc;
                float v = pv;
                if (i + 1 < R) {
                    nb = __shfl(myb, i + 1, W);
                    ne = __shfl(mye, i + 1, W);
                    pc = 0;
                    pv = 0.0f;
                    if (nb + l < ne) {
                        pc = load_csr(a.colind + nb + l);
                        if constexpr (VALUED) pv = load_csr(a.val + nb + l);
                        else pv = 1.0f;
                    }
                }
                for (int t0 = b; t0 < e; t0 += W) {
                    if (t0 > b) {  // further tiles of a long segment
                        const int p = t0 + l;
                        c = 0;
                        v = 0.0f;
                        if (p < e) {
                            c = load_csr(a.colind + p);
                            if constexpr (VALUED) v = load_csr(a.val + p);
                            else v = 1.0f;
                        }
                    }
                    const int cnt = (e - t0 < W) ? e - t0 : W;
                    for (int k = 0; k < cnt; k += U) {
                        float bv[U][V];
                        float vv[U];
#pragma unroll
                        for (int j = 0; j < U; ++j) {
                            const int src = (k + j < cnt) ? k + j : cnt - 1;  // clamped: always a live entry, dropped below
                            const uint32_t off = (uint32_t)__shfl(c, src, W) * rowbytes;
                            vv[j] = __shfl(v, src, W);
                            load_vec<V>(bv[j], Bbase + (off + cbytes));
                        }
#pragma unroll
                        for (int j = 0; j < U; ++j) {
                            if (k + j < cnt) {
#pragma unroll
                                for (int q = 0; q < V; ++q) acc[i][q] = __builtin_fmaf(vv[j], bv[j][q], acc[i][q]);
                            }
                        }
                    }
                }
            }
            grid_barrier(a.bar, nwg, gen);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long long row = row0 + i;
            if (row < a.M && colok) store_vec<V, false>(a.C + (size_t)row * (size_t)a.N + col0, acc[i]);
        }
    }
}

}  // namespace

}  // namespace gespmm

extern "C" int gespmm_debug_slabsweep_f32(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B,
                                          float* C, int64_t M, int64_t K, int64_t N, int64_t slab_rows, void* split_ws,
                                          int32_t build_split, void* stream) {
    using namespace gespmm;
    if (M <= 0 || N <= 0 || N > 128 || (N % 4) != 0 || slab_rows < 1 || !split_ws) return GESPMM_EINVAL;
    if ((uint64_t)K * (uint64_t)N * 4ull >= (1ull << 32)) return GESPMM_ERANGE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nslab = (int)((K + slab_rows - 1) / slab_rows);
    int32_t* split = static_cast<int32_t*>(split_ws);  // (nslab + 1) * M ints + 4 words for the barrier
    unsigned* bar = reinterpret_cast<unsigned*>(split + (size_t)(nslab + 1) * M);
    hipError_t e = hipSuccess;
    if (build_split) e = launch_slabplan(rowptr, colind, split, (int)M, nslab, (int)slab_rows, st);
    if (e == hipSuccess) e = hipMemsetAsync(bar, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    SweepArgs a = {rowptr, colind, val, B, C, split, (int32_t)M, (int32_t)N, nslab, bar};
    void* params[] = {&a};
    const void* fn = val ? (const void*)spmm_slabsweep_kernel<true> : (const void*)spmm_slabsweep_kernel<false>;
    e = val ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spmm_slabsweep_kernel<true>, kThreads, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spmm_slabsweep_kernel<false>, kThreads, 0);
    if (e != hipSuccess || per_cu < 1) return e != hipSuccess ? (int)e : GESPMM_EINVAL;
    long long nwg = (long long)cus * per_cu;
    const long long need = (M + kWaves * 2 * kSweepRows - 1) / (kWaves * 2 * kSweepRows);
    if (nwg > need) nwg = need;
    e = hipLaunchCooperativeKernel(fn, dim3((unsigned)nwg), dim3(kThreads), params, 0, st);
    if (e != hipSuccess) return (int)e;
    unsigned h[3] = {0, 0, 0};
    e = hipMemcpyAsync(h, bar, 12, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    return h[2] ? -100 : (int)nwg;  // -100: a grid barrier timed out (result invalid); otherwise the grid size used
}
