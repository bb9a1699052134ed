// spmm_ldsrow.hip — the plan's kernel for row-clustered matrices: distinct B rows of a task staged in LDS.
//
// A clustered plan (plan.cpp) processes rows that share neighbours next to each other. The streaming kernels
// still gather a B row once per USE (one global load per non-zero); with clustered rows half or more of those
// loads name a row that the same wavefront needs again a few entries later. This kernel fetches every DISTINCT
// B row of a task once, straight into LDS (gfx950 `global_load_lds_dwordx4`: no VGPR staging, all of a task's
// row fetches in flight at the same time), and the row sums then read LDS:
//
//   * a task is one 640-byte RECORD the plan writes at analysis time — header, C row ids, the distinct column
//     ids (<= 32), per non-zero the value and the LDS slot of its B row, per row the entry range — so the whole
//     description of a task is ONE memory round trip at a computable address (record w belongs to wavefront w);
//   * all distinct rows are requested back to back (64/W rows per instruction: W lanes x 16 bytes cover the
//     column tile of one row), then one wait;
//   * the W-lane groups of the wavefront walk the task's rows; each output element is ONE fp32 chain over the
//     row's non-zeros in CSR order with one fused multiply-add per non-zero — the same arithmetic as every other
//     variant (spmm_test.cu:182-203 semantics), so the bits are unchanged;
//   * rows that do not fit one record (more than 64 entries or 32 distinct columns) are a chain of records
//     handled by ONE wavefront that carries the accumulator from record to record (the wavefronts of the
//     continuation records exit at once).
//
// Column tiles are 4W floats (W = 4..32 lanes x dwordx4), wider N takes several tiles (several workgroups per
// record). N must be a multiple of 4; everything else stays on the streaming kernel with the plan's task table.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "spmm_kernels.h"

namespace gespmm {

namespace {

__device__ __forceinline__ int xcd_contiguous_id(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int RED, bool VALUED>
__device__ __forceinline__ float combine1(float acc, float a, float b) {
    if constexpr (RED == kReduceMax) return fmaxf(acc, b);
    else if constexpr (VALUED) return __builtin_fmaf(a, b, acc);
    else return acc + b;
}

using f4 = float __attribute__((ext_vector_type(4)));

template <int W, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_ldsrow_kernel(LdsRowArgs a) {
    constexpr int G = 64 / W;       // lane groups per wavefront = B rows fetched per instruction
    constexpr int ROWB = W * 16;    // bytes of one staged row (this column tile)
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    __shared__ __attribute__((aligned(16))) char s_rows[kWaves][kRecDistinct * ROWB];
    __shared__ int s_off[kWaves][kRecEntries];
    __shared__ float s_val[VALUED ? kWaves : 1][VALUED ? kRecEntries : 1];
    __shared__ int s_rp[kWaves][kRecRows + 1];
    __shared__ int s_crow[kWaves][kRecRows];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    const int item = xcd_contiguous_id(blockIdx.x, a.nblk * a.ntile);
    int tile = 0, rb = item;
    if (a.ntile > 1) {
        tile = item % a.ntile;
        rb = item / a.ntile;
    }
    const int wid = rb * kWaves + wave;
    if (wid >= a.nrec) return;
    const int32_t* rec = a.recs + (size_t)wid * kRecWords;
    int4 h = *reinterpret_cast<const int4*>(rec);
    const int kind = __builtin_amdgcn_readfirstlane(h.w);
    if (kind < 0) return;  // continuation of a long row: the wavefront of its first record does it
    const int nseg = kind > 0 ? kind : 1;

    const int col0 = tile * (W * 4) + l * 4;
    const bool colok = col0 < a.N;  // N % 4 == 0: a lane's four columns are in range together
    const off_t rowbytes = (off_t)a.N * 4u;
    const off_t cbyte = colok ? (off_t)col0 * 4u : (off_t)0;
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;
    char* rows_lds = s_rows[wave];

    float acc[4] = {init, init, init, init};
    for (int seg = 0; seg < nseg; ++seg) {
        if (seg > 0) {
            rec += kRecWords;
            h = *reinterpret_cast<const int4*>(rec);
        }
        const int nrows = __builtin_amdgcn_readfirstlane(h.x);
        const int ndist = __builtin_amdgcn_readfirstlane(h.z);
        // ---- the record: five independent loads at fixed offsets
        const int crow = rec[kRecOffCrow + (lane & 31)];
        const int dcol = rec[kRecOffDcol + (lane & 31)];
        float v = 1.0f;
        if constexpr (VALUED) v = reinterpret_cast<const float*>(rec)[kRecOffVal + lane];
        const int slot = reinterpret_cast<const uint8_t*>(rec)[kRecOffSlotBytes + lane];
        const int rpb = reinterpret_cast<const uint8_t*>(rec)[kRecOffRpBytes + (lane <= kRecRows ? lane : kRecRows)];
        // ---- every distinct B row of the record: global -> LDS, G rows per instruction, all in flight together
        for (int j = 0; j < ndist; j += G) {
            const int c = __shfl(dcol, (j + g) & 31, 64);
            if (j + g < ndist && colok) {
                const char* src = Bbase + (off_t)((off_t)(uint32_t)c * rowbytes + cbyte);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(rows_lds + j * ROWB), 16, 0, 0);
            }
        }
        s_off[wave][lane] = slot * ROWB;
        if constexpr (VALUED) s_val[wave][lane] = v;
        if (lane <= kRecRows) s_rp[wave][lane] = rpb;
        if (lane < kRecRows) s_crow[wave][lane] = crow;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();

        for (int r = g; r < nrows; r += G) {
            const int lb = s_rp[wave][r], hb = s_rp[wave][r + 1];
            if (seg == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = init;
            }
            int k = lb;
            for (; k + 4 <= hb; k += 4) {
                int o[4];
                float vv[4];
                f4 b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = s_off[wave][k + j];
                    if constexpr (VALUED) vv[j] = s_val[wave][k + j];
                    else vv[j] = 1.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f4*>(rows_lds + o[j] + l * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = combine1<RED, VALUED>(acc[i], vv[j], b[j][i]);
            }
            for (; k < hb; ++k) {
                const int o = s_off[wave][k];
                float vv = 1.0f;
                if constexpr (VALUED) vv = s_val[wave][k];
                const f4 b = *reinterpret_cast<const f4*>(rows_lds + o + l * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = combine1<RED, VALUED>(acc[i], vv, b[i]);
            }
            if (seg == nseg - 1 && colok) {
                float* dst = a.C + (size_t)s_crow[wave][r] * (size_t)a.N + col0;
                f4 o4 = {acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f4*>(dst) = o4;
            }
        }
        wave_sync();  // reads of this record's LDS image precede the next record's writes
    }
}

template <int W, bool VALUED, bool IDX64, int RED>
hipError_t launch_w(const LdsRowArgs& a0, hipStream_t st) {
    LdsRowArgs a = a0;
    a.ntile = (a.N + W * 4 - 1) / (W * 4);
    a.nblk = (a.nrec + kWaves - 1) / kWaves;
    const int64_t nitems = (int64_t)a.nblk * a.ntile;
    if (nitems <= 0) return hipSuccess;
    if (nitems > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((spmm_ldsrow_kernel<W, VALUED, IDX64, RED>), dim3((unsigned)nitems), dim3(kThreads), 0, st, a);
    return hipGetLastError();
}

template <bool VALUED, bool IDX64, int RED>
hipError_t launch_vs(const LdsRowArgs& a, int W, hipStream_t st) {
    switch (W) {
        case 4: return launch_w<4, VALUED, IDX64, RED>(a, st);
        case 8: return launch_w<8, VALUED, IDX64, RED>(a, st);
        case 16: return launch_w<16, VALUED, IDX64, RED>(a, st);
        case 32: return launch_w<32, VALUED, IDX64, RED>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace

int ldsrow_group_width(int64_t N) {
    if (N <= 0 || N % 4 != 0) return 0;
    int W = 4;
    while (W < 32 && (int64_t)W * 4 < N) W <<= 1;
    return W;
}

hipError_t launch_spmm_ldsrow(const LdsRowArgs& a, bool valued, bool idx64, int reduce, hipStream_t st) {
    const int W = ldsrow_group_width(a.N);
    if (W == 0) return hipErrorInvalidValue;
    if (reduce == kReduceMax) {
        if (valued) return hipErrorInvalidValue;
        return idx64 ? launch_vs<false, true, kReduceMax>(a, W, st) : launch_vs<false, false, kReduceMax>(a, W, st);
    }
    if (valued) return idx64 ? launch_vs<true, true, kReduceSum>(a, W, st) : launch_vs<true, false, kReduceSum>(a, W, st);
    return idx64 ? launch_vs<false, true, kReduceSum>(a, W, st) : launch_vs<false, false, kReduceSum>(a, W, st);
}

}  // namespace gespmm
