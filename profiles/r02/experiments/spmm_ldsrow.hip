// spmm_ldsrow.hip — the plan's kernel for row-clustered matrices: distinct B rows of a task staged in LDS by
// persistent, software-pipelined wavefronts.
//
// A clustered plan (plan.cpp) processes rows that share neighbours next to each other. The streaming kernels
// still gather a B row once per USE (one global load per non-zero) and a wavefront walks its task phase after
// phase — CSR, gathers, sums, stores — each phase one exposed memory round trip. Here
//
//   * a task is one fixed-size RECORD the plan writes at analysis time — header, C row ids, the DISTINCT column
//     ids (<= 16), per non-zero its value and the LDS slot of its B row, per row its entry range: the whole
//     description of a task is two coalesced loads at a computable address;
//   * every distinct B row of a record is fetched ONCE, straight into LDS (gfx950 `global_load_lds_dwordx4`: no
//     VGPR staging; 64/W rows per instruction, all of a record's fetches in flight together);
//   * wavefronts are PERSISTENT and pipelined: while record i is summed out of one LDS buffer, the row fetches of
//     record i+1 land in the other and the description of record i+2 is on its way;
//   * records are dealt round-robin to the wavefronts of an XCD inside that XCD's contiguous slice of the clustered
//     order, so at any moment an XCD works on a narrow window of neighbouring clusters — what keeps the shared B
//     rows in its L2; a long row is a CHAIN of records walked by the wavefront that owns the first of them (it
//     carries the accumulator from record to record; the owners of the other records skip them);
//   * each W-lane group of a wavefront takes a contiguous share of the record's rows and walks their entries as one
//     stream (four LDS row reads in flight), storing a row the moment it ends; each output element is ONE fp32 chain
//     over the row's non-zeros in CSR order with one fused multiply-add per non-zero — the arithmetic of every other
//     variant (spmm_test.cu:182-203 semantics), so the bits are unchanged.
//
// Column tiles are 4W floats (W = 4..32 lanes x dwordx4); wider N takes several tiles. N must be a multiple of 4;
// everything else stays on the streaming kernel with the plan's task table.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "spmm_kernels.h"

namespace gespmm {

namespace {

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

struct RecRegs {
    int nrows, ndist, flags;  // wave-uniform
    int w;                    // word 4 + lane of the record: C rows (lanes 0-15), distinct columns (16-31), values (32-63)
    int b;                    // byte: slot of entry `lane` (lanes 0-31), first entry of row `lane - 32` (lanes 32-48)
};

__device__ __forceinline__ void load_record(const int32_t* recs, int idx, int lane, RecRegs& r) {
    const int32_t* rec = recs + (size_t)idx * kRecWords;
    const int4 h = *reinterpret_cast<const int4*>(rec);
    r.nrows = h.x;
    r.ndist = h.z;
    r.flags = h.w;
    r.w = rec[kRecOffCrow + lane];
    r.b = reinterpret_cast<const uint8_t*>(rec)[kRecOffSlotBytes + lane];  // slots then row starts: contiguous bytes
}

template <int W, bool VALUED, bool IDX64, int RED>
__global__ __launch_bounds__(kThreads) void spmm_ldsrow_kernel(LdsRowArgs a) {
    constexpr int G = 64 / W;                 // lane groups per wavefront = B rows fetched per instruction
    constexpr int ROWB = W * 16;              // bytes of one staged row (this column tile)
    constexpr int BUFB = kRecDistinct * ROWB; // one LDS buffer
    using off_t = typename std::conditional<IDX64, uint64_t, uint32_t>::type;

    // ONE __shared__ object, carved by hand (a second object makes hipcc drain the LDS-DMA queue before every LDS read:
    // cdna_hip_programming.md, ".s-level traps"): per wavefront two row buffers, then the record's description
    constexpr int META = kRecEntries * 4 * 2 + 32 * 4 + kRecRows * 4;  // s_off, s_val, s_rp, s_crow
    constexpr int WAVEB = 2 * BUFB + META;
    __shared__ __attribute__((aligned(16))) char s_all[kWaves * WAVEB];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / W;
    const int l = lane % W;
    // grid = 8 XCDs x slots x column tiles; the hardware deals workgroup b to XCD b % 8
    const int xcd = blockIdx.x & 7;
    const int rest = blockIdx.x >> 3;
    const int tile = rest % a.ntile;
    const int slot = rest / a.ntile;
    const int nwx = a.nblk * kWaves;  // wavefronts per XCD (and tile)
    const int lw = slot * kWaves + wave;
    const int x0 = (int)((int64_t)a.nrec * xcd / 8), x1 = (int)((int64_t)a.nrec * (xcd + 1) / 8);

    char* const my_lds = s_all + wave * WAVEB;
    int* const s_off = reinterpret_cast<int*>(my_lds + 2 * BUFB);
    float* const s_val = reinterpret_cast<float*>(my_lds + 2 * BUFB + kRecEntries * 4);
    int* const s_rp = reinterpret_cast<int*>(my_lds + 2 * BUFB + kRecEntries * 8);
    int* const s_crow = reinterpret_cast<int*>(my_lds + 2 * BUFB + kRecEntries * 8 + 32 * 4);

    const int col0 = tile * (W * 4) + l * 4;
    const bool colok = col0 < a.N;  // N % 4 == 0: a lane's four columns are in range together
    const off_t rowbytes = (off_t)a.N * 4u;
    const off_t cbyte = colok ? (off_t)col0 * 4u : (off_t)0;
    const char* Bbase = reinterpret_cast<const char*>(a.B);
    const float init = (RED == kReduceMax) ? a.empty : 0.0f;

    // Records of this XCD's slice are dealt round-robin: wavefront lw owns records x0 + lw, x0 + lw + nwx, ...
    // A chain (one long row) is done by the wavefront that owns its FIRST record, which simply walks on through
    // the following records; the owners of those records skip them (flag bit 0).
    int own0 = x0 + lw;
    if (own0 >= x1) return;

    auto issue_rows = [&](const RecRegs& r, int buf) {
        char* dst = my_lds + buf * BUFB;
        constexpr int T = kRecDistinct / G > 0 ? kRecDistinct / G : 1;  // fetch instructions per record
        int cj[T];
#pragma unroll
        for (int t = 0; t < T; ++t)  // all column ids first (cross-lane reads back to back), then the fetches
            cj[t] = __shfl(r.w, kRecRows + ((t * G + g) & (kRecDistinct - 1)), 64);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t * G < r.ndist) {  // wave-uniform
                if (t * G + g < r.ndist && colok) {
                    const char* src = Bbase + (off_t)((off_t)(uint32_t)cj[t] * rowbytes + cbyte);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + t * G * ROWB), 16, 0, 0);
                }
            }
        }
    };
    // make the header wave-uniform; a continuation record met as an OWN record belongs to somebody else's chain
    auto settle = [&](RecRegs& r, int own, int idx) {
        r.nrows = __builtin_amdgcn_readfirstlane(r.nrows);
        r.ndist = __builtin_amdgcn_readfirstlane(r.ndist);
        r.flags = __builtin_amdgcn_readfirstlane(r.flags);
        if (idx == own && (r.flags & 1)) {
            r.nrows = 0;
            r.ndist = 0;
            r.flags = 0;
        }
    };
    // record after (own, idx): the chain goes on, or the next own record; idx < 0 = nothing left
    auto advance = [&](int& own, int& idx, int flags) {
        if (flags & 2) {
            ++idx;
        } else {
            own += nwx;
            idx = own < x1 ? own : -1;
        }
    };

    RecRegs m0, m1, m2;
    int own_a = own0, idx0 = own0;  // cursor of m0
    load_record(a.recs, idx0, lane, m0);
    settle(m0, own_a, idx0);
    int own_b = own_a, idx1 = idx0;  // cursor of m1
    advance(own_b, idx1, m0.flags);
    if (idx1 >= 0) load_record(a.recs, idx1, lane, m1);
    if (!(a.debug & 1)) issue_rows(m0, 0);

    float carry[4] = {init, init, init, init};  // running sum of a chained row between its records

    int buf = 0;
    for (;;) {
        // ---- everything issued during the previous iteration has had one whole iteration to arrive: the rows of m0 and
        //      the description of m1 (C rows are stored as they finish; the youngest of them are the price of this wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        int own_c = own_b, idx2 = -1;  // cursor of m2
        if (idx1 >= 0) {
            settle(m1, own_b, idx1);
            idx2 = idx1;
            advance(own_c, idx2, m1.flags);
            if (idx2 >= 0) load_record(a.recs, idx2, lane, m2);  // two records ahead
        }
        if (idx1 >= 0 && !(a.debug & 1)) issue_rows(m1, buf ^ 1);
        // ---- the description of m0 moves from the registers that loaded it to LDS, where every lane can index it
        const RecRegs& cur = m0;
        if (lane < kRecEntries) s_off[lane] = (cur.b & (kRecDistinct - 1)) * ROWB;
        else s_rp[lane - 32] = cur.b;  // lanes 32..48 carry the rows' first entries
        if (lane < kRecRows) s_crow[lane] = cur.w;
        if constexpr (VALUED) {
            if (lane >= 32) s_val[lane - 32] = __int_as_float(cur.w);
        }
        wave_sync();
        // ---- sums of m0 out of LDS buffer `buf`: lane group g takes the rows [g*R, (g+1)*R) and walks their entries as ONE
        //      stream, four at a time; a finished row is stored at once
        const char* rows_lds = my_lds + buf * BUFB;
        const bool from_prev = (cur.flags & 1) != 0;  // the (single) row continues from the previous record
        const bool to_next = (cur.flags & 2) != 0;    // ... and into the next one
        const int R = (cur.nrows + G - 1) / G;
        int r = g * R;
        const int r1 = (r + R < cur.nrows) ? r + R : cur.nrows;
        if (r < r1 && !(a.debug & 2)) {
            int k = s_rp[r];
            const int kend = s_rp[r1];
            int end_cur = s_rp[r + 1];
            float acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = from_prev ? carry[i] : init;
            auto finish_row = [&]() {
                if (colok && !(a.debug & 4)) {
                    float* dst = a.C + (size_t)s_crow[r] * (size_t)a.N + col0;
                    f4 o4 = {acc[0], acc[1], acc[2], acc[3]};
                    *reinterpret_cast<f4*>(dst) = o4;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = init;
                ++r;
                end_cur = s_rp[r + 1];  // (r + 1 <= 16: inside the array)
            };
            for (; k < kend; k += 4) {
                int o[4];
                float vv[4];
                f4 bb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kj = (k + j) & (kRecEntries - 1);
                    o[j] = s_off[kj];
                    if constexpr (VALUED) vv[j] = s_val[kj];
                    else vv[j] = 1.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bb[j] = *reinterpret_cast<const f4*>(rows_lds + o[j] + l * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (k + j < kend) {
                        while (k + j >= end_cur) finish_row();  // rows that end before this entry (empty ones included)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = combine1<RED, VALUED>(acc[i], vv[j], bb[j][i]);
                    }
                }
            }
            if (to_next) {  // a chained row: its sum travels on to the next record
#pragma unroll
                for (int i = 0; i < 4; ++i) carry[i] = acc[i];
            } else {
                while (r < r1) finish_row();  // the last row and trailing empty rows
            }
        }
        wave_sync();  // this record's description and rows are read before the next iteration overwrites them
        if (idx1 < 0) break;
        m0 = m1;
        m1 = m2;
        own_a = own_b;
        idx0 = idx1;
        own_b = own_c;
        idx1 = idx2;
        buf ^= 1;
    }
}

template <int W, bool VALUED, bool IDX64, int RED>
hipError_t launch_w(const LdsRowArgs& a0, hipStream_t st) {
    LdsRowArgs a = a0;
    a.ntile = (a.N + W * 4 - 1) / (W * 4);
    // persistent wavefronts: LDS allows 2 workgroups per CU at W = 32 (4 at W = 16, ...): 64 slots per XCD for the
    // widest tile, never more wavefronts than an XCD has units
    int per_xcd = 64 * (32 / W);
    if (per_xcd > 256) per_xcd = 256;
    const int64_t units_per_xcd = (a.nrec + 7) / 8;
    while (per_xcd > 1 && (int64_t)per_xcd * kWaves > units_per_xcd) per_xcd >>= 1;
    a.nblk = per_xcd;
    const int64_t nblocks = (int64_t)8 * per_xcd * a.ntile;
    if (a.nrec <= 0) return hipSuccess;
    if (nblocks > kMaxGridBlocks) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((spmm_ldsrow_kernel<W, VALUED, IDX64, RED>), dim3((unsigned)nblocks), dim3(kThreads), 0, st, a);
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
