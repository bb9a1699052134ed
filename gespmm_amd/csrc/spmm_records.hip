// spmm_records.hip — the padded-record kernel: narrow widths (4 <= N <= 64), short rows, plans only (round 6).
//
// What bounds the streaming kernels at N = 32 on a graph of 5.5 entries per row is not bandwidth (150 MB at 4.4 TB/s) but the
// chain of dependent round trips a wavefront walks per task — task, then row pointers / C rows / CSR tile, then LDS, then U = 4
// gathers per row step, twice per row — with 4 KB of gathers in flight while it is in its gather phase. A plan owns its copy of
// the matrix, so it can lay the entries out the way the wavefront consumes them:
//
//   * a wavefront = G = 64 / W CHAINS (lane groups of W lanes, W * 4 >= N columns); a row is cut into PIECES of P = 8 entry
//     slots (padded), all pieces of a row sit in ONE chain, in order, so the row keeps ONE accumulator and the reference's order
//     of additions (ascending CSR position, one fused multiply-add per entry: spmm_test.cu:182-203) — bit-identical results;
//   * a BATCH = one piece per chain = G headers {C-row byte offset, entries | last << 8} + G * P entries {B-row byte offset,
//     value}: ONE coalesced 8-byte load per lane (lane j of chain q loads entry j) and one for the header; the next batch is
//     requested before the current one is used;
//   * per batch a lane group broadcasts its 8 entries (ds_bpermute: no LDS allocation, no barrier), issues all 8 gathers of
//     16 B per lane back to back (8 KB per wavefront in flight), then the 8 multiply-adds — those past the piece's length are
//     predicated off (their gathers repeat the piece's last address: same cache line) —, then stores the row if the piece
//     was its last;
//   * a TASK = consecutive rows of the plan's order dealt greedily to the chain with the fewest pieces so far, cut by WORK (a task
//     takes rows while the least loaded chain can hold the next one within max(T, its longest chain) batches); wavefront w reads
//     task w = {first batch, #batches}: one scalar load, then the pipeline above. No row pointers, no LDS, no barrier.
//
// Offsets are pre-multiplied 32-bit byte offsets from B / C (the plan knows N): served when max(M, K) * N * 4 < 4 GB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "spmm_device.h"
#include "spmm_kernels.h"

namespace gespmm {

namespace {

constexpr int P = kRecordPiece;

template <int W>
struct RecGeom {
    static constexpr int G = 64 / W;
    static constexpr int kBatchBytes = G * 8 * (1 + P);
};

__device__ __forceinline__ int bperm(int byte_addr, int x) { return __builtin_amdgcn_ds_bpermute(byte_addr, x); }

// A vector of four floats at ANY 4-byte boundary (widths that are not multiples of 4: rows of B and C start where they start).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// STORE: 0 plain, 1 nt, 2 sc1 nt (written through the XCD's L2 and not kept: what the staged kernels do).
// ANYN: the width is not a multiple of 4 — the lane that would reach past the row's end takes the row's LAST four columns instead
// (it recomputes up to three columns of its neighbour: same inputs, same order, same bits — the two stores overlap with equal values),
// and every vector access is 4-byte aligned only (gfx950 serves unaligned global accesses; the compiler is told through the type).
template <int W, int STORE, bool ANYN>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 6))) void spmm_records_kernel(RecordArgs a) {
    constexpr int G = 64 / W;
    constexpr int BB = RecGeom<W>::kBatchBytes;
    constexpr int EPL = (P + W - 1) / W;  // entry slots a lane loads (W = 4: two)
    static_assert(EPL == 1 || EPL == 2, "piece of 8 slots over 4, 8 or 16 lanes");
    if (a.guard != nullptr && *a.guard != a.guard_want) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane / W, l = lane % W;
    const int item = xcd_contiguous(blockIdx.x, gridDim.x);
    const int task = item * kWaves + wave;
    if (task >= a.ntasks) return;
    const int2 t = reinterpret_cast<const int2*>(a.tasks)[task];
    const int first = __builtin_amdgcn_readfirstlane(t.x);
    const int nb = __builtin_amdgcn_readfirstlane(t.y);
    const char* bp = a.batches + (size_t)first * BB;
    const bool colok = l * 4 < a.n;
    uint32_t lbytes = colok ? (uint32_t)l * 16u : 0u;
    if constexpr (ANYN) {
        if (colok && l * 4 + 4 > a.n) lbytes = (uint32_t)(a.n - 4) * 4u;  // (n >= 4)
    }
    const char* Bb = reinterpret_cast<const char*>(a.B);
    char* Cb = reinterpret_cast<char*>(a.C);
    const bool loader = l * EPL < P;                       // (W = 16: half the lanes carry an entry)
    const int eoff = G * 8 + (q * P + (loader ? l * EPL : 0)) * 8;  // this lane's entry slot(s) inside a batch
    const int hoff = q * 8;
    const int qbase = (lane - l) * 4;  // ds_bpermute address of lane 0 of the chain

    int2 hdr = *reinterpret_cast<const int2*>(bp + hoff);
    int ex[EPL], ev[EPL];
    auto load_entries = [&](const char* b, int (&x)[EPL], int (&v)[EPL]) {
        if constexpr (EPL == 1) {
            const int2 e = *reinterpret_cast<const int2*>(b + eoff);
            x[0] = e.x;
            v[0] = e.y;
        } else {
            const int4 e = *reinterpret_cast<const int4*>(b + eoff);
            x[0] = e.x;
            v[0] = e.y;
            x[1] = e.z;
            v[1] = e.w;
        }
    };
    load_entries(bp, ex, ev);

    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < nb; ++i) {
        int2 nhdr = hdr;
        int nx[EPL], nv[EPL];
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            nx[k] = ex[k];
            nv[k] = ev[k];
        }
        if (i + 1 < nb) {  // wave-uniform
            const char* nbp = bp + (size_t)(i + 1) * BB;
            nhdr = *reinterpret_cast<const int2*>(nbp + hoff);
            load_entries(nbp, nx, nv);
        }
        const int len = hdr.y & 0xff;
        uint32_t off[P];
        float v[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int src = qbase + (j / EPL) * 4;
            off[j] = (uint32_t)bperm(src, ex[j % EPL]);
            v[j] = __int_as_float(bperm(src, ev[j % EPL]));
        }
        float bv[P][4];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if constexpr (ANYN) {
                const f4u t4 = *reinterpret_cast<const f4u*>(Bb + (size_t)(off[j] + lbytes));
                bv[j][0] = t4[0];
                bv[j][1] = t4[1];
                bv[j][2] = t4[2];
                bv[j][3] = t4[3];
            } else {
                load_vec<4>(bv[j], Bb + (size_t)(off[j] + lbytes));
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const bool live = j < len;  // (a select, not a branch: a branch lets the compiler sink the gather under it)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t2 = __builtin_fmaf(v[j], bv[j][k], acc[k]);
                acc[k] = live ? t2 : acc[k];
            }
        }
        if (hdr.y & 0x100) {  // the row's last piece
            if (colok) {
                if constexpr (ANYN) {
                    f4u o;
                    o[0] = acc[0];
                    o[1] = acc[1];
                    o[2] = acc[2];
                    o[3] = acc[3];
                    *reinterpret_cast<f4u*>(Cb + (size_t)((uint32_t)hdr.x + lbytes)) = o;
                } else if constexpr (STORE == 2) {
                    using f4 = typename VecT<4>::type;
                    f4 o;
                    o[0] = acc[0];
                    o[1] = acc[1];
                    o[2] = acc[2];
                    o[3] = acc[3];
                    asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" ::"v"((uint32_t)hdr.x + lbytes), "v"(o), "s"(Cb) : "memory");
                } else {
                    store_vec<4, STORE == 1>(reinterpret_cast<float*>(Cb + (size_t)((uint32_t)hdr.x + lbytes)), acc);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = 0.0f;
        }
        hdr = nhdr;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            ex[k] = nx[k];
            ev[k] = nv[k];
        }
    }
}

// ----------------------------------------------------------------------------- tables (plan time, on the device)

// Tasks are cut by WORK, one thread per span of kRecordSpan consecutive rows (tasks do not cross spans): rows are dealt in order to
// the chain with the fewest pieces so far (ties: lowest chain), and a task takes the next row while that chain can hold it without
// raising the task's length beyond max(T, its longest chain) — so every chain ends within one row of the longest, whatever the row
// lengths (a long row opens a task that the following short rows fill up beside it). write == false: tasks per span -> count[span];
// write == true: base[span] = first task id of the span: slot[i] = first piece position of row i inside its task * 16 + chain,
// row_task[i], nb[task] = batches of the task.
constexpr int kRecordSpan = 256;
template <int G>
__global__ void rec_assign_kernel(const int32_t* __restrict__ rowptr, int M, int T, int nspans, bool write, const int32_t* __restrict__ base,
                                  int32_t* __restrict__ count, int32_t* __restrict__ slot, int32_t* __restrict__ row_task,
                                  int32_t* __restrict__ nb) {
    const int sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= nspans) return;
    int lens[G];
#pragma unroll
    for (int c = 0; c < G; ++c) lens[c] = 0;
    const int r0 = sp * kRecordSpan;
    const int r1 = (M - r0 < kRecordSpan) ? M : r0 + kRecordSpan;
    int task = write ? base[sp] : 0;
    int ntasks = 0, mx = 0, rows_in_task = 0;
    int prev = rowptr[r0];
    for (int i = r0; i < r1; ++i) {
        const int next = rowptr[i + 1];
        const int d = next - prev;
        prev = next;
        const int pieces = d > 0 ? (d + P - 1) / P : 1;
        int best = 0, bl = lens[0];
#pragma unroll
        for (int c = 1; c < G; ++c) {
            if (lens[c] < bl) {
                bl = lens[c];
                best = c;
            }
        }
        const int cap = mx > T ? mx : T;
        if (rows_in_task > 0 && bl + pieces > cap) {  // the task is full: close it, the row opens the next one
            if (write) nb[task] = mx;
            ++task;
            ++ntasks;
#pragma unroll
            for (int c = 0; c < G; ++c) lens[c] = 0;
            best = 0;
            bl = 0;
            mx = 0;
            rows_in_task = 0;
        }
        if (write) {
            slot[i] = bl * 16 + best;
            row_task[i] = task;
        }
#pragma unroll
        for (int c = 0; c < G; ++c)
            if (c == best) lens[c] = bl + pieces;
        mx = bl + pieces > mx ? bl + pieces : mx;
        ++rows_in_task;
    }
    if (rows_in_task > 0) {
        if (write) nb[task] = mx;
        ++ntasks;
    }
    if (!write) count[sp] = ntasks;
}

__global__ void rec_tasks_kernel(const int32_t* __restrict__ first, int ntasks, int2* __restrict__ tasks) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntasks) tasks[t] = make_int2(first[t], first[t + 1] - first[t]);
}

// One lane group of 8 per row: lane j writes entry slot j of every piece of the row; lane 0 the header. The array was zeroed:
// an untouched slot is {offset 0, value 0, 0 entries, not last}.
template <int G>
__global__ void rec_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
                                const int32_t* __restrict__ perm, const int32_t* __restrict__ slot, const int32_t* __restrict__ row_task,
                                const int32_t* __restrict__ first, int M, uint32_t rowbytes, char* __restrict__ batches, bool values_only) {
    constexpr int BB = G * 8 * (1 + P);
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / P;
    const int j = threadIdx.x % P;
    if (gid >= M) return;
    const int i = gid;
    const int lb = rowptr[i], hb = rowptr[i + 1];
    const int d = hb - lb;
    const int pieces = d > 0 ? (d + P - 1) / P : 1;
    const int s = slot[i];
    const int c = s & 15;
    char* b = batches + (size_t)(first[row_task[i]] + (s >> 4)) * BB;
    const uint32_t crow = (uint32_t)(perm ? perm[i] : i) * rowbytes;
    for (int k = 0; k < pieces; ++k, b += BB) {
        const int plen = (d - k * P < P) ? d - k * P : P;
        if (j == 0 && !values_only) *reinterpret_cast<int2*>(b + c * 8) = make_int2((int)crow, plen | (k == pieces - 1 ? 0x100 : 0));
        if (plen > 0) {
            const int pos = lb + k * P + (j < plen ? j : plen - 1);
            int2 e;
            e.x = (int)((uint32_t)colind[pos] * rowbytes);
            e.y = j < plen ? __float_as_int(val ? val[pos] : 1.0f) : 0;
            *reinterpret_cast<int2*>(b + G * 8 + (c * P + j) * 8) = e;
        }
    }
}

template <int G>
hipError_t build_g(int64_t M, const int32_t* rowptr, const int32_t* colind, const float* val, const int32_t* perm, int T, int64_t N,
                   RecordTables* out, hipStream_t st) {
    constexpr int BB = G * 8 * (1 + P);
    const int64_t nspans = (M + kRecordSpan - 1) / kRecordSpan;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // side block (kept: gespmm_plan_set_values refills the stream): slot[M], row_task[M], first[M + 1]; scratch behind it: nb[M + 1],
    // per-span counts and bases, the scans' temporary storage
    size_t scan_a = 0, scan_b = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_a, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)nspans + 1,
                                           rocprim::plus<int32_t>(), st);
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(nullptr, scan_b, (const int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)M + 1, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return e;
    const size_t scan_bytes = scan_a > scan_b ? scan_a : scan_b;
    const size_t b_row = up((size_t)M * 4), b_row1 = up(((size_t)M + 1) * 4), b_span = up(((size_t)nspans + 1) * 4);
    char* side = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&side), 2 * b_row + 2 * b_row1 + 2 * b_span + up(scan_bytes) + 256);
    if (e != hipSuccess) return e;
    int32_t* slot = reinterpret_cast<int32_t*>(side);
    int32_t* row_task = reinterpret_cast<int32_t*>(side + b_row);
    int32_t* first = reinterpret_cast<int32_t*>(side + 2 * b_row);
    int32_t* nb = reinterpret_cast<int32_t*>(side + 2 * b_row + b_row1);
    int32_t* count = reinterpret_cast<int32_t*>(side + 2 * b_row + 2 * b_row1);
    int32_t* base = reinterpret_cast<int32_t*>(side + 2 * b_row + 2 * b_row1 + b_span);
    void* scan_tmp = side + 2 * b_row + 2 * b_row1 + 2 * b_span;
    const dim3 sgrid((unsigned)((nspans + 63) / 64)), sblock(64);
    e = hipMemsetAsync(nb, 0, ((size_t)M + 1) * 4, st);  // (tasks beyond the last one: no batches — the scan below runs over M + 1 counts)
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, ((size_t)nspans + 1) * 4, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((rec_assign_kernel<G>), sgrid, sblock, 0, st, rowptr, (int)M, T, (int)nspans, false, (const int32_t*)nullptr, count,
                           (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(scan_tmp, scan_a, (const int32_t*)count, base, 0, (size_t)nspans + 1, rocprim::plus<int32_t>(), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((rec_assign_kernel<G>), sgrid, sblock, 0, st, rowptr, (int)M, T, (int)nspans, true, (const int32_t*)base, count, slot,
                           row_task, nb);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(scan_tmp, scan_b, (const int32_t*)nb, first, 0, (size_t)M + 1, rocprim::plus<int32_t>(), st);
    int32_t totals[2] = {0, 0};  // tasks, batches
    if (e == hipSuccess) e = hipMemcpyAsync(&totals[0], base + nspans, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&totals[1], first + M, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const int64_t ntasks = totals[0], total = totals[1];
    if (e == hipSuccess && (ntasks <= 0 || total < 0 || total * BB > (int64_t)kRecordMaxBytes)) e = hipErrorOutOfMemory;
    char* main_block = nullptr;
    const size_t b_tasks = up((size_t)(ntasks > 0 ? ntasks : 1) * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&main_block), b_tasks + (size_t)total * BB + 256);
    if (e == hipSuccess) e = hipMemsetAsync(main_block + b_tasks, 0, (size_t)total * BB, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rec_tasks_kernel, dim3((unsigned)((ntasks + 255) / 256)), dim3(256), 0, st, first, (int)ntasks,
                           reinterpret_cast<int2*>(main_block));
        hipLaunchKernelGGL((rec_fill_kernel<G>), dim3((unsigned)((M * P + 255) / 256)), dim3(256), 0, st, rowptr, colind, val, perm, slot,
                           row_task, first, (int)M, (uint32_t)(N * 4), main_block + b_tasks, false);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        (void)hipFree(side);
        if (main_block) (void)hipFree(main_block);
        return e;
    }
    out->block = main_block;
    out->side = side;
    out->tasks = reinterpret_cast<int32_t*>(main_block);
    out->batches = main_block + b_tasks;
    out->slot = slot;
    out->row_task = row_task;
    out->first = first;
    out->ntasks = (int32_t)ntasks;
    out->nbatches = (int32_t)total;
    out->target_batches = T;
    out->group = 64 / G;
    out->N = N;
    return hipSuccess;
}

}  // namespace

int records_group(int64_t N) {
    if (N < 4 || N > 64) return 0;
    return N <= 16 ? 4 : (N <= 32 ? 8 : 16);
}

bool records_serves(int64_t M, int64_t K, int64_t N, int32_t max_degree) {
    if (!records_group(N) || M <= 0) return false;
    const int64_t rows = M > K ? M : K;
    return rows * N * 4 < (1ll << 32) && max_degree <= kRecordMaxRow;
}

hipError_t device_build_records(int64_t M, const int32_t* rowptr, const int32_t* colind, const float* val, const int32_t* perm,
                                int target_batches, int64_t N, RecordTables* out, hipStream_t st) {
    const int W = records_group(N);
    if (!W || M <= 0 || target_batches < 1) return hipErrorInvalidValue;
    switch (W) {
        case 4: return build_g<16>(M, rowptr, colind, val, perm, target_batches, N, out, st);
        case 8: return build_g<8>(M, rowptr, colind, val, perm, target_batches, N, out, st);
        default: return build_g<4>(M, rowptr, colind, val, perm, target_batches, N, out, st);
    }
}

hipError_t device_records_set_values(const RecordTables& t, int64_t M, const int32_t* rowptr, const int32_t* colind, const float* val,
                                     hipStream_t st) {
    if (!t.batches) return hipSuccess;
    const dim3 grid((unsigned)((M * P + 255) / 256));
    const uint32_t rowbytes = (uint32_t)(t.N * 4);
    switch (t.group) {
        case 4: hipLaunchKernelGGL((rec_fill_kernel<16>), grid, dim3(256), 0, st, rowptr, colind, val, (const int32_t*)nullptr, t.slot, t.row_task, t.first, (int)M, rowbytes, t.batches, true); break;
        case 8: hipLaunchKernelGGL((rec_fill_kernel<8>), grid, dim3(256), 0, st, rowptr, colind, val, (const int32_t*)nullptr, t.slot, t.row_task, t.first, (int)M, rowbytes, t.batches, true); break;
        default: hipLaunchKernelGGL((rec_fill_kernel<4>), grid, dim3(256), 0, st, rowptr, colind, val, (const int32_t*)nullptr, t.slot, t.row_task, t.first, (int)M, rowbytes, t.batches, true); break;
    }
    return hipGetLastError();
}

void free_records(RecordTables* t) {
    if (t->block) (void)hipFree(t->block);
    if (t->side) (void)hipFree(t->side);
    *t = RecordTables();
}

hipError_t launch_spmm_records(const RecordTables& t, const float* B, float* C, int64_t N, int flags, const LaunchGuard* guard,
                               hipStream_t st) {
    if (!t.batches || N != t.N) return hipErrorInvalidValue;
    RecordArgs a;
    a.tasks = t.tasks;
    a.batches = t.batches;
    a.B = B;
    a.C = C;
    a.ntasks = t.ntasks;
    a.n = (int32_t)N;
    a.guard = guard ? guard->word : nullptr;
    a.guard_want = guard ? guard->want : 0;
    const unsigned nwg = (unsigned)((t.ntasks + kWaves - 1) / kWaves);
    if (nwg == 0) return hipSuccess;
    static const int env_store = getenv("GESPMM_REC_STORE") ? atoi(getenv("GESPMM_REC_STORE")) : -1;  // experiments
    const int store = env_store >= 0 ? env_store : ((flags & kFlagNtStore) ? 1 : 2);
#define GESPMM_REC_LAUNCH(W_)                                                                                               \
    do {                                                                                                                    \
        if (N % 4 != 0) hipLaunchKernelGGL((spmm_records_kernel<W_, 0, true>), dim3(nwg), dim3(kThreads), 0, st, a);        \
        else if (store == 2) hipLaunchKernelGGL((spmm_records_kernel<W_, 2, false>), dim3(nwg), dim3(kThreads), 0, st, a);  \
        else if (store == 1) hipLaunchKernelGGL((spmm_records_kernel<W_, 1, false>), dim3(nwg), dim3(kThreads), 0, st, a);  \
        else hipLaunchKernelGGL((spmm_records_kernel<W_, 0, false>), dim3(nwg), dim3(kThreads), 0, st, a);                  \
    } while (0)
    switch (t.group) {
        case 4: GESPMM_REC_LAUNCH(4); break;
        case 8: GESPMM_REC_LAUNCH(8); break;
        case 16: GESPMM_REC_LAUNCH(16); break;
        default: return hipErrorInvalidValue;
    }
#undef GESPMM_REC_LAUNCH
    return hipGetLastError();
}

}  // namespace gespmm
