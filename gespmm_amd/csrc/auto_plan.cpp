// auto_plan.cpp — opt-in plan reuse behind the STATELESS boundaries, and the library's warm-up (round 6).
//
// The reference's callers keep no state across products: spmmWrapper (spmm_test.cu:456-492), spmm_cuda (spmm_kernel.cu:425-458) and the
// DGL patch's CustomCsrmm (dgl-custom/binary_reduce_sum.cu:338-360) take the CSR arrays and launch. A caller shaped like that cannot
// hold a gespmm_plan, so on a graph with structure it stays at the plain kernels' rate (0.30 of the roofline on the headline graph
// against 0.50 through a plan). gespmm_set_auto_plan(k) lets the library keep the plan instead:
//
//   * a small cache keyed on what the caller passes — device, the rowptr / colind pointers, M, K, N, valued?, variant, reducer;
//   * pointer identity is not pattern identity, so every call that would use a cached plan first runs a FINGERPRINT of the arrays on
//     the device (one kernel: position-mixed 64-bit sums over ALL of rowptr and colind, separately over the values; rowptr[M]; the
//     largest column) and reads 32 bytes back — one stream synchronisation, which the DGL entry points perform anyway to learn nnz.
//     A pattern that changed in place drops the plan (the call runs the plain kernels, the count starts again); values that changed
//     are re-permuted (gespmm_plan_set_values) before the launch;
//   * the plan is made at the k-th call with the same key (k >= 1), synchronously (3-4 ms on a com-Amazon-sized graph after
//     gespmm_init, section "cold plans" of DESIGN.md); earlier calls run the plain kernels;
//   * never on a capturing stream (no synchronisation there), never when the switch is off (the default: one relaxed atomic load on
//     the plain path, nothing else).
//
// A plan changes the ORDER rows are processed in, not the sums: the bits are the plain call's (tests/test_gpu_auto_plan.py).

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/gespmm.h"
#include "auto_plan.h"
#include "plan.h"
#include "spmm_kernels.h"

namespace gespmm {

namespace {

constexpr int kAutoEntries = 8;
constexpr int kAutoDevices = 16;

struct AutoEntry {
    bool used = false;
    int device = 0;
    const int32_t* rowptr = nullptr;
    const int32_t* colind = nullptr;
    int64_t M = 0, K = 0, N = 0;
    bool valued = false;
    int variant = 0, reduce = 0;
    int count = 0;  // calls seen with this key since the entry was (re)started
    gespmm_plan* plan = nullptr;
    bool no_gain = false;  // the analysis kept the storage order (or was not worth its cost): the plain path is the plan's own launch
    const float* val_seen = nullptr;  // the values the plan was last given (pointer; their content is fingerprinted)
    unsigned long long fp_pattern = 0, fp_values = 0;
    int64_t nnz = 0;
    unsigned long long stamp = 0;  // last use (LRU)
    // asynchronous mode (below): the plan's launch and the plain launch are one kernel each, so both can sit behind a device-side guard
    bool async_ok = false;
    int32_t* guard_word = nullptr;              // device: 1 = the arrays still have the plan's fingerprint (written by k_fingerprint_check)
    unsigned long long* rec = nullptr;          // pinned + mapped host record {seq, pattern, values, nnz, match} of the latest finished check
    unsigned long long* rec_dev = nullptr;
    unsigned long long seq_launched = 0, seq_seen = 0;
};

std::atomic<int> g_auto_k{0};
std::atomic<bool> g_auto_async{true};  // GESPMM_AUTO_PLAN_SYNC=1: every planned call takes the synchronous fingerprint (experiments)
std::mutex g_lock;
AutoEntry g_entries[kAutoEntries];
unsigned long long g_clock = 0;
gespmm_auto_plan_stats g_stats = {0, 0, 0, 0, 0, 0, 0};

constexpr int kFpSlots = 64;  // partial fingerprints: workgroup b adds into slot b % 64 (same-address atomics serialise at ~9 ns each)

struct FpScratch {
    unsigned long long* dev = nullptr;   // [kFpSlots][4] accumulators on the device
    unsigned long long* host = nullptr;  // the same, pinned
};
FpScratch g_fp[kAutoDevices];

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// One thread's share of the fingerprint: position-mixed sums over rowptr (tagged) and colind -> hp, over the value bits -> hv, the largest
// column + 1 -> mx. Grid-stride over 16-byte vectors where the arrays are 16-byte aligned (four independent loads' worth per step: with
// one element per step the kernel was a chain of ~28 dependent round trips per thread, 15-20 us on the headline graph — most of what the
// switch cost per call), element-wise otherwise and for the tails.
__device__ __forceinline__ void fp_accumulate(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
                                              long long M, long long nnz, long long tid, long long stride, unsigned long long& hp,
                                              unsigned long long& hv, unsigned int& mx) {
    constexpr unsigned long long kRowTag = 0xa5ull << 56;  // a row pointer is not a column
    auto vec_pass = [&](const int32_t* __restrict__ a, long long n, unsigned long long tag, unsigned long long& acc, bool track_max) {
        const bool aligned = (reinterpret_cast<uintptr_t>(a) & 15) == 0;
        const long long nv = aligned ? n / 4 : 0;
        const int4* __restrict__ a4 = reinterpret_cast<const int4*>(a);
        for (long long j = tid; j < nv; j += stride) {
            const int4 c = a4[j];
            const unsigned long long p = (unsigned long long)(4 * j) << 32;
            acc += mix64(p ^ (unsigned int)c.x ^ tag) + mix64((p + (1ull << 32)) ^ (unsigned int)c.y ^ tag) +
                   mix64((p + (2ull << 32)) ^ (unsigned int)c.z ^ tag) + mix64((p + (3ull << 32)) ^ (unsigned int)c.w ^ tag);
            if (track_max) {
                unsigned int m = (unsigned int)c.x > (unsigned int)c.y ? (unsigned int)c.x : (unsigned int)c.y;
                const unsigned int m2 = (unsigned int)c.z > (unsigned int)c.w ? (unsigned int)c.z : (unsigned int)c.w;
                m = m > m2 ? m : m2;
                mx = m + 1u > mx ? m + 1u : mx;
            }
        }
        for (long long i = 4 * nv + tid; i < n; i += stride) {
            const unsigned int c = (unsigned int)a[i];
            acc += mix64(((unsigned long long)i << 32) ^ c ^ tag);
            if (track_max) mx = c + 1u > mx ? c + 1u : mx;
        }
    };
    vec_pass(rowptr, M + 1, kRowTag, hp, false);
    vec_pass(colind, nnz, 0ull, hp, true);
    if (val) vec_pass(reinterpret_cast<const int32_t*>(val), nnz, 0ull, hv, false);
}

// slot[0] += sum over rowptr and colind of mix(position-tagged word), slot[1] += the same over the value bits, slot[2] = rowptr[M]
// (slot 0 only), slot[3] = largest column index + 1. Sums of mixed terms: independent of the order the workgroups run in. One set of
// atomics per WORKGROUP, spread over kFpSlots addresses; the host adds the slots up. What was measured on the headline graph
// (profiles/r06/auto_plan_timing.log, wall clock per planned call; the plan held by the caller: 88-90 us): one set of atomics per
// wavefront on ONE address (24 000 same-address atomics) 230-293 us; this form (memset + kernel + 2 KB copy + synchronise) 113-115 us;
// every workgroup storing its partials into mapped host memory (no memset, no copy) 120-123 us; a last-workgroup ticket that stores one
// 32-byte result into mapped host memory 122-126 us — stores to host memory make the synchronisation itself slower than a copy does.
__global__ void __launch_bounds__(256) k_fingerprint(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                     const float* __restrict__ val, long long M, unsigned long long* __restrict__ out) {
    const long long nnz = rowptr[M];
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long hp = 0, hv = 0;
    unsigned int mx = 0;
    fp_accumulate(rowptr, colind, val, M, nnz, tid, stride, hp, hv, mx);
    for (int o = 32; o > 0; o >>= 1) {
        hp += __shfl_down(hp, o);
        hv += __shfl_down(hv, o);
        const unsigned int m2 = __shfl_down(mx, o);
        mx = m2 > mx ? m2 : mx;
    }
    __shared__ unsigned long long s_hp[4], s_hv[4];
    __shared__ unsigned int s_mx[4];
    if ((threadIdx.x & 63) == 0) {
        s_hp[threadIdx.x >> 6] = hp;
        s_hv[threadIdx.x >> 6] = hv;
        s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* slot = out + 4 * (blockIdx.x & (kFpSlots - 1));
        atomicAdd(&slot[0], s_hp[0] + s_hp[1] + s_hp[2] + s_hp[3]);
        if (val) atomicAdd(&slot[1], s_hv[0] + s_hv[1] + s_hv[2] + s_hv[3]);
        unsigned int m = s_mx[0];
        for (int w = 1; w < 4; ++w) m = s_mx[w] > m ? s_mx[w] : m;
        atomicMax(&slot[3], (unsigned long long)m);
        if (blockIdx.x == 0) slot[2] = (unsigned long long)nnz;
    }
}

// Fingerprint of the caller's arrays on `st`, read back (ONE synchronisation). fp = {pattern, values, nnz, columns}.
hipError_t fingerprint(int dev, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M, hipStream_t st,
                       unsigned long long fp[4]) {
    if (dev < 0 || dev >= kAutoDevices) return hipErrorInvalidDevice;
    FpScratch& s = g_fp[dev];
    constexpr size_t kBytes = (size_t)kFpSlots * 4 * 8;
    if (!s.dev) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s.dev), kBytes);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s.host), kBytes, hipHostMallocDefault);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMemsetAsync(s.dev, 0, kBytes, st);
    if (e != hipSuccess) return e;
    // two workgroups per CU stream the arrays at the memory system's rate (grid-stride: coalesced)
    long long blocks = (M + 256) / 256;
    if (blocks < kFpSlots) blocks = kFpSlots;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_fingerprint, dim3((unsigned)blocks), dim3(256), 0, st, rowptr, colind, val, (long long)M, s.dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(s.host, s.dev, kBytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    fp[0] = fp[1] = fp[3] = 0;
    fp[2] = s.host[2];
    for (int i = 0; i < kFpSlots; ++i) {
        fp[0] += s.host[4 * i];
        fp[1] += s.host[4 * i + 1];
        if (s.host[4 * i + 3] > fp[3]) fp[3] = s.host[4 * i + 3];
    }
    return hipSuccess;
}

// ---- asynchronous mode -------------------------------------------------------------------------------------------------------------
// The synchronisation above costs the planned call ~25 us (the GPU idles while the host wakes up and launches). Where the plan's launch
// and the plain launch are ONE kernel each (no long-row pass, no cache blocking: the headline graph), nothing has to come back to the
// host: k_fingerprint_check computes the same fingerprint, compares it ON THE DEVICE with what the plan was made from and writes a guard
// word; the plan's kernel is launched behind "guard == 1", the plain kernel behind "guard == 0" — exactly one of them runs, the other's
// workgroups leave at once. The check also stores {seq, fingerprint, match} into mapped host memory, which the NEXT call reads without
// waiting: a changed pattern drops the plan then, changed values are re-permuted then. Correctness never depends on what the host
// knows — the device decides every launch.
constexpr int kFpCheckBlocks = 256;
constexpr int kFpCheckSlots = 32;
constexpr size_t kGuardBytes = 256;  // the guard word's share of an entry's device block; behind it: [kFpCheckSlots][4] partial sums + ticket

__global__ void __launch_bounds__(256) k_fingerprint_check(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                           const float* __restrict__ val, long long M, unsigned long long* __restrict__ slots,
                                                           unsigned long long want_pattern, unsigned long long want_values, long long want_nnz,
                                                           int32_t* __restrict__ guard_word, unsigned long long* __restrict__ rec,
                                                           unsigned long long seq) {
    const long long nnz = rowptr[M];
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    unsigned long long hp = 0, hv = 0;
    unsigned int mx_unused = 0;
    fp_accumulate(rowptr, colind, val, M, nnz, tid, stride, hp, hv, mx_unused);
    for (int o = 32; o > 0; o >>= 1) {
        hp += __shfl_down(hp, o);
        hv += __shfl_down(hv, o);
    }
    __shared__ unsigned long long s_hp[4], s_hv[4];
    __shared__ bool s_last;
    if ((threadIdx.x & 63) == 0) {
        s_hp[threadIdx.x >> 6] = hp;
        s_hv[threadIdx.x >> 6] = hv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* slot = slots + 4 * (blockIdx.x & (kFpCheckSlots - 1));
        atomicAdd(&slot[0], s_hp[0] + s_hp[1] + s_hp[2] + s_hp[3]);
        if (val) atomicAdd(&slot[1], s_hv[0] + s_hv[1] + s_hv[2] + s_hv[3]);
        __threadfence();
        s_last = atomicAdd(&slots[4 * kFpCheckSlots], 1ull) == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (s_last && threadIdx.x < 64) {  // the last workgroup to finish: one wavefront adds the slots up, decides, and clears them
        __threadfence();
        const int l = threadIdx.x;
        unsigned long long a0 = 0, a1 = 0;
        if (l < kFpCheckSlots) {
            a0 = __hip_atomic_load(&slots[4 * l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a1 = __hip_atomic_load(&slots[4 * l + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slots[4 * l] = 0;
            slots[4 * l + 1] = 0;
        }
        for (int o = 32; o > 0; o >>= 1) {
            a0 += __shfl_down(a0, o);
            a1 += __shfl_down(a1, o);
        }
        if (l == 0) {
            slots[4 * kFpCheckSlots] = 0;
            const bool match = a0 == want_pattern && nnz == want_nnz && (!val || a1 == want_values);
            *guard_word = match ? 1 : 0;
            rec[1] = a0;
            rec[2] = a1;
            rec[3] = (unsigned long long)nnz;
            rec[4] = match ? 1ull : 0ull;
            __threadfence_system();
            __hip_atomic_store(&rec[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t async_setup(AutoEntry& en, int dev) {
    if (dev < 0 || dev >= kAutoDevices) return hipErrorInvalidDevice;
    // per ENTRY: guard word + the check's partial sums and ticket (zero between launches: the last workgroup clears them) — two keys on
    // two streams may have their checks in flight together
    constexpr size_t kBytes = kGuardBytes + ((size_t)kFpCheckSlots * 4 + 1) * 8;
    if (!en.guard_word) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&en.guard_word), kBytes);
        if (e == hipSuccess) e = hipMemset(en.guard_word, 0, kBytes);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&en.rec), 64, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void**>(&en.rec_dev), en.rec, 0);
        if (e != hipSuccess) return e;
        std::memset(en.rec, 0, 64);
    }
    // records of checks launched for an earlier plan of this key are history (drop_plan has waited for them): sequence numbers go on
    en.seq_seen = en.seq_launched;
    return hipSuccess;
}

void drop_plan(AutoEntry& en) {
    if (en.plan) gespmm_plan_destroy(en.plan);  // (hipFree inside: every launch that used the plan has finished when it returns)
    en.plan = nullptr;
    en.async_ok = false;
}

void release_entry(AutoEntry& en) {
    drop_plan(en);
    if (en.guard_word) (void)hipFree(en.guard_word);
    if (en.rec) (void)hipHostFree(en.rec);
    en.guard_word = nullptr;
    en.rec = en.rec_dev = nullptr;
}

}  // namespace

bool auto_plan_enabled() { return g_auto_k.load(std::memory_order_relaxed) > 0; }

// Returns true when the call was served (then *rc is its result); false = the caller runs the plain path.
bool auto_plan_try(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C, int64_t M, int64_t K,
                   int64_t N, int64_t nnz_arg, int variant, int reduce, float empty, void* stream, int* rc) {
    const int k = g_auto_k.load(std::memory_order_relaxed);
    if (k <= 0 || M <= 0 || N <= 0 || !rowptr || !colind || !B || !C) return false;
    if (reduce == kReduceMax && val) return false;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;  // no synchronisation on a capturing stream: stateless
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;

    std::lock_guard<std::mutex> guard(g_lock);
    AutoEntry* en = nullptr;
    AutoEntry* victim = &g_entries[0];
    for (AutoEntry& e : g_entries) {
        if (e.used && e.device == dev && e.rowptr == rowptr && e.colind == colind && e.M == M && e.K == K && e.N == N &&
            e.valued == (val != nullptr) && e.variant == variant && e.reduce == reduce) {
            en = &e;
            break;
        }
        if (!e.used || (victim->used && e.stamp < victim->stamp)) victim = &e;
    }
    if (!en) {
        // a new key: remember it (the least recently used entry goes) and run the plain kernels
        release_entry(*victim);
        *victim = AutoEntry();
        victim->used = true;
        victim->device = dev;
        victim->rowptr = rowptr;
        victim->colind = colind;
        victim->M = M;
        victim->K = K;
        victim->N = N;
        victim->valued = val != nullptr;
        victim->variant = variant;
        victim->reduce = reduce;
        en = victim;
    }
    en->stamp = ++g_clock;
    en->count += 1;
    if (en->no_gain || (!en->plan && en->count < k)) return false;
    const bool dgl = K <= 0;  // (the DGL entry points pass neither K nor nnz: capi.cpp)
    // the plain launch behind "guard == 0" almost never runs: the largest tasks the kernels take (32 rows per wavefront / lane group) make
    // its grid — whose workgroups all have to be dispatched just to leave — an eighth of the usual one (same bits whatever the geometry)
    gespmm_launch_cfg fallback_cfg = {0, 0, 0, kMaxRowsPerWave, 0, dgl ? GESPMM_FLAG_FORCE_IDX64 : 0};

    // ---- asynchronous mode: nothing comes back to the host before the launch (see k_fingerprint_check)
    if (en->plan && en->async_ok && val == en->val_seen) {
        // what the checks of earlier calls found (whatever has landed; seqlock: seq, fields, seq again)
        const unsigned long long done = __atomic_load_n(&en->rec[0], __ATOMIC_ACQUIRE);
        if (done > en->seq_seen) {
            const unsigned long long r_pat = en->rec[1], r_val = en->rec[2], r_nnz = en->rec[3], r_match = en->rec[4];
            if (__atomic_load_n(&en->rec[0], __ATOMIC_ACQUIRE) == done) {
                en->seq_seen = done;
                if (!r_match) {
                    if (r_pat != en->fp_pattern || (int64_t)r_nnz != en->nnz) {
                        drop_plan(*en);  // the pattern changed under the same pointers (the calls since then ran the plain kernel)
                        en->count = 1;
                        g_stats.invalidated += 1;
                        if (en->count < k) return false;
                    } else if (val) {
                        if (gespmm_plan_set_values(en->plan, val, stream) != 0) {
                            drop_plan(*en);
                            en->count = 0;
                            return false;
                        }
                        en->fp_values = r_val;
                        g_stats.values_refreshed += 1;
                    }
                }
            }
        }
        if (en->plan && en->async_ok) {
            const unsigned long long seq = ++en->seq_launched;
            hipLaunchKernelGGL(k_fingerprint_check, dim3(kFpCheckBlocks), dim3(256), 0, st, rowptr, colind, val, (long long)M,
                               reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(en->guard_word) + kGuardBytes), en->fp_pattern, en->fp_values, (long long)en->nnz, en->guard_word, en->rec_dev, seq);
            if (hipGetLastError() == hipSuccess) {
                const LaunchGuard run_plan = {en->guard_word, 1}, run_plain = {en->guard_word, 0};
                int r1 = plan_spmm_guarded(en->plan, B, C, N, reduce, empty, stream, &run_plan);
                int r0 = r1;
                if (r1 == 0)
                    r0 = run_spmm(rowptr, colind, val, B, C, M, dgl ? M : K, N, en->nnz, variant, &fallback_cfg, reduce, empty, stream, nullptr, 0,
                                  nullptr, &run_plain);
                if (r1 == 0 && r0 == 0) {
                    g_stats.calls_planned += 1;
                    g_stats.calls_async += 1;
                    *rc = 0;
                    return true;
                }
                // (cannot happen after the dry run at plan creation; if it does, the synchronous path below recomputes everything)
                en->async_ok = false;
                (void)hipStreamSynchronize(st);
            }
            (void)hipGetLastError();
        }
    }

    unsigned long long fp[4] = {0, 0, 0, 0};
    hipError_t e = fingerprint(dev, rowptr, colind, val, M, st, fp);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    g_stats.fingerprints += 1;
    const int64_t nnz = (int64_t)fp[2];
    if (nnz_arg >= 0 && nnz_arg != nnz) {  // the caller's nnz contradicts rowptr[M]: let the plain path report what it reports
        drop_plan(*en);
        en->count = 0;
        return false;
    }
    if (en->plan && (fp[0] != en->fp_pattern || nnz != en->nnz)) {
        // the pattern changed under the same pointers: the plan is void
        drop_plan(*en);
        en->count = 1;
        g_stats.invalidated += 1;
        if (en->count < k) return false;
    }
    if (!en->plan) {
        // the caller asked for plans: the first one of the process does not start cold (the analysis kernels are loaded and the arena is
        // made now, once; a cold AUTO plan would price that into its cost rule and decline)
        if (!analysis_is_warm() && gespmm_plan_wants_warmup(M, K > 0 && K < 0x7fffffffLL ? K : M, nnz, N, 0) == 1) (void)gespmm_init(M, nnz, stream);
        // columns: the caller's K where it is a real bound, else what the arrays hold (the DGL entry points do not know K)
        int64_t Kp = K;
        if (Kp <= 0 || Kp >= 0x7fffffffLL) Kp = (int64_t)fp[3] > M ? (int64_t)fp[3] : M;
        else if ((int64_t)fp[3] > Kp) return false;  // a column outside [0, K): the plain path's business
        gespmm_plan_options opt;
        std::memset(&opt, 0, sizeof opt);
        opt.expected_launches = 0;  // the library's default: the reference's 200
        gespmm_plan* p = nullptr;
        const int prc = gespmm_plan_create_v2(&p, rowptr, colind, val, M, Kp, nnz, N, variant, &opt, (int64_t)sizeof opt, stream);
        if (prc != 0 || !p) {
            (void)hipGetLastError();
            en->count = 0;  // (not again on the next call: the count starts over)
            return false;
        }
        g_stats.plans_created += 1;
        if (!plan_is_clustered(p)) {
            // nothing to reuse: the plan's launch IS the plain call's. No fingerprint, no synchronisation from here on.
            gespmm_plan_destroy(p);
            en->no_gain = true;
            return false;
        }
        en->plan = p;
        en->fp_pattern = fp[0];
        en->fp_values = fp[1];
        en->nnz = nnz;
        en->val_seen = val;
        // asynchronous mode from the next call on, if both launches are single kernels (dry runs: nothing is launched)
        const LaunchGuard dry = {nullptr, -1};
        en->async_ok = g_auto_async.load(std::memory_order_relaxed) && plan_spmm_guarded(p, B, C, N, reduce, empty, stream, &dry) == 0 &&
                       run_spmm(rowptr, colind, val, B, C, M, dgl ? M : K, N, nnz, variant, &fallback_cfg, reduce, empty, stream, nullptr, 0, nullptr,
                                &dry) == 0 &&
                       async_setup(*en, dev) == hipSuccess;
        (void)hipGetLastError();
    } else if (val && (fp[1] != en->fp_values || val != en->val_seen)) {
        const int src = gespmm_plan_set_values(en->plan, val, stream);
        if (src != 0) {
            drop_plan(*en);
            en->count = 0;
            return false;
        }
        en->fp_values = fp[1];
        en->val_seen = val;
        g_stats.values_refreshed += 1;
    }
    *rc = reduce == kReduceMax ? gespmm_plan_spmm_max_f32(en->plan, B, C, N, empty, stream) : gespmm_plan_spmm_f32(en->plan, B, C, N, stream);
    g_stats.calls_planned += 1;
    return true;
}

}  // namespace gespmm

namespace gespmm {
namespace {

// gespmm_init's matrix: 16 communities of 1 024 rows, every row six neighbours inside its community and two anywhere — small, and the
// clustering, the model, the task cutting and the staging tables all have something to do
__global__ void k_init_matrix(int32_t* __restrict__ rowptr, int32_t* __restrict__ colind, int M, int deg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= M) rowptr[i] = i * deg;
    if (i >= M) return;
    const int base = i & ~1023;
    for (int j = 0; j < deg; ++j) {
        const unsigned int h = (unsigned int)mix64(((unsigned long long)i << 8) | (unsigned int)j);
        colind[i * deg + j] = j < deg - 2 ? base + (int)(h & 1023u) : (int)(h % (unsigned int)M);
    }
}

std::atomic<unsigned int> g_init_done{0};  // bit d: gespmm_init has run on device d

}  // namespace
}  // namespace gespmm

extern "C" {

int gespmm_init(int64_t rows_hint, int64_t nnz_hint, void* stream) {
    if (rows_hint < 0 || nnz_hint < 0) return GESPMM_EINVAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return GESPMM_EINVAL;
    const unsigned int bit = dev >= 0 && dev < 32 ? 1u << dev : 0u;
    int rc = 0;
    if (!(gespmm::g_init_done.load() & bit)) {
        constexpr int M = 16384, deg = 8, N = 128;
        constexpr int64_t nnz = (int64_t)M * deg;
        char* block = nullptr;  // rowptr | colind | B | C
        const size_t b_rp = ((size_t)(M + 1) * 4 + 255) & ~(size_t)255, b_ci = (size_t)nnz * 4, b_d = (size_t)M * N * 4;
        e = hipMalloc(reinterpret_cast<void**>(&block), b_rp + b_ci + 2 * b_d);
        if (e != hipSuccess) return (int)e;
        int32_t* rp = reinterpret_cast<int32_t*>(block);
        int32_t* ci = reinterpret_cast<int32_t*>(block + b_rp);
        float* B = reinterpret_cast<float*>(block + b_rp + b_ci);
        float* C = B + (size_t)M * N;
        hipLaunchKernelGGL(gespmm::k_init_matrix, dim3((M + 256) / 256), dim3(256), 0, st, rp, ci, M, deg);
        e = hipMemsetAsync(B, 0, 2 * b_d, st);
        rc = e != hipSuccess ? (int)e : (int)hipGetLastError();
        // every kernel family once: the plain call, a clustered plan on the streaming kernels (both), the staged-rows kernels (tuned,
        // general, lane groups) — building the plans runs every analysis pass
        if (rc == 0) rc = gespmm::run_spmm(rp, ci, nullptr, B, C, M, M, N, nnz, GESPMM_VARIANT_AUTO, nullptr, gespmm::kReduceSum, 0.0f, stream, nullptr, 0, nullptr);
        const int kernels[5] = {GESPMM_PLAN_KERNEL_STREAM, GESPMM_PLAN_KERNEL_SEG_STREAM, GESPMM_PLAN_KERNEL_STAGED, GESPMM_PLAN_KERNEL_STAGED,
                                GESPMM_PLAN_KERNEL_RECORDS};  // (... and the padded-record kernel with its table passes)
        const int widths[5] = {N, N, N, 32, 32};
        for (int i = 0; i < 5 && rc == 0; ++i) {
            gespmm_plan_options opt;
            std::memset(&opt, 0, sizeof opt);
            opt.reorder = GESPMM_PLAN_REORDER;
            opt.kernel = kernels[i];
            gespmm_plan* p = nullptr;
            rc = gespmm_plan_create_v2(&p, rp, ci, nullptr, M, M, nnz, widths[i], GESPMM_VARIANT_AUTO, &opt, (int64_t)sizeof opt, stream);
            if (rc == 0) rc = gespmm_plan_spmm_f32(p, B, C, widths[i], stream);
            if (rc == 0 && i == 2) rc = gespmm_plan_spmm_max_f32(p, B, C, widths[i], -10000.0f, stream);  // (the general kernel)
            if (p) {
                (void)hipStreamSynchronize(st);
                gespmm_plan_destroy(p);
            }
        }
        const hipError_t es = hipStreamSynchronize(st);
        (void)hipFree(block);
        if (rc == 0 && es != hipSuccess) rc = (int)es;
        if (rc != 0) return rc;
        gespmm::g_init_done.fetch_or(bit);
        gespmm::mark_analysis_warm();
    }
    if (rows_hint > 0 && nnz_hint > 0) rc = gespmm::reserve_analysis_arena(rows_hint, rows_hint, nnz_hint, stream);
    return rc;
}

int gespmm_set_auto_plan(int32_t kth_call) {
    if (kth_call < 0) return GESPMM_EINVAL;
    gespmm::g_auto_k.store(kth_call, std::memory_order_relaxed);
    static const bool sync_only = getenv("GESPMM_AUTO_PLAN_SYNC") != nullptr;
    if (sync_only) gespmm::g_auto_async.store(false, std::memory_order_relaxed);
    if (kth_call == 0) gespmm_auto_plan_clear();
    return 0;
}

void gespmm_auto_plan_clear(void) {
    std::lock_guard<std::mutex> guard(gespmm::g_lock);
    for (gespmm::AutoEntry& e : gespmm::g_entries) {
        gespmm::release_entry(e);
        e = gespmm::AutoEntry();
    }
}

int gespmm_auto_plan_get_stats(gespmm_auto_plan_stats* out) {
    if (!out) return GESPMM_EINVAL;
    std::lock_guard<std::mutex> guard(gespmm::g_lock);
    *out = gespmm::g_stats;
    out->cached_plans = 0;
    for (const gespmm::AutoEntry& e : gespmm::g_entries) out->cached_plans += e.plan ? 1 : 0;
    return 0;
}

}  // extern "C"
