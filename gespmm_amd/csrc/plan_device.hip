// plan_device.hip — the plan's analysis stage on the device.
//
// Round 2 built plans on the host (plan.cpp copied the matrix over PCIe and ran reorder.cpp there): 0.17 s for a
// com-Amazon-sized graph, 4-6 s for a products-sized one, which the reference's own protocols (200 timed launches,
// spmm_test.cu:714; 200 epochs, gcn_custom.py:134) never amortise. The same analysis as device passes:
//
//   validation          one pass over rowptr / colind (monotone, in range, longest row)
//   row clustering      the multi-level label propagation of reorder.cpp, rule for rule (snapshot half-sweeps, size
//                       caps, hash tie-breaks, twins), so the order is IDENTICAL to cluster_rows() on the host —
//                       tests/test_gpu_plan_device.py compares the two permutations entry by entry:
//                         * a half-sweep = "heaviest label among my neighbours" per node, by degree class:
//                             1..8      8 lanes per node,  all-pairs match through lane shuffles
//                             9..64     64 lanes per node, all-pairs match through lane shuffles
//                             65..2048  one wavefront per node, open-addressing hash table in LDS
//                             > 2048    one workgroup per node, dense accumulator over the label space in HBM
//                           (integer weights: sums do not depend on the order of the additions);
//                         * contraction = relabel (scan), merged adjacency = radix sort of (cluster, column-cluster)
//                           pairs + run sums, transposed adjacency = one more stable sort (rocPRIM);
//                         * final order = stable sorts by the labels of each level, finest first.
//   L2 model            LRU stack distances: previous use of every column through one sort, then the exact number
//                       of distinct columns in between for a stratified sample of accesses (early exit at the
//                       window) — an unbiased estimate of simulate_l2_hits() to +-0.5 points
//   permuted copy       scan of the permuted degrees + one gather pass
//   task tables         the greedy cut of plan.cpp, parallel: next[] per row, per-block exit tables, one short
//                       serial walk over blocks, per-block marking, scan
//
// No reference counterpart (the reference has no analysis stage; see plan.cpp).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/block/block_scan.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/functional.hpp>

#include "auto_plan.h"
#include "plan_device.h"
#include "spmm_kernels.h"
#include "workspace.h"

namespace gespmm {

namespace {

#define GESPMM_TRY(expr)                       \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) return e__;     \
    } while (0)

// Temporaries of one analysis call: ONE device block carved into bump regions (a plan used to make ~250 allocator calls;
// each costs microseconds of host time and, the first time a size is seen, a real allocation). Regions: persistent for
// the call, two "level" regions that ping-pong between a level and the one built from it, and a temporary region that
// is rewound at phase boundaries. A request that does not fit its region gets a block of its own (correct, just
// slower), so the size estimates need not be tight.
//
// The block comes from plain hipMalloc through a one-entry cache (blocks up to 1/16 of the device memory, at most 16 GiB — GESPMM_ARENA_CACHE_MB — are kept for the next plan), NOT
// from the library's stream-ordered pool: with hipMallocFromPoolAsync / hipFreeAsync cycles of changing sizes interleaved
// with the plan's own hipMalloc calls, analysis results came out corrupted in processes without PyTorch's allocator in
// them (the spmm_test driver: wrong task tables, then a memory fault in the first launch; profiles/r03/pool_hazard.log).
// The analysis synchronises the stream anyway, so a synchronous free costs nothing here.
// A products-sized analysis (2.4 M rows, 124 M entries) takes ~10 GB for its clustering arena and ~2.5 GB for the model / staging
// passes. Blocks up to 1/16 of the device's memory (at most 16 GiB; GESPMM_ARENA_CACHE_MB) are kept for the next plan: a multi-GB
// hipMalloc was seen to take SECONDS now and then — after another library returned memory to the driver, or for no visible reason
// (profiles/r03/plan_repeat.log: the same 87 ms analysis took 1.9-3.6 s once in every few plans while its arena was a private block).
// Round 4: one cache entry PER DEVICE, a default cap of 1 GiB (a products-sized arena is ~10 GB: invisible to
// PyTorch's allocator, it must not stay behind by default — gespmm_set_cached_memory_limit / GESPMM_ARENA_CACHE_MB raise the
// cap for callers that build large plans in a row), and no hipMalloc / hipFree while the lock is held.
constexpr int kArenaDevices = 16;
std::atomic<long long> g_arena_cap{-1};
static size_t arena_cache_cap() {
    const long long set = g_arena_cap.load();
    if (set >= 0) return (size_t)set;
    static const size_t env_cap = []() -> size_t {
        if (getenv("GESPMM_ARENA_CACHE_MB")) return (size_t)atoll(getenv("GESPMM_ARENA_CACHE_MB")) << 20;
        return (size_t)1 << 30;
    }();
    return env_cap;
}
struct ArenaCache {
    std::mutex lock;
    void* block = nullptr;
    size_t bytes = 0;
    bool busy = false;
};
ArenaCache g_arena[kArenaDevices];

hipError_t arena_acquire(size_t bytes, void** out, bool* cached) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    *cached = false;
    if (dev < 0 || dev >= kArenaDevices) return hipMalloc(out, bytes);
    ArenaCache& c = g_arena[dev];
    void* stale = nullptr;
    bool reserve = false;
    {
        std::lock_guard<std::mutex> guard(c.lock);
        if (!c.busy) {
            if (c.block && c.bytes >= bytes) {
                c.busy = true;
                *out = c.block;
                *cached = true;
                return hipSuccess;
            }
            stale = c.block;  // too small: replaced (or dropped) below, outside the lock
            c.block = nullptr;
            c.bytes = 0;
            if (bytes <= arena_cache_cap()) c.busy = reserve = true;  // this thread fills the slot
        }
    }
    if (stale) (void)hipFree(stale);
    void* p = nullptr;
    e = hipMalloc(&p, bytes);
    if (reserve) {
        std::lock_guard<std::mutex> guard(c.lock);
        if (e == hipSuccess) {
            c.block = p;
            c.bytes = bytes;
            *cached = true;
        } else {
            c.busy = false;
        }
    }
    if (e == hipSuccess) *out = p;
    return e;  // !reserve: larger than the cache keeps, or another thread is analysing on this device: a private block
}

}  // namespace

// Drops the cached analysis arenas (gespmm_release_cached_memory): the next plan allocates a new one.
void release_cached_arena() {
    for (int d = 0; d < kArenaDevices; ++d) {
        void* stale = nullptr;
        {
            std::lock_guard<std::mutex> guard(g_arena[d].lock);
            if (!g_arena[d].busy && g_arena[d].block) {
                stale = g_arena[d].block;
                g_arena[d].block = nullptr;
                g_arena[d].bytes = 0;
            }
        }
        if (stale) (void)hipFree(stale);
    }
}

void set_arena_cache_limit(long long bytes) { g_arena_cap.store(bytes < 0 ? -1 : bytes); }

namespace {

void arena_release(void* p, bool cached) {
    if (!cached) {
        (void)hipFree(p);
        return;
    }
    for (int d = 0; d < kArenaDevices; ++d) {
        void* stale = nullptr;
        bool mine = false;
        {
            std::lock_guard<std::mutex> guard(g_arena[d].lock);
            if (g_arena[d].block == p) {
                mine = true;
                g_arena[d].busy = false;
                if (g_arena[d].bytes > arena_cache_cap()) {  // the cap was lowered meanwhile
                    stale = p;
                    g_arena[d].block = nullptr;
                    g_arena[d].bytes = 0;
                }
            }
        }
        if (stale) (void)hipFree(stale);
        if (mine) return;
    }
}

struct Scratch {
    enum { kPersist = 0, kLevelA = 1, kLevelB = 2, kTemp = 3, kRegions = 4 };
    hipStream_t st;
    char* base = nullptr;
    bool base_cached = false;
    size_t begin[kRegions] = {0, 0, 0, 0}, end[kRegions] = {0, 0, 0, 0}, top[kRegions] = {0, 0, 0, 0};
    int cur = kTemp;
    std::vector<void*> extras;
    explicit Scratch(hipStream_t s) : st(s) {}
    ~Scratch() {
        if (!base && extras.empty()) return;
        (void)hipStreamSynchronize(st);  // everything that used these bytes has finished
        for (void* b : extras) (void)hipFree(b);
        if (base) arena_release(base, base_cached);
    }
    static size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
    hipError_t init(size_t persist, size_t level, size_t temp) {
        const size_t sz[kRegions] = {up(persist), up(level), up(level), up(temp)};
        size_t total = 0;
        for (int r = 0; r < kRegions; ++r) {
            begin[r] = top[r] = total;
            total += sz[r];
            end[r] = total;
        }
        void* p = nullptr;
        hipError_t e = arena_acquire(total ? total : 256, &p, &base_cached);
        if (e != hipSuccess) return e;
        base = reinterpret_cast<char*>(p);
        return hipSuccess;
    }
    void use(int region) { cur = region; }
    size_t mark() const { return top[cur]; }
    void rewind(int region, size_t m) { top[region] = m; }
    void reset(int region) { top[region] = begin[region]; }
    template <typename T>
    hipError_t get(T** out, int64_t count) {
        const size_t bytes = up((size_t)(count > 0 ? count : 1) * sizeof(T));
        if (base && top[cur] + bytes <= end[cur]) {
            *out = reinterpret_cast<T*>(base + top[cur]);
            top[cur] += bytes;
            return hipSuccess;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
        extras.push_back(p);
        *out = reinterpret_cast<T*>(p);
        return hipSuccess;
    }
    void release(void*) {}  // regions are rewound as a whole
};

inline int bits_for(int64_t n) {  // bits that hold every value in [0, n)
    int b = 1;
    while (b < 62 && ((int64_t)1 << b) < n) ++b;
    return b;
}

inline unsigned grid_for(int64_t n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

__device__ inline uint32_t mix32(uint32_t x) {  // == reorder.cpp
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }

// row owning CSR position p: ptr[lo] <= p < ptr[lo + 1]
__device__ inline int owner_of(const int32_t* __restrict__ ptr, int n, int p) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------ validation

__global__ void k_validate(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t M, int64_t K,
                           int64_t nnz, int32_t* __restrict__ out /* {bad, max_degree} */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0, deg = 0;
    if (i < M) {
        const int a = rowptr[i], b = rowptr[i + 1];
        deg = b - a;
        if (deg < 0) bad |= 1;
        if (i == 0 && a != 0) bad |= 1;
        if (i == M - 1 && (int64_t)b != nnz) bad |= 1;
    }
    if (i < nnz && (uint32_t)colind[i] >= (uint64_t)K) bad |= 2;
    for (int o = 32; o > 0; o >>= 1) {
        bad |= __shfl_xor(bad, o);
        deg = max(deg, __shfl_xor(deg, o));
    }
    if (lane_id() == 0) {
        if (bad) atomicOr(&out[0], bad);
        // (the maximum only grows: a wavefront that sees it at or above its own skips the atomic — 29 000 same-address atomics on the
        //  headline graph were 65 us, ten times the reading of the matrix)
        if (deg > 0 && deg > __atomic_load_n(&out[1], __ATOMIC_RELAXED)) atomicMax(&out[1], deg);
    }
}

// The wedge probe (round 5, plan_policy.cpp: estimate_analysis_cost): for kProbeRows hash-chosen rows r of a SQUARE matrix and kProbePairs
// hash-chosen pairs (c1, c2) of r's columns each: is c2 a column of row c1? out[2] += hits, out[3] += wedges tried. Rides on the validation
// pass (same readback), so it may meet a matrix that pass is about to reject: every index is clamped before use. Row c1 is scanned
// linearly (columns may be unsorted), at most kProbeScan entries — longer rows count what the scan saw.
constexpr int kProbeRows = 4096, kProbePairs = 4, kProbeScan = 512;

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ void k_wedge_probe(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t M, int64_t nnz,
                              int32_t* __restrict__ out) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int hit = 0, tried = 0;
    if (t < kProbeRows * kProbePairs) {
        const int s = t / kProbePairs;
        const int64_t r = (int64_t)(probe_hash(0x9e3779b9u * (uint32_t)(s + 1)) % (uint64_t)M);
        int64_t a = rowptr[r], b = rowptr[r + 1];
        a = a < 0 ? 0 : (a > nnz ? nnz : a);
        b = b < a ? a : (b > nnz ? nnz : b);
        const int64_t d = b - a;
        if (d >= 2) {
            const uint32_t h = probe_hash((uint32_t)t * 0x85ebca6bu + 1u);
            const int64_t i = h % (uint64_t)d;
            int64_t j = (h >> 16 ^ h * 31u) % (uint64_t)(d - 1);
            if (j >= i) ++j;
            const int64_t c1 = colind[a + i], c2 = colind[a + j];
            if (c1 >= 0 && c1 < M) {
                tried = 1;
                int64_t a1 = rowptr[c1], b1 = rowptr[c1 + 1];
                a1 = a1 < 0 ? 0 : (a1 > nnz ? nnz : a1);
                b1 = b1 < a1 ? a1 : (b1 > nnz ? nnz : b1);
                if (b1 - a1 > kProbeScan) b1 = a1 + kProbeScan;
                for (int64_t q = a1; q < b1; ++q)
                    if (colind[q] == c2) {
                        hit = 1;
                        break;
                    }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        hit += __shfl_xor(hit, o);
        tried += __shfl_xor(tried, o);
    }
    if (lane_id() == 0 && tried) {
        atomicAdd(&out[2], hit);
        atomicAdd(&out[3], tried);
    }
}

// ------------------------------------------------------------------------------------------------ small helpers

__global__ void k_iota(int32_t* __restrict__ a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int32_t)i;
}

__global__ void k_fill(int32_t* __restrict__ a, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

// out[i] = first position u in sorted keys[0..n) with keys[u] >= i, for i in [0, nseg]
__global__ void k_lower_bound_ptr(const int32_t* __restrict__ keys, int n, int nseg, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nseg) return;
    int lo = 0, hi = n;  // first index with keys[idx] >= i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < i) lo = mid + 1;
        else hi = mid;
    }
    out[i] = lo;
}

// out[e] = the row that owns CSR position e, for every position. The 256 consecutive positions of a workgroup belong to a short range
// of rows: two full searches find it, every thread then searches inside (a few steps that hit L1) — against 19 dependent steps through
// L2 per entry when the owners of the SORTED positions were looked up after the transposition sort (86 us on the headline graph).
__global__ void __launch_bounds__(256) k_rows_of_entries(const int32_t* __restrict__ ptr, int n, int64_t E, int32_t* __restrict__ out) {
    __shared__ int s_lo, s_hi;
    const int64_t e0 = (int64_t)blockIdx.x * 256;
    if (threadIdx.x == 0) s_lo = owner_of(ptr, n, (int)e0);
    if (threadIdx.x == 64) s_hi = owner_of(ptr, n, (int)(e0 + 255 < E ? e0 + 255 : E - 1));
    __syncthreads();
    const int64_t e = e0 + threadIdx.x;
    if (e >= E) return;
    int lo = s_lo, hi = s_hi;  // the owner is the LAST row r in [lo, hi] with ptr[r] <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= (int)e) lo = mid;
        else hi = mid - 1;
    }
    out[e] = lo;
}

__global__ void k_gather2(const int32_t* __restrict__ src_a, const int32_t* __restrict__ src_b,
                          const int32_t* __restrict__ index, int cnt, int32_t* __restrict__ out_a,
                          int32_t* __restrict__ out_b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int j = index[i];
    out_a[i] = src_a[j];
    out_b[i] = src_b[j];
}

// ------------------------------------------------------------------------------------------------ label propagation

// One side of a bipartite level on the device: node -> (neighbour on the other side, weight).
struct DAdj {
    const int32_t* ptr = nullptr;  // [n + 1]
    const int32_t* idx = nullptr;
    const int32_t* w = nullptr;    // nullptr: all ones
};

constexpr int kDefaultClusterLevels = 6;  // == reorder.cpp. Levels 4-6 merge little (8872 > 6372 > 5957 > 5932 clusters on the structureless com-Amazon
                                          // stand-in) yet its plan runs 138.6 instead of 145.1 us at N = 128 (47.4 / 50.0 at 32, 537 / 545 at 512) with them:
                                          // profiles/r04/like_regression.log; round 3 had cut them for 2 ms of analysis time
constexpr int kBins = 6;  // degree classes 1..8, 9..16, 17..32, 33..64 (one lane per entry in a group of that many lanes), 65..2048, > 2048
                          // (round 5: 9..64 was ONE class of 64-lane groups — 0.6 ms of the headline graph's 3.6 ms analysis went into wavefronts
                          //  that were three quarters empty; a node's result does not depend on the width of its group)
constexpr int kHashSlots = 4096;
__device__ inline int bin_of(int d) { return d <= 8 ? 0 : (d <= 16 ? 1 : (d <= 32 ? 2 : (d <= 64 ? 3 : (d <= 2048 ? 4 : 5)))); }

// lists[b * n + ...] = nodes of class b (order irrelevant: every node's result depends only on the snapshot). A workgroup counts its
// nodes per class in LDS and claims its share of each list with ONE global atomic per class (one per wavefront and class before:
// 5 000 same-address atomics on the headline graph, 53 us per call — profiles/r05/plan_kernel_stats_before.csv).
constexpr int kBinThreads = 1024;
__global__ void __launch_bounds__(kBinThreads) k_bin_nodes(const int32_t* __restrict__ ptr, const int32_t* __restrict__ twin, int n,
                                                           int32_t* __restrict__ lists, int32_t* __restrict__ counts) {
    __shared__ int s_cnt[kBins], s_base[kBins];
    if (threadIdx.x < kBins) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b = -1;
    if (i < n) {
        const int d = ptr[i + 1] - ptr[i];
        if (d > 0 && !(twin && twin[i] >= 0)) b = bin_of(d);
    }
    const int lane = lane_id();
    int off = 0;  // this node's place inside the workgroup's share of list b
    for (int k = 0; k < kBins; ++k) {
        const unsigned long long mask = __ballot(b == k);
        if (mask == 0) continue;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&s_cnt[k], __popcll(mask));
        base = __shfl(base, leader);
        if (b == k) off = base + __popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < kBins) s_base[threadIdx.x] = s_cnt[threadIdx.x] > 0 ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0;
    __syncthreads();
    if (b >= 0) lists[(int64_t)b * n + s_base[b] + off] = i;
}

struct LpArgs {
    DAdj adj;                    // adjacency of the side that moves
    const int32_t* nbr_label;    // labels of the other side (snapshot)
    const int32_t* cur;          // labels of this side (snapshot)
    int32_t* next;
    const int32_t* size;         // rows side: original rows owned by each label (snapshot); nullptr on the column side
    const int32_t* rweight;      // rows side: original rows owned by each node
    long long cap;
    uint32_t seed;
    int skip_half;               // rows side, twins, not the last sweep: half of the nodes sit this sweep out
    const int32_t* list;
    const int32_t* count;
    const int32_t* done;
};

__device__ inline bool sits_out(const LpArgs& a, int node) {
    return a.skip_half && (mix32((uint32_t)node * 0x85ebca6bu ^ a.seed) & 1u);
}

// (weight, hash, label) butterfly maximum inside a group of G lanes: larger weight wins, then the smaller hash
template <int G>
__device__ inline void best_of_group(long long& bt, uint32_t& bh, int& bL) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        const long long ot = __shfl_xor(bt, o, G);
        const uint32_t oh = (uint32_t)__shfl_xor((int)bh, o, G);
        const int oL = __shfl_xor(bL, o, G);
        if (ot > bt || (ot == bt && ot >= 0 && oh < bh)) {
            bt = ot;
            bh = oh;
            bL = oL;
        }
    }
}

// degree <= G: one entry per lane, every lane sums the weights of the lanes that carry its label. `block` = the workgroup's index
// inside its degree class (k_lp_small: the four classes of 1..64 entries share ONE launch).
template <int G, bool ROWS>
__device__ __forceinline__ void lp_small_body(const LpArgs& a, const int32_t* __restrict__ list, int count, long long block) {
    const long long tid = block * 256 + threadIdx.x;  // (count * G may exceed 2^31 on forced huge plans)
    const int g = (int)(tid / G), l = (int)(tid % G);
    const bool active = g < count;
    const int node = active ? list[g] : 0;
    const int beg = active ? a.adj.ptr[node] : 0, end = active ? a.adj.ptr[node + 1] : 0;
    const int e = beg + l;
    int L = -1, wt = 0;
    if (e < end) {
        L = a.nbr_label[a.adj.idx[e]];
        wt = (L >= 0) ? (a.adj.w ? a.adj.w[e] : 1) : 0;
    }
    long long tot = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int Lj = __shfl(L, j, G);
        const int wj = __shfl(wt, j, G);
        tot += (Lj == L) ? wj : 0;
    }
    const int own = active ? a.cur[node] : -1;
    long long own_w = (L >= 0 && L == own) ? tot : 0;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) own_w = max(own_w, __shfl_xor(own_w, o, G));
    bool cand = L >= 0 && L != own && tot > own_w;  // the own label keeps every tie it is part of
    if (ROWS && cand) cand = (long long)a.size[L] + a.rweight[node] <= a.cap;
    long long bt = cand ? tot : -1;
    uint32_t bh = mix32((uint32_t)L ^ a.seed);
    int bL = L;
    best_of_group<G>(bt, bh, bL);
    if (active && l == 0) a.next[node] = (ROWS && sits_out(a, node)) ? own : (bt >= 0 ? bL : own);
}

// The classes of 1..8, 9..16, 17..32 and 33..64 entries in one launch: workgroups [blk_end[k - 1], blk_end[k]) serve class k.
// (Round 6: four launches of 5-11 us each per half-sweep, 30 half-sweeps per plan — most of each was its own ramp:
//  profiles/r06/plan_kernels_before.log. A node's result does not depend on which workgroup computes it.)
struct LpSmallClasses {
    const int32_t* lists;   // [kBins][n_side]
    const int32_t* counts;  // [kBins] (device)
    int n_side;
    int blk_end[4];
};
template <bool ROWS>
__global__ void __launch_bounds__(256) k_lp_small(LpArgs a, LpSmallClasses c) {
    if (*a.done) return;
    const int b = (int)blockIdx.x;
    if (b < c.blk_end[0]) lp_small_body<8, ROWS>(a, c.lists, c.counts[0], b);
    else if (b < c.blk_end[1]) lp_small_body<16, ROWS>(a, c.lists + (int64_t)c.n_side, c.counts[1], b - c.blk_end[0]);
    else if (b < c.blk_end[2]) lp_small_body<32, ROWS>(a, c.lists + 2 * (int64_t)c.n_side, c.counts[2], b - c.blk_end[1]);
    else lp_small_body<64, ROWS>(a, c.lists + 3 * (int64_t)c.n_side, c.counts[3], b - c.blk_end[2]);
}

// 64 < degree <= 2048: one wavefront per node, label -> weight in an LDS hash table
template <bool ROWS>
__global__ void __launch_bounds__(64) k_lp_wave(LpArgs a) {
    __shared__ int s_key[kHashSlots];
    __shared__ unsigned long long s_wt[kHashSlots];
    if (*a.done) return;
    const int g = blockIdx.x;
    if (g >= *a.count) return;
    const int node = a.list[g];
    const int lane = threadIdx.x;
    const int own = a.cur[node];
    if (ROWS && sits_out(a, node)) {
        if (lane == 0) a.next[node] = own;
        return;
    }
    const int beg = a.adj.ptr[node], end = a.adj.ptr[node + 1];
    int T = 128;
    while (T < 2 * (end - beg)) T <<= 1;
    for (int s = lane; s < T; s += 64) {
        s_key[s] = -1;
        s_wt[s] = 0ull;
    }
    __syncthreads();
    for (int e = beg + lane; e < end; e += 64) {
        const int L = a.nbr_label[a.adj.idx[e]];
        if (L < 0) continue;
        const unsigned long long wt = (unsigned long long)(a.adj.w ? a.adj.w[e] : 1);
        int slot = (int)(mix32((uint32_t)L) & (uint32_t)(T - 1));
        for (;;) {
            const int prev = atomicCAS(&s_key[slot], -1, L);
            if (prev == -1 || prev == L) {
                atomicAdd(&s_wt[slot], wt);
                break;
            }
            slot = (slot + 1) & (T - 1);
        }
    }
    __syncthreads();
    long long own_w = 0;
    if (own >= 0) {
        int slot = (int)(mix32((uint32_t)own) & (uint32_t)(T - 1));
        for (;;) {
            const int k = s_key[slot];
            if (k == own) {
                own_w = (long long)s_wt[slot];
                break;
            }
            if (k == -1) break;
            slot = (slot + 1) & (T - 1);
        }
    }
    long long bt = -1;
    uint32_t bh = 0;
    int bL = -1;
    const int rw = ROWS ? a.rweight[node] : 0;
    for (int s = lane; s < T; s += 64) {
        const int L = s_key[s];
        if (L < 0 || L == own) continue;
        const long long t = (long long)s_wt[s];
        if (t <= own_w) continue;
        if (ROWS && (long long)a.size[L] + rw > a.cap) continue;
        const uint32_t h = mix32((uint32_t)L ^ a.seed);
        if (t > bt || (t == bt && h < bh)) {
            bt = t;
            bh = h;
            bL = L;
        }
    }
    best_of_group<64>(bt, bh, bL);
    if (lane == 0) a.next[node] = bt >= 0 ? bL : own;
}

// degree > 2048: a workgroup per node, weights accumulated in a dense array over the label space (one array per
// workgroup, zero on entry and zero again on exit); all accesses through L2 atomics / device-scope loads
template <bool ROWS>
__global__ void __launch_bounds__(256) k_lp_dense(LpArgs a, unsigned long long* __restrict__ acc_all, int64_t nlabels) {
    __shared__ long long s_t[256];
    __shared__ uint32_t s_h[256];
    __shared__ int s_L[256];
    if (*a.done) return;
    unsigned long long* acc = acc_all + (int64_t)blockIdx.x * nlabels;
    const int cnt = *a.count;
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < cnt; g += gridDim.x) {
        const int node = a.list[g];
        const int own = a.cur[node];
        if (ROWS && sits_out(a, node)) {
            if (tid == 0) a.next[node] = own;
            continue;
        }
        const int beg = a.adj.ptr[node], end = a.adj.ptr[node + 1];
        for (int e = beg + tid; e < end; e += 256) {
            const int L = a.nbr_label[a.adj.idx[e]];
            if (L >= 0) atomicAdd(&acc[L], (unsigned long long)(a.adj.w ? a.adj.w[e] : 1));
        }
        __threadfence();
        __syncthreads();
        const long long own_w =
            own >= 0 ? (long long)__hip_atomic_load(&acc[own], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        long long bt = -1;
        uint32_t bh = 0;
        int bL = -1;
        const int rw = ROWS ? a.rweight[node] : 0;
        for (int e = beg + tid; e < end; e += 256) {
            const int L = a.nbr_label[a.adj.idx[e]];
            if (L < 0 || L == own) continue;
            const long long t = (long long)__hip_atomic_load(&acc[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t <= own_w) continue;
            if (ROWS && (long long)a.size[L] + rw > a.cap) continue;
            const uint32_t h = mix32((uint32_t)L ^ a.seed);
            if (t > bt || (t == bt && h < bh)) {
                bt = t;
                bh = h;
                bL = L;
            }
        }
        s_t[tid] = bt;
        s_h[tid] = bh;
        s_L[tid] = bL;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const long long ot = s_t[tid + o];
                const uint32_t oh = s_h[tid + o];
                if (ot > s_t[tid] || (ot == s_t[tid] && ot >= 0 && oh < s_h[tid])) {
                    s_t[tid] = ot;
                    s_h[tid] = oh;
                    s_L[tid] = s_L[tid + o];
                }
            }
            __syncthreads();
        }
        if (tid == 0) a.next[node] = s_t[0] >= 0 ? s_L[0] : own;
        for (int e = beg + tid; e < end; e += 256) {
            const int L = a.nbr_label[a.adj.idx[e]];
            if (L >= 0) __hip_atomic_store(&acc[L], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        __syncthreads();
    }
}

// Column side: labels move from `next` to `cur`; a column with a twin carries its cluster's (new) row label instead.
__global__ void k_commit_cols(int32_t* __restrict__ cur, const int32_t* __restrict__ next, const int32_t* __restrict__ twin,
                              const int32_t* __restrict__ rlab, int n, const int32_t* __restrict__ done) {
    if (*done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cur[i] = (twin && twin[i] >= 0) ? rlab[twin[i]] : next[i];
}

// Row side: labels move, the per-label sizes follow the rows that moved (so `size` is always the snapshot the next
// half-sweep needs), and the last workgroup to finish decides whether the level's remaining sweeps are skipped:
// fewer than 0.25 % of the row nodes moved (reorder.cpp).
__global__ void k_commit_rows(int32_t* __restrict__ cur, const int32_t* __restrict__ next, const int32_t* __restrict__ rweight,
                              int32_t* __restrict__ size, int n, int32_t* __restrict__ changed, int32_t* __restrict__ ticket,
                              int32_t* __restrict__ done) {
    if (*done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int ch = 0;
    if (i < n) {
        const int v = next[i], old = cur[i];
        if (v != old) {
            ch = 1;
            cur[i] = v;
            const int w = rweight[i];
            atomicAdd(&size[v], w);
            atomicSub(&size[old], w);
        }
    }
    // (one global atomic per workgroup, of 1024 threads: a per-wavefront atomic on `changed` made this kernel 36 us per call)
    __shared__ int s_changed;
    if (threadIdx.x == 0) s_changed = 0;
    __syncthreads();
    const unsigned long long m = __ballot(ch);
    if (m && lane_id() == (__ffsll((long long)m) - 1)) atomicAdd(&s_changed, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_changed) atomicAdd(changed, s_changed);
        __threadfence();
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
            const int c = atomicAdd(changed, 0);
            if ((long long)c * 400 < n) atomicExch(done, 1);
            atomicExch(changed, 0);
            atomicExch(ticket, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ contraction

__global__ void k_mark_used(const int32_t* __restrict__ rlab, int R, const int32_t* __restrict__ clab, int C,
                            int32_t* __restrict__ used_r, int32_t* __restrict__ used_c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) used_r[rlab[i]] = 1;
    if (i < C) {
        const int L = clab[i];
        if (L >= 0) used_c[L] = 1;
    }
}

// used / pos hold two halves of R + 1 entries (row labels, column labels) scanned as ONE array: positions of the
// second half are offset by the number of row labels in use (*row_total = pos[R])
__global__ void k_make_ids(const int32_t* __restrict__ used, const int32_t* __restrict__ pos,
                           const int32_t* __restrict__ row_total, int R, int32_t* __restrict__ rid, int32_t* __restrict__ cid) {
    const int L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= R) return;
    rid[L] = used[L] ? pos[L] : -1;
    const int j = R + 1 + L;
    cid[L] = used[j] ? pos[j] - *row_total : -1;
}

// start of a level: every row node is its own label (and owns itself), columns are unlabelled, flags cleared
__global__ void k_level_init(int32_t* __restrict__ rlab, int32_t* __restrict__ rnext, int32_t* __restrict__ size,
                             const int32_t* __restrict__ rweight, int R, int32_t* __restrict__ clab,
                             int32_t* __restrict__ cnext, int C, int32_t* __restrict__ flags /* [16] */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) {
        rlab[i] = i;
        rnext[i] = i;
        size[i] = rweight[i];
    }
    if (i < C) {
        clab[i] = -1;
        cnext[i] = -1;
    }
    if (i < 16) flags[i] = 0;
}

__global__ void k_fill2(int32_t* __restrict__ a, int64_t na, int32_t va, int32_t* __restrict__ b, int64_t nb, int32_t vb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = va;
    if (i < nb) b[i] = vb;
}

// parent[r] = node of the next level that row node r becomes part of
__global__ void k_parents(const int32_t* __restrict__ rlab, const int32_t* __restrict__ rid, int R,
                          int32_t* __restrict__ parent) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) parent[r] = rid[rlab[r]];
}

__global__ void k_new_weights(const int32_t* __restrict__ rlab, const int32_t* __restrict__ rid,
                              const int32_t* __restrict__ rweight, int R, int32_t* __restrict__ rweight2) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) atomicAdd(&rweight2[rid[rlab[r]]], rweight[r]);
}

__global__ void k_new_twin(const int32_t* __restrict__ rid, const int32_t* __restrict__ cid, int R,
                           int32_t* __restrict__ twin2) {
    const int L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L < R && cid[L] >= 0) twin2[cid[L]] = rid[L];
}

// one (new row node, new column node) key per edge of the level; edges that vanish (unlabelled column, a cluster's
// edge to its own twin) get the sentinel row R2 and sort behind everything
__global__ void k_emit_edges(DAdj rows, int R, int E, const int32_t* __restrict__ rlab, const int32_t* __restrict__ clab,
                             const int32_t* __restrict__ rid, const int32_t* __restrict__ cid,
                             const int32_t* __restrict__ twin2, int R2, int shift, unsigned long long* __restrict__ keys,
                             int32_t* __restrict__ vals) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int r = owner_of(rows.ptr, R, e);
    const int cl = clab[rows.idx[e]];
    unsigned long long key = (unsigned long long)R2 << shift;
    if (cl >= 0) {
        const int n = rid[rlab[r]], cn = cid[cl];
        if (twin2[cn] != n) key = ((unsigned long long)n << shift) | (unsigned long long)cn;
    }
    keys[e] = key;
    vals[e] = rows.w ? rows.w[e] : 1;
}

__global__ void k_heads(const unsigned long long* __restrict__ keys, int E, int32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    if (i == E) flags[i] = 0;
}

// every run of equal keys becomes one weighted edge: a wavefront sums the pieces of runs it sees with a segmented
// shuffle scan, the last lane of every piece adds it to the run's 64-bit total (integer adds: order-free)
__global__ void __launch_bounds__(256) k_run_sums(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ vals,
                                                  const int32_t* __restrict__ flags, const int32_t* __restrict__ uid, int E,
                                                  int R2, int shift, int32_t* __restrict__ n_of, int32_t* __restrict__ idx2,
                                                  unsigned long long* __restrict__ total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = lane_id();
    const bool in = i < E;
    const unsigned long long key = in ? keys[i] : ~0ull;
    long long sum = in ? (long long)vals[i] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long ko = __shfl_up(key, o);
        const long long so = __shfl_up(sum, o);
        if (lane >= o && ko == key) sum += so;
    }
    const unsigned long long kn = __shfl_down(key, 1);
    const bool piece_end = in && (lane == 63 || kn != key || i + 1 >= E);
    if (!in) return;
    const int n = (int)(key >> shift);
    if (n >= R2) return;  // the sentinel run
    const int head = flags[i];
    const int u = uid[i] + head - 1;  // uid = heads strictly before i
    if (head) {
        n_of[u] = n;
        idx2[u] = (int)(key & ((1ull << shift) - 1ull));
    }
    if (piece_end) atomicAdd(&total[u], (unsigned long long)sum);
}

__global__ void k_clamp_weights(const unsigned long long* __restrict__ total, int n, int32_t* __restrict__ w2) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) {
        const unsigned long long t = total[u];
        w2[u] = (int32_t)(t < 0x7fffffffull ? t : 0x7fffffffull);  // as on the host
    }
}

// ------------------------------------------------------------------------------------------------ order, copy

// keys[c] = rank of c's parent (rank == nullptr: the parent's id), the sort key that orders level-l clusters
__global__ void k_parent_rank(const int32_t* __restrict__ parent, const int32_t* __restrict__ rank, int64_t n,
                              int32_t* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = rank ? rank[parent[i]] : parent[i];
    ids[i] = (int32_t)i;
}

// The same order from ONE sort of the rows (round 5): key = the row's ancestors (cluster of level hi, ..., cluster of level lo + 1), the
// coarser the more significant; parent[l][node of level l] = its cluster of level l + 1.
constexpr int kOrderMaxLevels = 16;
struct OrderLevels {
    const int32_t* parent[kOrderMaxLevels];
    int bits[kOrderMaxLevels];
};
__global__ void k_order_keys(OrderLevels lv, const int32_t* __restrict__ rows /* nullptr: row i */, int64_t M, int lo, int hi,
                             unsigned long long* __restrict__ keys, int32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int row = rows ? rows[i] : (int)i;
    int node = row, shift = 0;
    unsigned long long key = 0;
    for (int l = 0; l < hi; ++l) {
        node = lv.parent[l][node];
        if (l >= lo) {
            key |= (unsigned long long)(uint32_t)node << shift;
            shift += lv.bits[l];
        }
    }
    keys[i] = key;
    vals[i] = row;
}

__global__ void k_invert(const int32_t* __restrict__ sorted_ids, int64_t n, int32_t* __restrict__ rank) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rank[sorted_ids[i]] = (int32_t)i;
}

__global__ void k_perm_degrees(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, int64_t M,
                               int32_t* __restrict__ deg /* [M + 1] */, int32_t* __restrict__ src_begin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) {
        const int r = perm[i];
        const int b = rowptr[r];
        deg[i] = rowptr[r + 1] - b;
        src_begin[i] = b;
    } else if (i == M) {
        deg[i] = 0;
    }
}

__global__ void k_copy_entries(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ src_begin,
                               const int32_t* __restrict__ colind, int M, int nnz, int32_t* __restrict__ colind_p) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int lo = owner_of(rowptr_p, M, p);
    colind_p[p] = colind[src_begin[lo] + (p - rowptr_p[lo])];
}

// ------------------------------------------------------------------------------------------------ L2 model

struct SliceInfo {
    int32_t begin[17];  // first modelled CSR position of slice s
    int32_t end[17];    // one past the last modelled position
};

__global__ void k_slice_bounds(const int32_t* __restrict__ rowptr, int M, int nnz, int slices, long long cap,
                               SliceInfo* __restrict__ out) {
    const int s = threadIdx.x;
    if (s >= slices) return;
    // cut s = smallest row count j (>= 1 for s >= 1) with rowptr[j] >= nnz * s / slices (simulate_l2_hits)
    auto cut = [&](int k) -> int {
        if (k <= 0) return 0;
        if (k >= slices) return M;
        const long long target = (long long)nnz * k / slices;
        int lo = 1, hi = M;  // first j in [1, M] with rowptr[j] >= target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rowptr[mid] < target) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    const int r0 = cut(s), r1 = cut(s + 1);
    const int b = rowptr[r0];
    int e = rowptr[r1];
    if (cap > 0 && (long long)e - b > cap) {
        // whole rows until `cap` entries have been seen: first row boundary at or beyond b + cap
        int lo = r0, hi = r1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)rowptr[mid] - b < cap) lo = mid + 1;
            else hi = mid;
        }
        e = rowptr[lo];
    }
    out->begin[s] = b;
    out->end[s] = e;
}

__global__ void k_model_keys(const int32_t* __restrict__ colind, const SliceInfo* __restrict__ info, int slices,
                             int64_t total, int32_t* __restrict__ cols, int32_t* __restrict__ poss) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // i-th modelled access overall -> slice, position
    int64_t off = i;
    int s = 0;
    while (s < slices - 1 && off >= info->end[s] - info->begin[s]) {
        off -= info->end[s] - info->begin[s];
        ++s;
    }
    const int p = info->begin[s] + (int)off;
    cols[i] = colind[p];
    poss[i] = p;
}

// after a STABLE sort by column the positions of one column are ascending: prev[p] = position of the previous access to
// the same column (any slice; the consumer checks the slice), -1 if none
__global__ void k_model_prev(const int32_t* __restrict__ cols_sorted, const int32_t* __restrict__ pos_sorted, int64_t total,
                             int32_t* __restrict__ prev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    prev[pos_sorted[i]] = (i > 0 && cols_sorted[i - 1] == cols_sorted[i]) ? pos_sorted[i - 1] : -1;
}

// One workgroup (four wavefronts) per sampled access p: hit iff the column was used before in this slice at position a and fewer than
// `window` DISTINCT columns were used in (a, p) — an access q in between is a first use there iff prev[q] < a. The wavefronts take
// alternate 512-access pieces of (a, p) and add what they count to one LDS counter; each leaves as soon as the counter has reached
// the window (the answer is then "miss" whatever the others still add), so the count is exact whenever it matters.
// (Round 5: one wavefront per sample walked up to a whole slice in 256-access steps — 425 us per call on the headline graph, a sixth
//  of the analysis: profiles/r05/plan_kernel_stats_before.csv.)
constexpr int kModelSampleThreads = 256;
__global__ void __launch_bounds__(kModelSampleThreads) k_model_sample(const int32_t* __restrict__ prev, const SliceInfo* __restrict__ info,
                                                                      int slices, int samples, long long window,
                                                                      int32_t* __restrict__ hits /* [slices] */,
                                                                      int32_t* __restrict__ taken /* [slices] */) {
    __shared__ int s_distinct;
    const int s = blockIdx.y;
    const int k = blockIdx.x;
    const int b = info->begin[s], e = info->end[s];
    const long long len = (long long)e - b;
    if (len <= 0) return;
    const long long n = len < samples ? len : samples;
    if (k >= n) return;
    const int p = b + (int)(((2 * (long long)k + 1) * len) / (2 * n));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kModelSampleThreads / 64, kPiece = 512;
    const int a = prev[p];
    int hit = 0;
    if (a >= b) {  // (uniform over the workgroup: every thread read the same prev[p])
        if ((long long)p - a - 1 < window) hit = 1;  // fewer accesses than the window holds
        else {
            if (threadIdx.x == 0) s_distinct = 0;
            __syncthreads();
            for (long long q0 = (long long)a + 1 + (long long)wave * kPiece; q0 < p; q0 += (long long)kWaves * kPiece) {
                int first = 0;
#pragma unroll
                for (int j = 0; j < kPiece / 64; ++j) {  // eight independent coalesced loads per lane in flight
                    const long long q = q0 + j * 64 + lane;
                    first += (q < p && prev[q] < a) ? 1 : 0;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) first += __shfl_xor(first, o);
                int seen = 0;
                if (lane == 0) seen = atomicAdd(&s_distinct, first) + first;
                seen = __shfl(seen, 0);
                if (seen >= window) break;
            }
            __syncthreads();
            hit = s_distinct < window ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        atomicAdd(&taken[s], 1);
        if (hit) atomicAdd(&hits[s], 1);
    }
}

// ------------------------------------------------------------------------------------------------ task cutting
// Both task tables of a plan (wavefront tasks, lane-group tasks) are cut together: variant v in {0, 1} has its own
// budget and row floor; arrays carry a leading variant dimension.

constexpr int kCutBlock = 1024;  // rows per block of the parallel greedy cut

struct CutParams {
    long long budget[2];
    long long row_floor[2];
};

__global__ void k_row_costs(const int32_t* __restrict__ rp, int64_t M, CutParams cp, long long* __restrict__ cost /* [2][M + 1] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    const long long d = i < M ? rp[i + 1] - rp[i] : 0;
#pragma unroll
    for (int v = 0; v < 2; ++v) cost[v * (M + 1) + i] = i < M ? (d > cp.row_floor[v] ? d : cp.row_floor[v]) : 0;
}

// next[i] = row after the last row of the task that starts at row i
__global__ void k_task_next(const long long* __restrict__ P_all /* exclusive prefixes of the costs, [2][M + 1] */, int64_t M,
                            CutParams cp, int32_t* __restrict__ next_all /* [2][M] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (i >= M) return;
    const long long* P = P_all + v * (M + 1);
    int64_t hi = i + kMaxRowsPerWave < M ? i + kMaxRowsPerWave : M;
    int64_t lo = i + 1;  // largest j in [i + 1, hi] with P[j] - P[i] <= budget (P is non-decreasing)
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (P[mid] - P[i] <= cp.budget[v]) lo = mid;
        else hi = mid - 1;
    }
    next_all[v * M + i] = (int32_t)lo;
}

// A chain enters a block within its first kMaxRowsPerWave rows (a task never has more rows): exit row of the chain
// for each of those entry points. The block's next[] is staged in LDS: the walks are serial chains of lookups.
__global__ void __launch_bounds__(256) k_task_exits(const int32_t* __restrict__ next_all, int64_t M, int64_t nblk,
                                                    int32_t* __restrict__ exits_all /* [2][nblk][kMaxRowsPerWave] */) {
    __shared__ int32_t s_next[kCutBlock];
    const int v = blockIdx.y;
    const int32_t* next = next_all + v * M;
    const int64_t b0 = (int64_t)blockIdx.x * kCutBlock;
    const int64_t b1 = b0 + kCutBlock < M ? b0 + kCutBlock : M;
    for (int i = threadIdx.x; i < b1 - b0; i += blockDim.x) s_next[i] = next[b0 + i];
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= kMaxRowsPerWave) return;
    int64_t cur = b0 + t;
    while (cur < b1) cur = s_next[cur - b0];
    exits_all[((int64_t)v * nblk + blockIdx.x) * kMaxRowsPerWave + t] = (int32_t)cur;
}

// first task start inside every block: one serial walk over the blocks per variant, out of LDS when the table fits
__global__ void __launch_bounds__(256) k_task_entries(const int32_t* __restrict__ exits_all, int64_t nblk,
                                                      int32_t* __restrict__ entry_all /* [2][nblk] */) {
    constexpr int kLdsInts = 12288;  // 48 KB
    __shared__ int32_t s_exits[kLdsInts];
    const int v = blockIdx.x;
    const int32_t* exits = exits_all + (int64_t)v * nblk * kMaxRowsPerWave;
    const bool in_lds = nblk * kMaxRowsPerWave <= kLdsInts;
    if (in_lds)
        for (int64_t i = threadIdx.x; i < nblk * kMaxRowsPerWave; i += blockDim.x) s_exits[i] = exits[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    int64_t cur = 0;
    for (int64_t b = 0; b < nblk; ++b) {
        entry_all[v * nblk + b] = (int32_t)cur;  // always within the block's first kMaxRowsPerWave rows
        const int64_t k = b * kMaxRowsPerWave + (cur - b * kCutBlock);
        cur = in_lds ? s_exits[k] : exits[k];
    }
}

__global__ void __launch_bounds__(256) k_task_mark(const int32_t* __restrict__ next_all, const int32_t* __restrict__ entry_all,
                                                   int64_t M, int64_t nblk, int32_t* __restrict__ flags_all /* zeroed, [2][M + 1] */) {
    __shared__ int32_t s_next[kCutBlock];
    const int v = blockIdx.y;
    const int32_t* next = next_all + v * M;
    int32_t* flags = flags_all + v * (M + 1);
    const int64_t b0 = (int64_t)blockIdx.x * kCutBlock;
    const int64_t b1 = b0 + kCutBlock < M ? b0 + kCutBlock : M;
    for (int i = threadIdx.x; i < b1 - b0; i += blockDim.x) s_next[i] = next[b0 + i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    int64_t cur = entry_all[v * nblk + blockIdx.x];
    while (cur < b1) {
        flags[cur] = 1;
        cur = s_next[cur - b0];
    }
}

__global__ void k_task_write(const int32_t* __restrict__ flags, const int32_t* __restrict__ tid_of,
                             const int32_t* __restrict__ next, const int32_t* __restrict__ rp, int64_t M,
                             int32_t* __restrict__ tasks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || !flags[i]) return;
    const int j = next[i];
    int4 t;
    t.x = (int)i;
    t.y = j - (int)i;
    t.z = rp[i];
    t.w = rp[j];
    reinterpret_cast<int4*>(tasks)[tid_of[i]] = t;
}

// ------------------------------------------------------------------------------------------------ host-side glue

template <typename T>
hipError_t exclusive_scan(Scratch& sc, const T* in, T* out, int64_t n, hipStream_t st) {
    size_t bytes = 0;
    GESPMM_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), st));
    char* tmp = nullptr;
    const size_t m = sc.mark();
    GESPMM_TRY(sc.get(&tmp, (int64_t)bytes));
    GESPMM_TRY(rocprim::exclusive_scan(tmp, bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), st));
    sc.rewind(sc.cur, m);  // later users of these bytes are later on the stream
    return hipSuccess;
}

// Keys here have 10-35 significant bits: Onesweep (a histogram + one kernel per 8 bits) instead of rocPRIM's default
// merge sort below 2^20 items (block sort + two kernels per doubling: ~20 launches for the 3*10^5-row rank sorts).
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

template <typename KeyT>
hipError_t sort_pairs(Scratch& sc, const KeyT* kin, KeyT* kout, const int32_t* vin, int32_t* vout, int64_t n, int bits,
                      hipStream_t st) {
    size_t bytes = 0;
    GESPMM_TRY(rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, st));
    char* tmp = nullptr;
    const size_t m = sc.mark();
    GESPMM_TRY(sc.get(&tmp, (int64_t)bytes));
    GESPMM_TRY(rocprim::radix_sort_pairs<SortConfig>(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)bits, st));
    sc.rewind(sc.cur, m);
    return hipSuccess;
}

template <typename T>
hipError_t fetch(T* host, const T* dev, int64_t count, hipStream_t st) {
    GESPMM_TRY(hipMemcpyAsync(host, dev, (size_t)count * sizeof(T), hipMemcpyDeviceToHost, st));
    return hipStreamSynchronize(st);
}

struct DLevel {
    int32_t R = 0, C = 0;
    int64_t E = 0;
    DAdj rows, cols;
    const int32_t* rweight = nullptr;
    const int32_t* twin = nullptr;  // nullptr at level 0
    int region = Scratch::kLevelA;  // where this level's arrays live (rewound when the next level replaces it)
};

// transposed adjacency: stable sort of the entries by column node keeps the row nodes ascending inside a column
// (round 6: counts + one atomic per entry instead of the sort — the order inside a column is free — measured 0.15 ms less per plan
//  on the headline graph, 39 + 52 us of atomics per level against three radix passes; not kept: a hub column serialises its atomics,
//  the sort does not care. profiles/r06/plan_kernels_fused.log)
hipError_t build_cols(Scratch& sc, DLevel& lv, const int32_t* n_of /* nullptr: rows.ptr decides (level 0) */,
                      hipStream_t st) {
    const int64_t E = lv.E;
    int32_t *iota = nullptr, *keys_out = nullptr, *order = nullptr, *cidx = nullptr, *cw = nullptr, *cptr = nullptr;
    sc.use(lv.region);
    GESPMM_TRY(sc.get(&cidx, E));
    GESPMM_TRY(sc.get(&cptr, (int64_t)lv.C + 1));
    if (n_of) GESPMM_TRY(sc.get(&cw, E));
    sc.use(Scratch::kTemp);
    const size_t m = sc.mark();
    GESPMM_TRY(sc.get(&iota, E));
    GESPMM_TRY(sc.get(&keys_out, E));
    GESPMM_TRY(sc.get(&order, E));
    if (n_of) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(E)), dim3(256), 0, st, iota, E);
        GESPMM_TRY(sort_pairs<int32_t>(sc, lv.rows.idx, keys_out, iota, order, E, bits_for(lv.C), st));
        hipLaunchKernelGGL(k_gather2, dim3(grid_for(E)), dim3(256), 0, st, n_of, lv.rows.w, order, (int)E, cidx, cw);
    } else {
        // level 0: the sort carries every entry's ROW along (stable: rows ascend inside a column) — no lookup afterwards
        hipLaunchKernelGGL(k_rows_of_entries, dim3(grid_for(E)), dim3(256), 0, st, lv.rows.ptr, lv.R, E, iota);
        GESPMM_TRY(sort_pairs<int32_t>(sc, lv.rows.idx, keys_out, iota, cidx, E, bits_for(lv.C), st));
    }
    hipLaunchKernelGGL(k_lower_bound_ptr, dim3(grid_for((int64_t)lv.C + 1)), dim3(256), 0, st, keys_out, (int)E, lv.C,
                       cptr);
    GESPMM_TRY(hipGetLastError());
    sc.rewind(Scratch::kTemp, m);
    lv.cols.ptr = cptr;
    lv.cols.idx = cidx;
    lv.cols.w = cw;
    return hipSuccess;
}

template <bool ROWS>
hipError_t launch_half_sweep(const LpArgs& base, const int32_t* lists, const int32_t* counts_dev,
                             const int32_t* counts_host, int n_side, unsigned long long* acc, int acc_wgs,
                             int64_t nlabels, hipStream_t st) {
    LpArgs a = base;
    {
        LpSmallClasses c;
        c.lists = lists;
        c.counts = counts_dev;
        c.n_side = n_side;
        int64_t blocks = 0;
        for (int b = 0; b < 4; ++b) {
            blocks += ((int64_t)counts_host[b] * (8 << b) + 255) / 256;
            if (blocks > 0x7fffffff) return hipErrorInvalidValue;  // (2^31 workgroups of 256 lanes: no matrix with 32-bit positions gets there)
            c.blk_end[b] = (int)blocks;
        }
        if (blocks > 0) hipLaunchKernelGGL((k_lp_small<ROWS>), dim3((unsigned)blocks), dim3(256), 0, st, a, c);
    }
    for (int b = 4; b < kBins; ++b) {
        const int cnt = counts_host[b];
        if (cnt == 0) continue;
        a.list = lists + (int64_t)b * n_side;
        a.count = counts_dev + b;
        if (b == 4) hipLaunchKernelGGL((k_lp_wave<ROWS>), dim3((unsigned)cnt), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_lp_dense<ROWS>), dim3((unsigned)std::min(cnt, acc_wgs)), dim3(256), 0, st, a, acc, nlabels);
    }
    return hipGetLastError();
}

}  // namespace

// ====================================================================================================================

hipError_t device_validate_csr(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                               int32_t* max_degree_host, int32_t* bad_host, double* wedge_probe_host, hipStream_t st) {
    Scratch sc(st);
    GESPMM_TRY(sc.init(0, 0, 1024));
    int32_t* out = nullptr;
    GESPMM_TRY(sc.get(&out, 4));
    GESPMM_TRY(hipMemsetAsync(out, 0, 16, st));
    const int64_t n = std::max<int64_t>(M, nnz);
    if (n > 0) hipLaunchKernelGGL(k_validate, dim3(grid_for(n)), dim3(256), 0, st, rowptr, colind, M, K, nnz, out);
    // the structure probe: square matrices only (column c1 must also be a row), same readback as the validation
    const bool probe = wedge_probe_host && M == K && M > 1 && nnz > 0;
    if (probe)
        hipLaunchKernelGGL(k_wedge_probe, dim3(grid_for((int64_t)kProbeRows * kProbePairs)), dim3(256), 0, st, rowptr, colind, M, nnz, out);
    GESPMM_TRY(hipGetLastError());
    int32_t h[4] = {0, 0, 0, 0};
    GESPMM_TRY(fetch(h, out, 4, st));
    *bad_host = h[0];
    *max_degree_host = h[1];
    if (wedge_probe_host) *wedge_probe_host = (probe && h[3] >= 256) ? (double)h[2] / (double)h[3] : -1.0;
    return hipSuccess;
}

// Region sizes of the clustering's arena (upper bounds for level 0, every later level is smaller; see Scratch for what happens beyond them)
static void cluster_arena_regions(int64_t M, int64_t K, int64_t nnz, hipStream_t st, size_t* persist, size_t* level, size_t* temp) {
    size_t sort_tmp = 0;
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, sort_tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)nnz, 0u, 64u, st);
    const size_t E = (size_t)nnz, V = (size_t)(M + K);
    *persist = 4 * (3 * (size_t)M + 4096);
    *level = 4 * (4 * E + 2 * V + 64) + 16 * 256;
    *temp = 48 * E + 56 * V + 2 * sort_tmp + (1 << 20);
}

// gespmm_init: the arena the first plan of a matrix of this size will ask for, made now and parked in the per-device cache (when it
// fits the cache's limit; a larger one would be freed again at once, so nothing is done)
int reserve_analysis_arena(int64_t M, int64_t K, int64_t nnz, void* stream) {
    if (M <= 0 || nnz <= 0) return 0;
    size_t persist = 0, level = 0, temp = 0;
    cluster_arena_regions(M, K, nnz, reinterpret_cast<hipStream_t>(stream), &persist, &level, &temp);
    const size_t total = Scratch::up(persist) + 2 * Scratch::up(level) + Scratch::up(temp);
    if (total > arena_cache_cap()) return 0;
    void* p = nullptr;
    bool cached = false;
    const hipError_t e = arena_acquire(total, &p, &cached);
    if (e != hipSuccess) return (int)e;
    arena_release(p, cached);
    return 0;
}

hipError_t device_cluster_rows(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr, const int32_t* colind,
                               const ClusterOptions& opt, int32_t* perm, ClusterStats* stats, hipStream_t st) {
    if (stats) *stats = ClusterStats{};
    if (M <= 0) return hipSuccess;
    if (nnz <= 0) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(M)), dim3(256), 0, st, perm, M);
        return hipGetLastError();
    }
    Scratch sc(st);
    {
        size_t persist = 0, level = 0, temp = 0;
        cluster_arena_regions(M, K, nnz, st, &persist, &level, &temp);
        GESPMM_TRY(sc.init(persist, level, temp));
    }

    // ---- level 0: the matrix itself (rows = the caller's CSR), its transpose, unit weights
    DLevel lv;
    lv.R = (int32_t)M;
    lv.C = (int32_t)K;
    lv.E = nnz;
    lv.rows.ptr = rowptr;
    lv.rows.idx = colind;
    lv.region = Scratch::kLevelA;
    GESPMM_TRY(build_cols(sc, lv, nullptr, st));
    {
        int32_t* rw = nullptr;
        sc.use(lv.region);
        GESPMM_TRY(sc.get(&rw, M));
        hipLaunchKernelGGL(k_fill, dim3(grid_for(M)), dim3(256), 0, st, rw, M, 1);
        lv.rweight = rw;
    }
    // parents[l][c] = node of level l + 1 that node c of level l belongs to (level 0: c = original row)
    std::vector<int32_t*> parents;
    std::vector<int32_t> parent_count;  // nodes of level l (length of parents[l])
    std::vector<int32_t> cluster_count; // nodes of level l + 1

    int32_t* flags_dev = nullptr;  // {done, changed, ticket, -, counts_rows[kBins], counts_cols[kBins]}
    static_assert(4 + 2 * kBins <= 16, "k_level_init zeroes 16 flags");
    sc.use(Scratch::kPersist);
    GESPMM_TRY(sc.get(&flags_dev, 16));

    long long cap = opt.first_cap > 0 ? opt.first_cap : 256;
    const int max_levels = opt.max_levels > 0 ? opt.max_levels : kDefaultClusterLevels;
    const int sweeps = opt.sweeps > 0 ? opt.sweeps : 5;
    const int stop_percent = opt.stop_percent > 0 ? opt.stop_percent : 97;

    static const bool timing = getenv("GESPMM_PLAN_TIMING") != nullptr;
    auto lap = [&](const char* what, int level) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        static thread_local std::chrono::steady_clock::time_point last;
        const auto now = std::chrono::steady_clock::now();
        if (what) fprintf(stderr, "[plan]   L%d %-16s %8.3f ms\n", level, what, std::chrono::duration<double>(now - last).count() * 1e3);
        last = now;
    };
    lap(nullptr, 0);
    for (int level = 0; level < max_levels; ++level) {
        const int32_t R = lv.R, C = lv.C;
        if (R <= 1 || lv.E == 0) break;
        lap("(cols/setup)", level);
        sc.reset(Scratch::kTemp);
        sc.use(Scratch::kTemp);
        int32_t *rlab = nullptr, *clab = nullptr, *rnext = nullptr, *cnext = nullptr, *size = nullptr, *rlists = nullptr,
                *clists = nullptr;
        GESPMM_TRY(sc.get(&rlab, R));
        GESPMM_TRY(sc.get(&clab, C));
        const size_t after_labels = sc.mark();
        GESPMM_TRY(sc.get(&rnext, R));
        GESPMM_TRY(sc.get(&cnext, C));
        GESPMM_TRY(sc.get(&size, R));
        GESPMM_TRY(sc.get(&rlists, (int64_t)kBins * R));
        GESPMM_TRY(sc.get(&clists, (int64_t)kBins * C));
        // every row node starts as its own label; a node that never moves keeps next == cur for the whole level
        hipLaunchKernelGGL(k_level_init, dim3(grid_for(std::max(R, C))), dim3(256), 0, st, rlab, rnext, size, lv.rweight, R, clab,
                           cnext, C, flags_dev);
        int32_t* done = flags_dev;
        int32_t* changed = flags_dev + 1;
        int32_t* ticket = flags_dev + 2;
        int32_t* rcounts = flags_dev + 4;
        int32_t* ccounts = flags_dev + 4 + kBins;
        hipLaunchKernelGGL(k_bin_nodes, dim3((unsigned)((R + kBinThreads - 1) / kBinThreads)), dim3(kBinThreads), 0, st, lv.rows.ptr,
                           (const int32_t*)nullptr, R, rlists, rcounts);
        hipLaunchKernelGGL(k_bin_nodes, dim3((unsigned)((C + kBinThreads - 1) / kBinThreads)), dim3(kBinThreads), 0, st, lv.cols.ptr,
                           lv.twin, C, clists, ccounts);
        GESPMM_TRY(hipGetLastError());
        int32_t h_counts[2 * kBins];
        GESPMM_TRY(fetch(h_counts, (const int32_t*)(flags_dev + 4), 2 * kBins, st));
        // dense accumulators of the > 2048-entry class (label space = the row labels of this level)
        unsigned long long* acc = nullptr;
        int acc_wgs = 0;
        if (h_counts[kBins - 1] > 0 || h_counts[2 * kBins - 1] > 0) {
            const int want = std::max(h_counts[kBins - 1], h_counts[2 * kBins - 1]);
            int64_t wgs = std::min<int64_t>(want, 64);
            while (wgs > 1 && wgs * (int64_t)R * 8 > (1ll << 30)) wgs >>= 1;
            acc_wgs = (int)wgs;
            GESPMM_TRY(sc.get(&acc, (int64_t)acc_wgs * R));
            GESPMM_TRY(hipMemsetAsync(acc, 0, (size_t)acc_wgs * (size_t)R * 8, st));
        }
        const bool twins = lv.twin != nullptr;
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            LpArgs a;
            a.seed = 0x9e3779b9u * (uint32_t)(level * 16 + sweep + 1);
            a.done = done;
            a.cap = cap;
            a.list = nullptr;
            a.count = nullptr;
            // columns <- heaviest row label (columns with a twin simply carry their cluster's label: k_commit_cols)
            a.adj = lv.cols;
            a.nbr_label = rlab;
            a.cur = clab;
            a.next = cnext;
            a.size = nullptr;
            a.rweight = nullptr;
            a.skip_half = 0;
            GESPMM_TRY(launch_half_sweep<false>(a, clists, ccounts, h_counts + kBins, C, acc, acc_wgs, R, st));
            hipLaunchKernelGGL(k_commit_cols, dim3(grid_for(C)), dim3(256), 0, st, clab, (const int32_t*)cnext, lv.twin,
                               (const int32_t*)rlab, C, (const int32_t*)done);
            // rows <- heaviest column label, within the size cap (sizes: as of the start of the half-sweep)
            a.adj = lv.rows;
            a.nbr_label = clab;
            a.cur = rlab;
            a.next = rnext;
            a.size = size;
            a.rweight = lv.rweight;
            a.skip_half = (twins && sweep + 1 < sweeps) ? 1 : 0;
            GESPMM_TRY(launch_half_sweep<true>(a, rlists, rcounts, h_counts, R, acc, acc_wgs, R, st));
            hipLaunchKernelGGL(k_commit_rows, dim3((unsigned)((R + 1023) / 1024)), dim3(1024), 0, st, rlab, (const int32_t*)rnext,
                               lv.rweight, size, R, changed, ticket, done);
            GESPMM_TRY(hipGetLastError());
        }
        lap("sweeps", level);
        sc.rewind(Scratch::kTemp, after_labels);  // rlab / clab stay

        // ---- contract: compact row labels -> new row nodes, column labels -> new column nodes
        int32_t *used = nullptr, *pos = nullptr, *rid = nullptr, *cid = nullptr, *parent = nullptr;
        const int64_t R1 = (int64_t)R + 1;
        sc.use(Scratch::kPersist);
        GESPMM_TRY(sc.get(&parent, R));
        sc.use(Scratch::kTemp);
        GESPMM_TRY(sc.get(&rid, R));
        GESPMM_TRY(sc.get(&cid, R));
        const size_t after_ids = sc.mark();
        GESPMM_TRY(sc.get(&used, 2 * R1));
        GESPMM_TRY(sc.get(&pos, 2 * R1));
        GESPMM_TRY(hipMemsetAsync(used, 0, (size_t)(2 * R1) * 4, st));
        hipLaunchKernelGGL(k_mark_used, dim3(grid_for(std::max(R, C))), dim3(256), 0, st, (const int32_t*)rlab, R,
                           (const int32_t*)clab, C, used, used + R1);
        GESPMM_TRY(exclusive_scan<int32_t>(sc, used, pos, 2 * R1, st));  // one scan over both halves: the second half's
                                                                        // positions are offset by the first half's total
        hipLaunchKernelGGL(k_make_ids, dim3(grid_for(R)), dim3(256), 0, st, (const int32_t*)used, (const int32_t*)pos,
                           (const int32_t*)(pos + R), R, rid, cid);
        hipLaunchKernelGGL(k_parents, dim3(grid_for(R)), dim3(256), 0, st, (const int32_t*)rlab, (const int32_t*)rid, R, parent);
        GESPMM_TRY(hipGetLastError());
        int32_t tot[2] = {0, 0};  // pos[R] = #row labels in use, pos[2 R1 - 1] = that + #column labels in use
        GESPMM_TRY(hipMemcpyAsync(&tot[0], pos + R, 4, hipMemcpyDeviceToHost, st));
        GESPMM_TRY(hipMemcpyAsync(&tot[1], pos + 2 * R1 - 1, 4, hipMemcpyDeviceToHost, st));
        GESPMM_TRY(hipStreamSynchronize(st));
        const int32_t R2 = tot[0], C2 = tot[1] - tot[0];
        sc.rewind(Scratch::kTemp, after_ids);
        parents.push_back(parent);
        parent_count.push_back(R);
        cluster_count.push_back(R2);
        if (stats) {
            stats->levels = level + 1;
            if (level < 16) stats->clusters[level] = R2;
        }
        lap("relabel", level);
        const bool last = (R2 <= 8 || (int64_t)R2 * 100 > (int64_t)R * stop_percent || level + 1 >= max_levels);
        if (last) break;

        // ---- the next level: members' weights, twins, merged adjacency over the new column nodes
        DLevel nx;
        nx.R = R2;
        nx.C = C2;
        nx.region = lv.region == Scratch::kLevelA ? Scratch::kLevelB : Scratch::kLevelA;
        sc.reset(nx.region);
        int32_t *rw2 = nullptr, *twin2 = nullptr;
        sc.use(nx.region);
        GESPMM_TRY(sc.get(&rw2, R2));
        GESPMM_TRY(sc.get(&twin2, C2));
        sc.use(Scratch::kTemp);
        hipLaunchKernelGGL(k_fill2, dim3(grid_for(std::max(R2, C2))), dim3(256), 0, st, rw2, (int64_t)R2, 0, twin2, (int64_t)C2, -1);
        hipLaunchKernelGGL(k_new_weights, dim3(grid_for(R)), dim3(256), 0, st, (const int32_t*)rlab, (const int32_t*)rid,
                           lv.rweight, R, rw2);
        hipLaunchKernelGGL(k_new_twin, dim3(grid_for(R)), dim3(256), 0, st, (const int32_t*)rid, (const int32_t*)cid, R, twin2);
        nx.rweight = rw2;
        nx.twin = twin2;
        const int64_t E = lv.E;
        const int shift = bits_for(C2);
        unsigned long long *keys = nullptr, *keys_sorted = nullptr;
        int32_t *vals = nullptr, *vals_sorted = nullptr, *hflags = nullptr, *uid = nullptr;
        GESPMM_TRY(sc.get(&keys, E));
        GESPMM_TRY(sc.get(&keys_sorted, E));
        GESPMM_TRY(sc.get(&vals, E));
        GESPMM_TRY(sc.get(&vals_sorted, E));
        hipLaunchKernelGGL(k_emit_edges, dim3(grid_for(E)), dim3(256), 0, st, lv.rows, R, (int)E, (const int32_t*)rlab,
                           (const int32_t*)clab, (const int32_t*)rid, (const int32_t*)cid, (const int32_t*)twin2, R2, shift,
                           keys, vals);
        GESPMM_TRY(hipGetLastError());
        GESPMM_TRY(sort_pairs<unsigned long long>(sc, keys, keys_sorted, vals, vals_sorted, E,
                                                  shift + bits_for((int64_t)R2 + 1), st));
        GESPMM_TRY(sc.get(&hflags, E + 1));
        GESPMM_TRY(sc.get(&uid, E + 1));
        hipLaunchKernelGGL(k_heads, dim3(grid_for(E + 1)), dim3(256), 0, st, (const unsigned long long*)keys_sorted, (int)E,
                           hflags);
        GESPMM_TRY(exclusive_scan<int32_t>(sc, hflags, uid, E + 1, st));
        int32_t U = 0;
        unsigned long long last_key = 0;
        GESPMM_TRY(hipMemcpyAsync(&U, uid + E, 4, hipMemcpyDeviceToHost, st));
        GESPMM_TRY(hipMemcpyAsync(&last_key, keys_sorted + (E - 1), 8, hipMemcpyDeviceToHost, st));
        GESPMM_TRY(hipStreamSynchronize(st));
        const int64_t E2 = ((int64_t)(last_key >> shift) >= R2) ? U - 1 : U;
        int32_t *n_of = nullptr, *idx2 = nullptr, *w2 = nullptr, *ptr2 = nullptr;
        unsigned long long* totals = nullptr;
        sc.use(nx.region);
        GESPMM_TRY(sc.get(&idx2, E2));
        GESPMM_TRY(sc.get(&w2, E2));
        GESPMM_TRY(sc.get(&ptr2, (int64_t)R2 + 1));
        sc.use(Scratch::kTemp);
        GESPMM_TRY(sc.get(&n_of, E2));
        GESPMM_TRY(sc.get(&totals, E2));
        GESPMM_TRY(hipMemsetAsync(totals, 0, (size_t)(E2 > 0 ? E2 : 1) * 8, st));
        hipLaunchKernelGGL(k_run_sums, dim3(grid_for(E)), dim3(256), 0, st, (const unsigned long long*)keys_sorted,
                           (const int32_t*)vals_sorted, (const int32_t*)hflags, (const int32_t*)uid, (int)E, R2, shift, n_of,
                           idx2, totals);
        if (E2 > 0)
            hipLaunchKernelGGL(k_clamp_weights, dim3(grid_for(E2)), dim3(256), 0, st, (const unsigned long long*)totals, (int)E2,
                               w2);
        hipLaunchKernelGGL(k_lower_bound_ptr, dim3(grid_for((int64_t)R2 + 1)), dim3(256), 0, st, (const int32_t*)n_of,
                           (int)E2, R2, ptr2);
        GESPMM_TRY(hipGetLastError());
        nx.E = E2;
        nx.rows.ptr = ptr2;
        nx.rows.idx = idx2;
        nx.rows.w = w2;
        if (E2 > 0) GESPMM_TRY(build_cols(sc, nx, n_of, st));
        lap("contract", level);
        lv = nx;  // the old level's region is rewound when the level after this one is built
        cap *= opt.cap_growth > 1 ? opt.cap_growth : 4;
    }

    // ---- order = lexicographic by (coarsest cluster, ..., finest cluster, original row id). The clusters nest, so it
    //      is enough to rank the clusters of each level top-down — a stable sort of the level's clusters by the rank of
    //      their parent — and finally sort the rows by the rank of their finest cluster (stable: ascending row ids).
    sc.reset(Scratch::kTemp);
    sc.use(Scratch::kTemp);
    if (parents.empty()) {
        hipLaunchKernelGGL(k_iota, dim3(grid_for(M)), dim3(256), 0, st, perm, M);
    } else if ((int)parents.size() <= kOrderMaxLevels) {
        // ... which is the order of the rows by the composite key (cluster of the coarsest level, ..., cluster of the finest level), ties
        // by row id: ONE stable sort of the rows when the fields fit 64 bits (three levels of a com-Amazon-sized graph: 46), else one
        // stable sort per group of levels from the finest group up. (The level-by-level ranking below took three sorts plus their
        // scatter kernels for three levels: 0.26 ms of the headline graph's analysis; this is ~0.07.)
        const int L = (int)parents.size();
        OrderLevels lv;
        for (int l = 0; l < kOrderMaxLevels; ++l) {
            lv.parent[l] = l < L ? parents[l] : nullptr;
            lv.bits[l] = l < L ? bits_for(cluster_count[l]) : 0;
        }
        unsigned long long *keys = nullptr, *keys_out = nullptr;
        int32_t *vals = nullptr, *vals_tmp[2] = {nullptr, nullptr};
        GESPMM_TRY(sc.get(&keys, M));
        GESPMM_TRY(sc.get(&keys_out, M));
        GESPMM_TRY(sc.get(&vals, M));
        GESPMM_TRY(sc.get(&vals_tmp[0], M));
        GESPMM_TRY(sc.get(&vals_tmp[1], M));
        const int32_t* rows = nullptr;  // the rows in the order of the passes so far (nullptr: storage order)
        int lo = 0, pass = 0;
        while (lo < L) {
            int hi = lo, bits = 0;
            while (hi < L && bits + lv.bits[hi] <= 64) bits += lv.bits[hi++];
            int32_t* out = hi == L ? perm : vals_tmp[pass & 1];
            hipLaunchKernelGGL(k_order_keys, dim3(grid_for(M)), dim3(256), 0, st, lv, rows, M, lo, hi, keys, vals);
            GESPMM_TRY(sort_pairs<unsigned long long>(sc, keys, keys_out, vals, out, M, bits, st));
            rows = out;
            lo = hi;
            ++pass;
        }
    } else {
        const int L = (int)parents.size();
        int32_t* rank = nullptr;  // rank of the nodes of level l + 1 (nullptr: their ids)
        int32_t* rank_buf[2] = {nullptr, nullptr};
        GESPMM_TRY(sc.get(&rank_buf[0], parent_count[0]));
        GESPMM_TRY(sc.get(&rank_buf[1], parent_count[0]));
        for (int l = L - 1; l >= 0; --l) {
            const int64_t n = parent_count[l];  // nodes of level l (l == 0: the rows)
            const size_t m = sc.mark();
            int32_t *keys = nullptr, *keys_out = nullptr, *ids = nullptr, *ids_out = nullptr;
            GESPMM_TRY(sc.get(&keys, n));
            GESPMM_TRY(sc.get(&keys_out, n));
            GESPMM_TRY(sc.get(&ids, n));
            hipLaunchKernelGGL(k_parent_rank, dim3(grid_for(n)), dim3(256), 0, st, (const int32_t*)parents[l],
                               (const int32_t*)rank, n, keys, ids);
            if (l == 0) ids_out = perm;
            else GESPMM_TRY(sc.get(&ids_out, n));
            GESPMM_TRY(sort_pairs<int32_t>(sc, keys, keys_out, ids, ids_out, n, bits_for(cluster_count[l]), st));
            if (l > 0) {
                int32_t* nr = rank_buf[l & 1];
                hipLaunchKernelGGL(k_invert, dim3(grid_for(n)), dim3(256), 0, st, (const int32_t*)ids_out, n, nr);
                rank = nr;
            }
            sc.rewind(Scratch::kTemp, m);
        }
    }
    GESPMM_TRY(hipGetLastError());
    lap("order", 99);
    return hipStreamSynchronize(st);  // callers may read perm now
}

hipError_t device_permute_csr(int64_t M, int64_t nnz, const int32_t* rowptr, const int32_t* colind, const int32_t* perm,
                              int32_t* rowptr_p, int32_t* colind_p, int32_t* src_begin, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    Scratch sc(st);
    GESPMM_TRY(sc.init(0, 0, 4 * (size_t)(M + 1) + (4 << 20)));
    int32_t* deg = nullptr;
    GESPMM_TRY(sc.get(&deg, M + 1));
    hipLaunchKernelGGL(k_perm_degrees, dim3(grid_for(M + 1)), dim3(256), 0, st, rowptr, perm, M, deg, src_begin);
    GESPMM_TRY(exclusive_scan<int32_t>(sc, deg, rowptr_p, M + 1, st));
    if (nnz > 0)
        hipLaunchKernelGGL(k_copy_entries, dim3(grid_for(nnz)), dim3(256), 0, st, (const int32_t*)rowptr_p,
                           (const int32_t*)src_begin, colind, (int)M, (int)nnz, colind_p);
    return hipGetLastError();
}

hipError_t device_l2_model(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr, const int32_t* colind, int slices,
                           int64_t window, int64_t max_entries_per_slice, int samples_per_slice, double* hits_host,
                           hipStream_t st) {
    *hits_host = 0.0;
    if (M <= 0 || K <= 0 || nnz <= 0 || slices < 1 || slices > 16 || window < 1) return hipSuccess;
    Scratch sc(st);
    {
        size_t sort_tmp = 0;
        (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, sort_tmp, (const int32_t*)nullptr, (int32_t*)nullptr,
                                                    (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)nnz, 0u, 32u, st);
        GESPMM_TRY(sc.init(0, 0, 20 * (size_t)nnz + 2 * sort_tmp + (1 << 20)));
    }
    SliceInfo* info = nullptr;
    GESPMM_TRY(sc.get(&info, 1));
    hipLaunchKernelGGL(k_slice_bounds, dim3(1), dim3(64), 0, st, rowptr, (int)M, (int)nnz, slices,
                       (long long)max_entries_per_slice, info);
    SliceInfo h;
    GESPMM_TRY(fetch(&h, (const SliceInfo*)info, 1, st));
    int64_t total = 0;
    for (int s = 0; s < slices; ++s) total += h.end[s] - h.begin[s];
    if (total <= 0) return hipSuccess;
    int32_t *cols = nullptr, *poss = nullptr, *cols_sorted = nullptr, *pos_sorted = nullptr, *prev = nullptr, *cnt = nullptr;
    GESPMM_TRY(sc.get(&cols, total));
    GESPMM_TRY(sc.get(&poss, total));
    GESPMM_TRY(sc.get(&cols_sorted, total));
    GESPMM_TRY(sc.get(&pos_sorted, total));
    GESPMM_TRY(sc.get(&prev, nnz));
    GESPMM_TRY(sc.get(&cnt, 32));
    GESPMM_TRY(hipMemsetAsync(cnt, 0, 32 * 4, st));
    hipLaunchKernelGGL(k_model_keys, dim3(grid_for(total)), dim3(256), 0, st, colind, (const SliceInfo*)info, slices, total,
                       cols, poss);
    GESPMM_TRY(sort_pairs<int32_t>(sc, cols, cols_sorted, poss, pos_sorted, total, bits_for(K), st));
    hipLaunchKernelGGL(k_model_prev, dim3(grid_for(total)), dim3(256), 0, st, (const int32_t*)cols_sorted,
                       (const int32_t*)pos_sorted, total, prev);
    hipLaunchKernelGGL(k_model_sample, dim3((unsigned)samples_per_slice, (unsigned)slices), dim3(kModelSampleThreads), 0, st,
                       (const int32_t*)prev, (const SliceInfo*)info, slices, samples_per_slice, (long long)window, cnt,
                       cnt + 16);
    GESPMM_TRY(hipGetLastError());
    int32_t hc[32];
    GESPMM_TRY(fetch(hc, (const int32_t*)cnt, 32, st));
    double hits = 0.0;
    for (int s = 0; s < slices; ++s)
        if (hc[16 + s] > 0) hits += (double)hc[s] / (double)hc[16 + s] * (double)(h.end[s] - h.begin[s]);
    *hits_host = hits / (double)total;
    return hipSuccess;
}

hipError_t device_cut_tasks(int64_t M, const int32_t* rowptr_p, const int64_t budget[2], const int64_t row_floor[2],
                            int32_t* tasks[2], int32_t ntasks_host[2], hipStream_t st) {
    tasks[0] = tasks[1] = nullptr;
    ntasks_host[0] = ntasks_host[1] = 0;
    if (M <= 0) return hipSuccess;
    Scratch sc(st);
    const int64_t nblk = (M + kCutBlock - 1) / kCutBlock;
    GESPMM_TRY(sc.init(0, 0, 48 * (size_t)(M + 1) + 8 * (size_t)nblk * (kMaxRowsPerWave + 1) + (4 << 20)));
    CutParams cp;
    for (int v = 0; v < 2; ++v) {
        cp.budget[v] = budget[v];
        cp.row_floor[v] = row_floor[v];
    }
    long long *cost = nullptr, *P = nullptr;
    int32_t *next = nullptr, *exits = nullptr, *entry = nullptr, *flags = nullptr, *tid_of = nullptr;
    GESPMM_TRY(sc.get(&cost, 2 * (M + 1)));
    GESPMM_TRY(sc.get(&P, 2 * (M + 1)));
    GESPMM_TRY(sc.get(&next, 2 * M));
    GESPMM_TRY(sc.get(&exits, 2 * nblk * kMaxRowsPerWave));
    GESPMM_TRY(sc.get(&entry, 2 * nblk));
    GESPMM_TRY(sc.get(&flags, 2 * (M + 1)));
    GESPMM_TRY(sc.get(&tid_of, 2 * (M + 1)));
    hipLaunchKernelGGL(k_row_costs, dim3(grid_for(M + 1)), dim3(256), 0, st, rowptr_p, M, cp, cost);
    // one scan over both variants: the second prefix carries the first one's total, which cancels in P[j] - P[i]
    GESPMM_TRY(exclusive_scan<long long>(sc, cost, P, 2 * (M + 1), st));
    hipLaunchKernelGGL(k_task_next, dim3(grid_for(M), 2), dim3(256), 0, st, (const long long*)P, M, cp, next);
    hipLaunchKernelGGL(k_task_exits, dim3((unsigned)nblk, 2), dim3(256), 0, st, (const int32_t*)next, M, nblk, exits);
    hipLaunchKernelGGL(k_task_entries, dim3(2), dim3(256), 0, st, (const int32_t*)exits, nblk, entry);
    GESPMM_TRY(hipMemsetAsync(flags, 0, (size_t)(2 * (M + 1)) * 4, st));
    hipLaunchKernelGGL(k_task_mark, dim3((unsigned)nblk, 2), dim3(256), 0, st, (const int32_t*)next, (const int32_t*)entry, M,
                       nblk, flags);
    GESPMM_TRY(exclusive_scan<int32_t>(sc, flags, tid_of, 2 * (M + 1), st));  // second half offset by the first table's size
    int32_t tot[2] = {0, 0};
    GESPMM_TRY(hipMemcpyAsync(&tot[0], tid_of + M, 4, hipMemcpyDeviceToHost, st));
    GESPMM_TRY(hipMemcpyAsync(&tot[1], tid_of + 2 * (M + 1) - 1, 4, hipMemcpyDeviceToHost, st));
    GESPMM_TRY(hipStreamSynchronize(st));
    const int32_t nt[2] = {tot[0], tot[1] - tot[0]};
    // ONE allocation for both tables (the first owns it); int4-aligned halves
    const size_t n0 = ((size_t)(nt[0] > 0 ? nt[0] : 1) + 15) & ~(size_t)15;
    int32_t* out = nullptr;
    GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&out), (n0 + (size_t)(nt[1] > 0 ? nt[1] : 1)) * 16));
    hipLaunchKernelGGL(k_task_write, dim3(grid_for(M)), dim3(256), 0, st, (const int32_t*)flags, (const int32_t*)tid_of,
                       (const int32_t*)next, rowptr_p, M, out);
    // the second table's ids start at nt[0]: shift the base pointer so that id nt[0] lands on its first slot
    hipLaunchKernelGGL(k_task_write, dim3(grid_for(M)), dim3(256), 0, st, (const int32_t*)(flags + (M + 1)),
                       (const int32_t*)(tid_of + (M + 1)), (const int32_t*)(next + M), rowptr_p, M,
                       out + 4 * ((int64_t)n0 - nt[0]));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(out);
        return e;
    }
    tasks[0] = out;
    tasks[1] = out + 4 * n0;
    ntasks_host[0] = nt[0];
    ntasks_host[1] = nt[1];
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------ staging tables (spmm_staged.hip)
//
// Blocks of R consecutive rows of the clustered matrix (R = staged_block_rows(N)). Per block: the block's column indices sorted (one segment of
// a segmented radix sort, payload = position of the entry), runs of equal columns = how often the block uses a B row; the H
// most used ones (>= 2 uses; ties taken in column order, so the tables are the same on every build) get LDS slots.

namespace {

constexpr int kStageHistBins = 256;
constexpr int kStageCounters = 64;  // partial sums of the staged-entry count (power of two)
constexpr int kStageKeysLds = 8192;  // keys of a block kept in LDS by k_stage_select (32 KB)

// First row of block `blk`: blocks of R rows that never straddle a SEGMENT of seg_rows rows (column-slab tables: one segment per slab —
// a launch covers the blocks of one segment; every other table: one segment = the matrix). nb_seg = blocks per segment.
__device__ __forceinline__ int64_t stage_block_begin(int64_t blk, int R, int64_t seg_rows, int64_t nb_seg, int64_t M) {
    const int64_t seg = blk / nb_seg, b = blk - seg * nb_seg;
    int64_t r = seg * seg_rows + b * R;
    const int64_t lim = (seg + 1) * seg_rows;
    if (r > lim) r = lim;
    return r < M ? r : M;
}

__global__ void k_stage_offsets(const int32_t* __restrict__ rowptr_p, int64_t M, int64_t nblk, int R, int64_t seg_rows, int64_t nb_seg,
                                int32_t* __restrict__ blkoff) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nblk) return;
    blkoff[i] = rowptr_p[stage_block_begin(i, R, seg_rows, nb_seg, M)];
}

__global__ void k_stage_iota(int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_stage_select(const int32_t* __restrict__ blkoff, const int32_t* __restrict__ keys,
                                                       const int32_t* __restrict__ idx, int H, int32_t* __restrict__ code,
                                                       int32_t* __restrict__ hot_cols, int32_t* __restrict__ nhot,
                                                       unsigned long long* __restrict__ staged_entries) {
    using BlockScan = rocprim::block_scan<int, 256>;
    __shared__ typename BlockScan::storage_type scan_storage;
    __shared__ int hist[kStageHistBins];
    __shared__ int s_t, s_ngt, s_quota;
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int b = blkoff[blk], e = blkoff[blk + 1];
    hist[tid] = 0;
    // the block's sorted keys are walked run by run, twice, by the thread at each run's head: from LDS when they fit (they do unless the
    // block holds more than kStageKeysLds entries) — the walk is a chain of dependent loads as long as the longest run, and from global
    // memory that chain was most of this kernel's 180-197 us on the headline graph
    __shared__ int32_t s_keys[kStageKeysLds];
    const bool in_lds = e - b <= kStageKeysLds;
    if (in_lds)
        for (int p = b + tid; p < e; p += 256) s_keys[p - b] = keys[p];
    __syncthreads();
    auto key_at = [&](int p) -> int32_t { return in_lds ? s_keys[p - b] : keys[p]; };
    // length of the run starting at p (0 if p is not the head of a run), clamped to the histogram
    auto run_at = [&](int p, int& key) -> int {
        if (p >= e) return 0;
        key = key_at(p);
        if (p > b && key_at(p - 1) == key) return 0;
        int len = 1;
        while (p + len < e && key_at(p + len) == key) ++len;
        return len;
    };
    for (int p0 = b; p0 < e; p0 += 256) {
        int key = 0;
        const int len = run_at(p0 + tid, key);
        if (len >= 2) atomicAdd(&hist[len < kStageHistBins ? len : kStageHistBins - 1], 1);
    }
    __syncthreads();
    {
        // every run longer than t is staged (ngt of them), `quota` runs of length exactly t fill the rest. Thread i looks at length
        // L = 255 - i: a scan over the histogram from the long end finds where the H slots run out (one thread walking the 254 bins
        // was 25 us of dependent LDS reads per workgroup: the kernel took 197 us on the headline graph)
        static_assert(kStageHistBins == 256, "one histogram bin per thread");
        const int L = kStageHistBins - 1 - tid;
        const int h = L >= 2 ? hist[L] : 0;
        int incl = 0;
        BlockScan().inclusive_scan(h, incl, scan_storage, rocprim::plus<int>());
        if (tid == 0) {
            s_t = 1;
            s_quota = 0;
        }
        if (L == 2) s_ngt = incl;  // (if nothing overflows: every run of two or more is staged)
        __syncthreads();
        if (L >= 2 && incl > H && incl - h <= H) {
            s_t = L;
            s_ngt = incl - h;
            s_quota = H - (incl - h);
        }
    }
    __syncthreads();
    const int t = s_t, ngt = s_ngt, quota = s_quota;
    int base_gt = 0, base_eq = 0;
    unsigned long long mine = 0;
    for (int p0 = b; p0 < e; p0 += 256) {
        const int p = p0 + tid;
        int key = 0;
        const int len = run_at(p, key);
        const int lenc = len < kStageHistBins ? len : kStageHistBins - 1;
        const int f_gt = (len >= 2 && lenc > t) ? 1 : 0;
        const int f_eq = (len >= 2 && t >= 2 && lenc == t) ? 1 : 0;
        int ex = 0, total = 0;
        BlockScan().exclusive_scan(f_gt | (f_eq << 16), ex, 0, total, scan_storage, rocprim::plus<int>());
        int slot = -1;
        if (f_gt) slot = base_gt + (ex & 0xffff);
        else if (f_eq && base_eq + (ex >> 16) < quota) slot = ngt + base_eq + (ex >> 16);
        if (len > 0) {
            const int c = slot >= 0 ? (int)(0x80000000u | (unsigned)slot) : key;
            for (int i = 0; i < len; ++i) code[idx[p + i]] = c;
            if (slot >= 0) {
                hot_cols[blk * H + slot] = key;
                mine += (unsigned long long)len;
            }
        }
        base_gt += total & 0xffff;
        base_eq += total >> 16;
        __syncthreads();  // scan_storage is reused by the next tile
    }
    if (tid == 0) nhot[blk] = ngt + (base_eq < quota ? base_eq : quota);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    // (one atomic per workgroup, spread over kStageCounters addresses the host adds up: 14 000 same-address 64-bit atomics, one per
    //  wavefront, were 130 of this kernel's 180 us on the headline graph)
    __shared__ unsigned long long s_mine[4];
    if ((tid & 63) == 0) s_mine[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long all = s_mine[0] + s_mine[1] + s_mine[2] + s_mine[3];
        if (all) atomicAdd(&staged_entries[blk & (kStageCounters - 1)], all);
    }
}

__global__ void k_stage_tasks(const int32_t* __restrict__ rowptr_p, int64_t M, int64_t nblk, int R, int64_t seg_rows, int64_t nb_seg,
                              int kStagedWaves, const int32_t* __restrict__ first_crow, int32_t* __restrict__ tasks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk * kStagedWaves) return;
    const int64_t blk = i / kStagedWaves;
    const int w = (int)(i % kStagedWaves);
    const int b0 = (int)stage_block_begin(blk, R, seg_rows, nb_seg, M);
    const int b1 = (int)stage_block_begin(blk + 1, R, seg_rows, nb_seg, M);
    const int e0 = rowptr_p[b0], e1 = rowptr_p[b1];
    auto bound = [&](int ww) -> int {  // first row of part ww: the block's entries cut into kStagedWaves equal shares
        if (ww <= 0) return b0;
        if (ww >= kStagedWaves) return b1;
        const int target = e0 + (int)(((long long)(e1 - e0) * ww) / kStagedWaves);
        int lo = b0, hi = b1;  // first r in [b0, b1] with rowptr_p[r] >= target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rowptr_p[mid] >= target) hi = mid;
            else lo = mid + 1;
        }
        return lo;
    };
    const int r0 = bound(w), r1 = bound(w + 1);
    // stream positions: entry p of row r is record p + r (one row-end record behind every row)
    // (word 0 — not read by the kernels since the stream carries the rows — holds the C row of the task's first row in column-slab tables)
    const int w0 = first_crow ? first_crow[r0 < M ? r0 : (int)M - 1] : r0;
    reinterpret_cast<int4*>(tasks)[i] = make_int4(w0, r1 - r0, rowptr_p[r0] + r0, rowptr_p[r1] + r1);
}

__device__ __forceinline__ int row_of_entry(const int32_t* __restrict__ rowptr, int M, int q) {
    int lo = 0, hi = M;  // rowptr[lo] <= q < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= q) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The record stream: entry p of row r -> record p + r = {code, value bits}
__global__ void k_stage_interleave(const int32_t* __restrict__ code, const float* __restrict__ val_p,
                                   const int32_t* __restrict__ rowptr_p, int M, int64_t nnz, int32_t* __restrict__ ev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int r = row_of_entry(rowptr_p, M, (int)i);
    reinterpret_cast<int2*>(ev)[i + r] = make_int2(code[i], val_p ? __float_as_int(val_p[i]) : 0x3f800000);
}

// ... row r's end record {kStagedRowEnd, C row of r} at rowptr_p[r + 1] + r, and the padding behind the last row: staged slot 0,
// value +0 (an LDS read, no memory gather; never summed)
// (with_next — column-slab tables: the code's low 30 bits carry the C row of the row BEHIND this one, where a continuing launch
//  picks its accumulators up; the other kernels' address shifts never see such a table)
__global__ void k_stage_rowends(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ perm, int64_t M, int64_t nnz,
                                int with_next, int32_t* __restrict__ ev) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) {
        const int64_t tn = t + 1 < M ? t + 1 : t;
        const int next = with_next ? ((perm ? perm[tn] : (int)tn) & 0x3fffffff) : 0;
        reinterpret_cast<int2*>(ev)[(int64_t)rowptr_p[t + 1] + t] = make_int2(kStagedRowEnd | next, perm ? perm[t] : (int)t);
    }
    else if (t < M + kStagedPad) reinterpret_cast<int2*>(ev)[nnz + t] = make_int2((int)0x80000000u, 0);
}

__global__ void k_stage_values(const float* __restrict__ val_p, const int32_t* __restrict__ rowptr_p, int M, int64_t nnz,
                               int32_t* __restrict__ ev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int r = row_of_entry(rowptr_p, M, (int)i);
    ev[2 * (i + r) + 1] = val_p ? __float_as_int(val_p[i]) : 0x3f800000;
}

}  // namespace

namespace {

__global__ void k_split_degrees(const int32_t* __restrict__ rowptr_p, int64_t M, int limit, int32_t* __restrict__ deg_s,
                                int32_t* __restrict__ is_long) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    int d = 0, l = 0;
    if (i < M) {
        d = rowptr_p[i + 1] - rowptr_p[i];
        if (d > limit) {
            d = 0;
            l = 1;
        }
    }
    deg_s[i] = d;
    is_long[i] = l;
}

__global__ void k_split_compact(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ rowptr_s,
                                const int32_t* __restrict__ colind_p, const float* __restrict__ val_p, int M, int64_t nnz_s,
                                int32_t* __restrict__ colind_s, float* __restrict__ val_s) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz_s) return;
    const int r = row_of_entry(rowptr_s, M, (int)q);
    const int src = rowptr_p[r] + ((int)q - rowptr_s[r]);
    colind_s[q] = colind_p[src];
    if (val_s) val_s[q] = val_p[src];
}

__global__ void k_split_ltasks(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ is_long,
                               const int32_t* __restrict__ lpos, int64_t M, int32_t* __restrict__ ltasks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || !is_long[i]) return;
    reinterpret_cast<int4*>(ltasks)[lpos[i]] = make_int4((int)i, 1, rowptr_p[i], rowptr_p[i + 1]);
}

__global__ void k_stage_values_split(const float* __restrict__ val_p, const int32_t* __restrict__ rowptr_p,
                                     const int32_t* __restrict__ rowptr_s, int M, int64_t nnz_s, int32_t* __restrict__ ev) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz_s) return;
    int bits = 0x3f800000;
    const int r = row_of_entry(rowptr_s, M, (int)q);
    if (val_p) bits = __float_as_int(val_p[rowptr_p[r] + ((int)q - rowptr_s[r])]);
    ev[2 * (q + r) + 1] = bits;  // (record position of entry q of row r in the staged kernel's stream)
}

}  // namespace

hipError_t device_split_long_rows(int64_t M, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p, const float* val_p,
                                  int limit, StagingTables* t, int32_t** colind_s, float** val_s, hipStream_t st) {
    *colind_s = nullptr;
    *val_s = nullptr;
    if (M <= 0 || nnz <= 0) return hipErrorInvalidValue;
    Scratch sc(st);
    GESPMM_TRY(sc.init(0, 0, 16 * (size_t)(M + 1) + (4 << 20)));
    sc.use(Scratch::kTemp);
    int32_t *deg_s = nullptr, *is_long = nullptr, *lpos = nullptr;
    GESPMM_TRY(sc.get(&deg_s, M + 1));
    GESPMM_TRY(sc.get(&is_long, M + 1));
    GESPMM_TRY(sc.get(&lpos, M + 1));
    GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&t->rowptr_s), (size_t)(M + 1) * 4));
    hipLaunchKernelGGL(k_split_degrees, dim3(grid_for(M + 1)), dim3(256), 0, st, rowptr_p, M, limit, deg_s, is_long);
    GESPMM_TRY(exclusive_scan<int32_t>(sc, deg_s, t->rowptr_s, M + 1, st));
    GESPMM_TRY(exclusive_scan<int32_t>(sc, is_long, lpos, M + 1, st));
    int32_t tot[2] = {0, 0};
    GESPMM_TRY(hipMemcpyAsync(&tot[0], t->rowptr_s + M, 4, hipMemcpyDeviceToHost, st));
    GESPMM_TRY(hipMemcpyAsync(&tot[1], lpos + M, 4, hipMemcpyDeviceToHost, st));
    GESPMM_TRY(hipStreamSynchronize(st));
    t->nnz_s = tot[0];
    t->nlong = tot[1];
    GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&t->ltasks), (size_t)(t->nlong > 0 ? t->nlong : 1) * 16));
    GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(colind_s), (size_t)(t->nnz_s > 0 ? t->nnz_s : 1) * 4));
    if (val_p) GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(val_s), (size_t)(t->nnz_s > 0 ? t->nnz_s : 1) * 4));
    if (t->nnz_s > 0)
        hipLaunchKernelGGL(k_split_compact, dim3(grid_for(t->nnz_s)), dim3(256), 0, st, rowptr_p, (const int32_t*)t->rowptr_s,
                           colind_p, val_p, (int)M, t->nnz_s, *colind_s, *val_s);
    hipLaunchKernelGGL(k_split_ltasks, dim3(grid_for(M)), dim3(256), 0, st, rowptr_p, (const int32_t*)is_long,
                       (const int32_t*)lpos, M, t->ltasks);
    GESPMM_TRY(hipGetLastError());
    return hipStreamSynchronize(st);  // the scratch goes out of scope
}

void free_staging(StagingTables* t) {
    if (!t) return;
    // (ev / hot_cols / nhot / tasks are parts of ONE block since round 5: a hipMalloc costs the analysis ~0.1 ms)
    void* ptrs[] = {t->block ? t->block : (void*)t->ev, t->block ? nullptr : (void*)t->hot_cols, t->block ? nullptr : (void*)t->nhot,
                    t->block ? nullptr : (void*)t->tasks, t->rowptr_s, t->ltasks};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    *t = StagingTables();
}

hipError_t device_build_staging(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p,
                                const float* val_p, const int32_t* perm, int R, int H, int waves, int parts, StagingTables* out,
                                hipStream_t st, int64_t seg_rows, bool slab_tables) {
    // (rowptr_p / colind_p / val_p / nnz describe what the staged kernel walks: the clustered matrix, or its copy without hub rows —
    // `out` then already carries rowptr_s / ltasks / nlong / nnz_s from device_split_long_rows, which stay)
    if (M <= 0 || nnz <= 0 || K <= 0 || H <= 0 || R <= 0 || waves <= 0 || waves > kStagedMaxWaves || parts < waves || parts > 64 * waves)
        return hipErrorInvalidValue;
    const int kStagedWaves = parts;  // tasks per block
    if (seg_rows <= 0 || seg_rows > M) seg_rows = M;
    if (M % seg_rows != 0 || (slab_tables && (!perm || M >= (1ll << 30)))) return hipErrorInvalidValue;
    const int64_t nb_seg = (seg_rows + R - 1) / R;  // blocks per segment (column-slab tables: per slab)
    const int64_t nblk = (M / seg_rows) * nb_seg;
    // (Round 3 marked columns whose own row sits far away in the clustered order and gathered them `nt`; level or harmful once the block
    // heights and the clustering depth had settled — profiles/r04/far_marks_by_graph.log — and removed in round 5.)
    if (!staged_stream_fits(M, nnz)) return hipErrorInvalidValue;
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < K) ++bits;
    size_t sort_bytes = 0;
    GESPMM_TRY(rocprim::segmented_radix_sort_pairs(nullptr, sort_bytes, colind_p, (int32_t*)nullptr, (const int32_t*)nullptr,
                                                   (int32_t*)nullptr, (size_t)nnz, (unsigned)nblk, (const int32_t*)nullptr,
                                                   (const int32_t*)nullptr, 0u, (unsigned)bits, st));
    // temporaries (16 bytes per entry + the sort's own) from the analysis arena: a products-sized matrix needs ~2.5 GB here, about
    // what its clustering needed a moment ago — the cached block is reused instead of a multi-GB hipMalloc / hipFree per plan
    Scratch sc(st);
    GESPMM_TRY(sc.init(0, 0, 16 * (size_t)nnz + 4 * (size_t)(M + nblk) + sort_bytes + (8 << 20)));
    sc.use(Scratch::kTemp);
    int32_t *blkoff = nullptr, *keys = nullptr, *idx_in = nullptr, *idx_out = nullptr, *code = nullptr;
    unsigned long long* staged = nullptr;
    char* tmp = nullptr;
    GESPMM_TRY(sc.get(&blkoff, nblk + 1));
    GESPMM_TRY(sc.get(&keys, nnz));
    GESPMM_TRY(sc.get(&idx_in, nnz));
    GESPMM_TRY(sc.get(&idx_out, nnz));
    GESPMM_TRY(sc.get(&code, nnz));
    GESPMM_TRY(sc.get(&staged, kStageCounters));
    GESPMM_TRY(sc.get(&tmp, (int64_t)(sort_bytes ? sort_bytes : 256)));
    StagingTables t;  // the four tables built here; merged into *out on success
    auto body = [&]() -> hipError_t {
        GESPMM_TRY(hipMemsetAsync(staged, 0, 8 * kStageCounters, st));
        hipLaunchKernelGGL(k_stage_offsets, dim3(grid_for(nblk + 1)), dim3(256), 0, st, rowptr_p, M, nblk, R, seg_rows, nb_seg, blkoff);
        hipLaunchKernelGGL(k_stage_iota, dim3(grid_for(nnz)), dim3(256), 0, st, idx_in, nnz);
        GESPMM_TRY(rocprim::segmented_radix_sort_pairs(tmp, sort_bytes, colind_p, keys, (const int32_t*)idx_in, idx_out, (size_t)nnz,
                                                       (unsigned)nblk, (const int32_t*)blkoff, (const int32_t*)blkoff + 1, 0u,
                                                       (unsigned)bits, st));
        // the tables themselves belong to the plan: blocks of their own
        {
            auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t b_hot = up((size_t)nblk * H * 4), b_nhot = up((size_t)nblk * 4), b_tasks = up((size_t)nblk * kStagedWaves * 16),
                         b_ev = up((size_t)(nnz + M + kStagedPad) * 8);
            GESPMM_TRY(hipMalloc(&t.block, b_ev + b_tasks + b_hot + b_nhot));
            char* base = reinterpret_cast<char*>(t.block);
            t.ev = reinterpret_cast<int32_t*>(base);
            t.tasks = reinterpret_cast<int32_t*>(base + b_ev);
            t.hot_cols = reinterpret_cast<int32_t*>(base + b_ev + b_tasks);
            t.nhot = reinterpret_cast<int32_t*>(base + b_ev + b_tasks + b_hot);
        }
        GESPMM_TRY(hipMemsetAsync(t.hot_cols, 0xFF, (size_t)nblk * H * 4, st));  // unused slots: -1 (the kernel copies nothing for them)
        hipLaunchKernelGGL(k_stage_select, dim3((unsigned)nblk), dim3(256), 0, st, (const int32_t*)blkoff, (const int32_t*)keys,
                           (const int32_t*)idx_out, H, code, t.hot_cols, t.nhot, staged);
        hipLaunchKernelGGL(k_stage_tasks, dim3(grid_for(nblk * kStagedWaves)), dim3(256), 0, st, rowptr_p, M, nblk, R, seg_rows, nb_seg,
                           kStagedWaves, slab_tables ? perm : (const int32_t*)nullptr, t.tasks);
        hipLaunchKernelGGL(k_stage_interleave, dim3(grid_for(nnz)), dim3(256), 0, st, (const int32_t*)code, val_p, rowptr_p, (int)M, nnz,
                           t.ev);
        hipLaunchKernelGGL(k_stage_rowends, dim3(grid_for(M + kStagedPad)), dim3(256), 0, st, rowptr_p, perm, M, nnz, slab_tables ? 1 : 0, t.ev);
        GESPMM_TRY(hipGetLastError());
        unsigned long long hs[kStageCounters], h = 0;
        GESPMM_TRY(fetch(hs, (const unsigned long long*)staged, kStageCounters, st));  // (synchronises: the temporaries may go)
        for (int i = 0; i < kStageCounters; ++i) h += hs[i];
        t.nblocks = (int32_t)nblk;
        t.staged_fraction = (double)h / (double)nnz;
        return hipSuccess;
    };
    const hipError_t e = body();
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        free_staging(&t);
        return e;
    }
    out->block = t.block;
    out->ev = t.ev;
    out->hot_cols = t.hot_cols;
    out->nhot = t.nhot;
    out->tasks = t.tasks;
    out->nblocks = t.nblocks;
    out->waves = waves;
    out->slots = H;
    out->staged_fraction = t.staged_fraction;
    if (!out->rowptr_s) out->nnz_s = nnz;
    return hipSuccess;
}

hipError_t device_staging_set_values(const StagingTables& t, const float* val_p, const int32_t* rowptr_p, int64_t M, int64_t nnz,
                                     hipStream_t st) {
    if (!t.ev) return hipSuccess;
    if (t.rowptr_s) {
        if (t.nnz_s > 0)
            hipLaunchKernelGGL(k_stage_values_split, dim3(grid_for(t.nnz_s)), dim3(256), 0, st, val_p, rowptr_p,
                               (const int32_t*)t.rowptr_s, (int)M, t.nnz_s, t.ev);
    } else if (nnz > 0) {
        hipLaunchKernelGGL(k_stage_values, dim3(grid_for(nnz)), dim3(256), 0, st, val_p, rowptr_p, (int)M, nnz, t.ev);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ column-slab view (round 6)
//
// Dense clustered matrices (a reddit-shaped community refers to ~100 000 distinct B rows: a block's 160 LDS slots cover a seventh of its
// entries): the matrix is cut into P ascending COLUMN ranges, each range of each row becomes a row of a VIEW with P x M rows (slab-major:
// view row p * M + r = the entries of clustered row r whose column falls in slab p), and the staged tables are built on the view with
// blocks that never straddle a slab: per (block of rows, slab) its own list of staged columns. Slab p is one launch over its blocks; launch
// p > 0 continues the rows from the partial sums in C. Needs rows with non-decreasing columns (slab order == CSR order in every row).

namespace {

__device__ __forceinline__ int row_lower_bound(const int32_t* __restrict__ colind, int b, int e, int64_t key) {  // first q in [b, e): col >= key
    while (b < e) {
        const int mid = (b + e) >> 1;
        if ((int64_t)colind[mid] >= key) e = mid;
        else b = mid + 1;
    }
    return b;
}

__global__ void k_slab_sorted(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int M, int64_t nnz,
                              int32_t* __restrict__ unsorted) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (q >= nnz) return;
    if (colind[q - 1] > colind[q]) {
        const int r = row_of_entry(rowptr, M, (int)q);
        if (rowptr[r] != (int)q) *unsorted = 1;  // a descent inside a row
    }
}

__global__ void k_slab_counts(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, int64_t M, int64_t K, int P,
                              int32_t* __restrict__ cnt_v, int32_t* __restrict__ max_row) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (i < M * P) {
        const int64_t p = i / M, r = i - p * M;
        const int b = rowptr[r], e = rowptr[r + 1];
        const int lo = row_lower_bound(colind, b, e, (K * p) / P);
        const int hi = row_lower_bound(colind, lo, e, (K * (p + 1)) / P);
        c = hi - lo;
        cnt_v[i] = c;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_down(c, o);
        c = other > c ? other : c;
    }
    if ((threadIdx.x & 63) == 0 && c > 0) atomicMax(max_row, c);
}

__global__ void k_slab_scatter(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const float* __restrict__ val,
                               const int32_t* __restrict__ rowptr_v, int64_t M, int64_t K, int P, int64_t nnz,
                               int32_t* __restrict__ colind_v, float* __restrict__ val_v, int32_t* __restrict__ src_v) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const int r = row_of_entry(rowptr, (int)M, (int)q);
    const int64_t c = colind[q];
    int64_t p = (c * P) / K;
    while (p > 0 && c < (K * p) / P) --p;
    while (p + 1 < P && c >= (K * (p + 1)) / P) ++p;
    const int s = row_lower_bound(colind, rowptr[r], rowptr[r + 1], (K * p) / P);
    const int64_t dest = (int64_t)rowptr_v[p * M + r] + (q - s);
    colind_v[dest] = (int32_t)c;
    if (val_v) val_v[dest] = val[q];
    src_v[dest] = (int32_t)q;
}

__global__ void k_slab_perm(const int32_t* __restrict__ perm, int64_t M, int P, int32_t* __restrict__ perm_v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * P) return;
    const int64_t r = i % M;
    perm_v[i] = perm ? perm[r] : (int32_t)r;
}

__global__ void k_stage_values_view(const float* __restrict__ val_p, const int32_t* __restrict__ src_v, const int32_t* __restrict__ rowptr_v,
                                    int Mv, int64_t nnz, int32_t* __restrict__ ev) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const int r = row_of_entry(rowptr_v, Mv, (int)q);
    ev[2 * (q + r) + 1] = val_p ? __float_as_int(val_p[src_v[q]]) : 0x3f800000;
}

}  // namespace

void free_slab_view(SlabView* v, bool temporaries_only) {
    if (!v) return;
    for (void* q : {(void*)v->colind_v, (void*)v->val_v, (void*)v->perm_v})
        if (q) (void)hipFree(q);
    v->colind_v = nullptr;
    v->val_v = nullptr;
    v->perm_v = nullptr;
    if (temporaries_only) return;
    for (void* q : {(void*)v->rowptr_v, (void*)v->src_v})
        if (q) (void)hipFree(q);
    *v = SlabView();
}

hipError_t device_build_slab_view(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p, const float* val_p,
                                  const int32_t* perm, int P, SlabView* out, hipStream_t st) {
    *out = SlabView();
    if (M <= 0 || K <= 0 || nnz <= 0 || P < 2 || P > 64 || M * P >= (1ll << 30) || nnz >= (1ll << 31) - 64) return hipErrorInvalidValue;
    const int64_t Mv = M * P;
    Scratch sc(st);
    GESPMM_TRY(sc.init(0, 0, 4 * (size_t)(Mv + 1) + (4 << 20)));
    sc.use(Scratch::kTemp);
    int32_t *cnt_v = nullptr, *flags = nullptr;
    GESPMM_TRY(sc.get(&cnt_v, Mv + 1));
    GESPMM_TRY(sc.get(&flags, 2));
    SlabView v;
    auto body = [&]() -> hipError_t {
        GESPMM_TRY(hipMemsetAsync(flags, 0, 8, st));
        GESPMM_TRY(hipMemsetAsync(cnt_v + Mv, 0, 4, st));
        if (nnz > 1) hipLaunchKernelGGL(k_slab_sorted, dim3(grid_for(nnz)), dim3(256), 0, st, rowptr_p, colind_p, (int)M, nnz, flags);
        hipLaunchKernelGGL(k_slab_counts, dim3(grid_for(Mv)), dim3(256), 0, st, rowptr_p, colind_p, M, K, P, cnt_v, flags + 1);
        GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&v.rowptr_v), (size_t)(Mv + 1) * 4));
        GESPMM_TRY(exclusive_scan<int32_t>(sc, cnt_v, v.rowptr_v, Mv + 1, st));
        int32_t h[2] = {0, 0};
        GESPMM_TRY(fetch(h, (const int32_t*)flags, 2, st));
        v.sorted = h[0] ? 0 : 1;
        v.max_row = h[1];
        v.slabs = P;
        if (!v.sorted) return hipSuccess;  // (no view: the caller keeps its other kernels)
        GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&v.colind_v), (size_t)nnz * 4));
        GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&v.src_v), (size_t)nnz * 4));
        GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&v.perm_v), (size_t)Mv * 4));
        if (val_p) GESPMM_TRY(hipMalloc(reinterpret_cast<void**>(&v.val_v), (size_t)nnz * 4));
        hipLaunchKernelGGL(k_slab_scatter, dim3(grid_for(nnz)), dim3(256), 0, st, rowptr_p, colind_p, val_p, (const int32_t*)v.rowptr_v, M, K,
                           P, nnz, v.colind_v, v.val_v, v.src_v);
        hipLaunchKernelGGL(k_slab_perm, dim3(grid_for(Mv)), dim3(256), 0, st, perm, M, P, v.perm_v);
        GESPMM_TRY(hipGetLastError());
        return hipStreamSynchronize(st);
    };
    const hipError_t e = body();
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        free_slab_view(&v, false);
        return e;
    }
    *out = v;
    return hipSuccess;
}

hipError_t device_slab_set_values(const StagingTables& t, const SlabView& v, const float* val_p, int64_t Mv, int64_t nnz, hipStream_t st) {
    if (!t.ev || !v.src_v || nnz <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_stage_values_view, dim3(grid_for(nnz)), dim3(256), 0, st, val_p, (const int32_t*)v.src_v, (const int32_t*)v.rowptr_v,
                       (int)Mv, nnz, t.ev);
    return hipGetLastError();
}

}  // namespace gespmm
