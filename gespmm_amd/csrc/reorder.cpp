// reorder.cpp — plan-time row clustering (host side, no HIP).
//
// Why: C[r,:] = sum_p val[p] * B[colind[p],:] gathers 4*N bytes of B per non-zero. When B does not fit
// the 4 MiB L2 of an XCD, every use of a B row by rows that run at unrelated times is an L2 miss that
// crosses the fabric (round 1: 3.1x the algorithmic bytes, 17 % L2 hits). The order in which ROWS are
// processed is free — each row's sum is computed by one lane group in its own CSR order whatever
// happens around it — so a processing order that puts rows sharing neighbours next to each other turns
// those re-fetches into L2 hits without touching a single bit of the result.
//
// The reference has no counterpart (its kernels walk rows in storage order, spmm_test.cu:97-236); this
// is the "analysis" stage of a vendor SpMM (rocsparse_spmm_stage_preprocess, cusparseSpMM_preprocess).
//
// Method: multi-level label propagation on the bipartite graph rows <-> columns.
//   * a level holds row nodes and column nodes with weighted edges (level 0: the matrix itself);
//   * one sweep = columns take the heaviest label among their rows, then rows take the heaviest label
//     among their columns (two-colour semi-synchronous propagation: it cannot oscillate), subject to a
//     size cap on the number of original rows a label may own;
//   * after a few sweeps nodes with equal labels are contracted into one node and the next level
//     clusters the clusters: a column node that shares its label with a row node is that cluster's
//     "twin" — edges between a cluster and its own twin are dropped (they would only vote for staying
//     alone) and the twin simply carries its cluster's label, so from level 1 on a row cluster adopts
//     the label of the cluster it is most heavily connected to (half of the nodes per sweep, chosen by
//     hash, which damps the label swaps of synchronous propagation);
//   * the processing order is the lexicographic order of a row's labels from the coarsest level down
//     (so rows of one fine cluster are adjacent, fine clusters of one coarse cluster are adjacent, ...).
// Everything is deterministic (hash tie-breaks, snapshot semantics inside a half-sweep), and the
// half-sweeps are data-parallel over nodes (std::thread, results independent of the thread count).

#include "reorder.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif
#include <vector>

namespace gespmm {

namespace {

inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// One side of a bipartite level in CSR form: node -> (neighbour on the other side, weight).
struct Adj {
    std::vector<int64_t> ptr;
    std::vector<int32_t> idx;
    std::vector<int32_t> w;  // empty: all ones
};

struct Level {
    int32_t R = 0, C = 0;          // row nodes, column nodes
    Adj rows, cols;                // row -> columns, column -> rows
    std::vector<int32_t> rweight;  // original rows owned by each row node
    std::vector<int32_t> twin;     // levels >= 1: per column node, the row node that carried the same label (-1: none)
};

// Worker t of a parallel region runs on the t-th CPU of the calling thread's 8-CPU neighbourhood (one L3 / memory-local
// group on the hosts this runs on), when the process is allowed there: on a 256-CPU box unpinned workers land on other
// sockets' cores, read the graph through the interconnect and are SLOWER than one thread (193 ms on one thread, 257 on two,
// 182 on 32; pinned: profiles/r02/cluster_time.log). A hint only: failures are ignored, results never depend on it.
inline void pin_near(int base_cpu, int t) {
#if defined(__linux__)
    if (base_cpu < 0) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET((base_cpu & ~7) + (t & 7), &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
#else
    (void)base_cpu;
    (void)t;
#endif
}

template <typename F>
void parallel_for(int64_t n, int threads, F&& body) {
    if (threads > 1 && n < (int64_t)threads * 32768) threads = (int)(n / 32768);  // a worker must be worth its spawn: small graphs run on one thread (pubmed 12 vs 17 ms)
    if (threads <= 1) {
        body(0, n, 0);
        return;
    }
#if defined(__linux__)
    const int base_cpu = sched_getcpu();
#else
    const int base_cpu = -1;
#endif
    std::vector<std::thread> pool;
    const int64_t chunk = (n + threads - 1) / threads;
    for (int t = 1; t < threads; ++t) {
        const int64_t lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([&, lo, hi, t]() {
            pin_near(base_cpu, t);
            body(lo, hi, t);
        });
    }
    body(0, std::min(n, chunk), 0);  // the caller is worker 0 (and stays where it is)
    for (auto& th : pool) th.join();
}

// Transpose an adjacency (n_from nodes -> n_to nodes) by counting sort. (Serial on purpose: a version in which every worker
// owned a range of destination nodes and scanned all entries measured level with this one — profiles/r02/cluster_time.log.)
void transpose(const Adj& a, int32_t n_from, int32_t n_to, Adj& out) {
    out.ptr.assign((size_t)n_to + 1, 0);
    const int64_t ne = a.ptr[n_from];
    for (int64_t e = 0; e < ne; ++e) out.ptr[(size_t)a.idx[e] + 1]++;
    for (int32_t i = 0; i < n_to; ++i) out.ptr[i + 1] += out.ptr[i];
    out.idx.resize((size_t)ne);
    if (!a.w.empty()) out.w.resize((size_t)ne);
    else out.w.clear();
    std::vector<int64_t> cur(out.ptr.begin(), out.ptr.end() - 1);
    for (int32_t r = 0; r < n_from; ++r)
        for (int64_t e = a.ptr[r]; e < a.ptr[r + 1]; ++e) {
            const int64_t q = cur[a.idx[e]]++;
            out.idx[q] = r;
            if (!a.w.empty()) out.w[q] = a.w[e];
        }
}

// Heaviest label among a node's neighbours. `acc` is a zeroed scratch array over the label space that
// is left zeroed again; `own` (>= 0) wins ties it takes part in; `allowed(label)` filters candidates.
template <typename Allowed>
inline int32_t heaviest_label(const Adj& a, int64_t node, const int32_t* nbr_label, int32_t own, uint32_t seed,
                              std::vector<int64_t>& acc, std::vector<int32_t>& touched, Allowed&& allowed) {
    touched.clear();
    for (int64_t e = a.ptr[node]; e < a.ptr[node + 1]; ++e) {
        const int32_t L = nbr_label[a.idx[e]];
        if (L < 0) continue;
        if (acc[L] == 0) touched.push_back(L);
        acc[L] += a.w.empty() ? 1 : a.w[e];
    }
    int32_t best = own;
    int64_t best_w = (own >= 0) ? acc[own] : 0;
    uint32_t best_h = 0;
    bool best_is_own = true;
    for (int32_t L : touched) {
        if (L == own) continue;
        const int64_t w = acc[L];
        if (w < best_w || (best_is_own && w == best_w)) continue;  // the own label keeps every tie it is part of
        if (!allowed(L)) continue;
        const uint32_t h = mix32((uint32_t)L ^ seed);
        if (w > best_w || h < best_h) {  // ties between foreign labels: smallest hash, not smallest id
            best = L;
            best_w = w;
            best_h = h;
            best_is_own = false;
        }
    }
    for (int32_t L : touched) acc[L] = 0;
    return best;
}

}  // namespace

int cluster_rows(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const ClusterOptions& opt,
                 int32_t* perm, ClusterStats* stats, int32_t* top_labels) {
    if (M < 0 || K < 0 || (M > 0 && (!rowptr || !perm))) return -1;
    if (stats) *stats = ClusterStats{};
    if (M == 0) return 0;
    const int64_t nnz = rowptr[M];
    if (nnz < 0 || (nnz > 0 && !colind)) return -1;
    int threads = opt.threads > 0 ? opt.threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 8) threads = 8;  // one L3 neighbourhood (parallel_for pins there); more only adds remote traffic

    // ---- level 0: the matrix (entries outside [0, K) are ignored; duplicates simply weigh twice)
    Level lv;
    lv.R = (int32_t)M;
    lv.C = (int32_t)K;
    lv.rows.ptr.resize((size_t)M + 1);
    {
        std::vector<char> bad_t((size_t)threads, 0);
        parallel_for(nnz, threads, [&](int64_t lo, int64_t hi, int t) {
            char bad = 0;
            for (int64_t p = lo; p < hi; ++p) bad |= (char)((uint32_t)colind[p] >= (uint64_t)K);
            bad_t[t] = bad;
        });
        bool bad = rowptr[0] != 0;
        for (char b : bad_t) bad = bad || b;
        if (!bad) {  // the usual case: a straight copy
            lv.rows.idx.resize((size_t)nnz);
            parallel_for(M + 1, threads, [&](int64_t lo, int64_t hi, int) {
                for (int64_t r = lo; r < hi; ++r) lv.rows.ptr[r] = rowptr[r];
            });
            parallel_for(nnz, threads, [&](int64_t lo, int64_t hi, int) {
                std::memcpy(lv.rows.idx.data() + lo, colind + lo, (size_t)(hi - lo) * sizeof(int32_t));
            });
        } else {
            lv.rows.idx.reserve((size_t)nnz);
            lv.rows.ptr[0] = 0;
            for (int64_t r = 0; r < M; ++r) {
                for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                    const int32_t c = colind[p];
                    if (c >= 0 && c < K) lv.rows.idx.push_back(c);
                }
                lv.rows.ptr[r + 1] = (int64_t)lv.rows.idx.size();
            }
        }
    }
    transpose(lv.rows, lv.R, lv.C, lv.cols);
    lv.rweight.assign((size_t)M, 1);

    std::vector<int32_t> node_of_row((size_t)M);  // current-level row node of every original row
    std::iota(node_of_row.begin(), node_of_row.end(), 0);
    std::vector<std::vector<int32_t>> level_labels;  // per level: row-node id of every original row AFTER contraction

    int64_t cap = opt.first_cap > 0 ? opt.first_cap : 256;
    const int max_levels = opt.max_levels > 0 ? opt.max_levels : 6;  // == plan_device.hip kDefaultClusterLevels (profiles/r04/like_regression.log)
    const int sweeps = opt.sweeps > 0 ? opt.sweeps : 5;

    for (int level = 0; level < max_levels; ++level) {
        const int32_t R = lv.R, C = lv.C;
        if (R <= 1 || lv.rows.ptr[R] == 0) break;
        std::vector<int32_t> rlab((size_t)R), clab((size_t)C, -1), next((size_t)std::max(R, C));
        std::iota(rlab.begin(), rlab.end(), 0);
        std::vector<int64_t> size((size_t)R);
        std::vector<std::vector<int64_t>> acc((size_t)threads);
        std::vector<std::vector<int32_t>> touched((size_t)threads);
        for (int sweep = 0; sweep < sweeps; ++sweep) {
            const uint32_t seed = 0x9e3779b9u * (uint32_t)(level * 16 + sweep + 1);
            // columns <- heaviest row label (no cap: columns are not ordered, they only relay labels)
            const bool twins = !lv.twin.empty();
            parallel_for(C, threads, [&](int64_t lo, int64_t hi, int t) {
                if (acc[t].empty()) acc[t].assign((size_t)R, 0);
                for (int64_t c = lo; c < hi; ++c) {
                    if (twins && lv.twin[c] >= 0) next[c] = rlab[lv.twin[c]];
                    else
                        next[c] = heaviest_label(lv.cols, c, rlab.data(), clab[c], seed, acc[t], touched[t],
                                                 [](int32_t) { return true; });
                }
            });
            std::copy(next.begin(), next.begin() + C, clab.begin());
            // rows <- heaviest column label, within the size cap (sizes: snapshot at the start of the half-sweep)
            std::fill(size.begin(), size.end(), 0);
            for (int32_t r = 0; r < R; ++r) size[rlab[r]] += lv.rweight[r];
            std::vector<int64_t> changed_t((size_t)threads, 0);
            parallel_for(R, threads, [&](int64_t lo, int64_t hi, int t) {
                if (acc[t].empty()) acc[t].assign((size_t)R, 0);
                int64_t ch = 0;
                for (int64_t r = lo; r < hi; ++r) {
                    const int64_t w = lv.rweight[r];
                    const int32_t own = rlab[r];
                    if (twins && sweep + 1 < sweeps && (mix32((uint32_t)r * 0x85ebca6bu ^ seed) & 1u)) {
                        next[r] = own;  // this sweep belongs to the other half
                        continue;
                    }
                    const int32_t L = heaviest_label(lv.rows, r, clab.data(), own, seed, acc[t], touched[t],
                                                     [&](int32_t cand) { return size[cand] + w <= cap; });
                    next[r] = L;
                    ch += (L != own);
                }
                changed_t[t] = ch;
            });
            std::copy(next.begin(), next.begin() + R, rlab.begin());
            int64_t changed = 0;
            for (int64_t c : changed_t) changed += c;
            if (changed * 400 < R) break;  // < 0.25 % of the row nodes moved
        }
        // ---- contract: compact row labels -> new row nodes, column labels -> new column nodes
        std::vector<int32_t> rid((size_t)R, -1), cid((size_t)R + 1, -1);
        int32_t R2 = 0, C2 = 0;
        for (int32_t r = 0; r < R; ++r) rid[rlab[r]] = 0;
        for (int32_t L = 0; L < R; ++L)
            if (rid[L] == 0) rid[L] = R2++;
        // columns nobody labelled (no rows) keep -1 and vanish; labels live in the row-label space [0, R)
        for (int32_t c = 0; c < C; ++c)
            if (clab[c] >= 0) cid[clab[c]] = 0;
        for (int32_t L = 0; L < R; ++L)
            if (cid[L] == 0) cid[L] = C2++;
        for (int64_t r = 0; r < M; ++r) node_of_row[r] = rid[rlab[node_of_row[r]]];
        level_labels.push_back(node_of_row);
        if (stats) {
            stats->levels = level + 1;
            if (level < 16) stats->clusters[level] = R2;
        }
        if (R2 <= 8 || (int64_t)R2 * 100 > (int64_t)R * (opt.stop_percent > 0 ? opt.stop_percent : 97)) break;  // nothing left to merge
        // members of every new row node, then their merged adjacency over new column nodes
        Level nx;
        nx.R = R2;
        nx.C = C2;
        nx.rweight.assign((size_t)R2, 0);
        std::vector<int64_t> mptr((size_t)R2 + 1, 0);
        for (int32_t r = 0; r < R; ++r) {
            const int32_t n = rid[rlab[r]];
            mptr[(size_t)n + 1]++;
            nx.rweight[n] += lv.rweight[r];
        }
        for (int32_t n = 0; n < R2; ++n) mptr[n + 1] += mptr[n];
        std::vector<int32_t> members((size_t)R);
        {
            std::vector<int64_t> cur(mptr.begin(), mptr.end() - 1);
            for (int32_t r = 0; r < R; ++r) members[cur[rid[rlab[r]]]++] = r;
        }
        nx.twin.assign((size_t)C2, -1);
        for (int32_t L = 0; L < R; ++L)
            if (cid[L] >= 0) nx.twin[cid[L]] = rid[L];
        nx.rows.ptr.assign((size_t)R2 + 1, 0);
        {
            // every worker merges the rows of a contiguous range of new nodes into its own buffers (dense accumulator over
            // the new column nodes + list of touched ones, entries in first-touch order); the buffers are then laid end to
            // end in node order — the same adjacency whatever the number of workers
            std::vector<std::vector<int32_t>> idx_t((size_t)threads), w_t((size_t)threads);
            std::vector<int64_t> lo_t((size_t)threads, 0), hi_t((size_t)threads, 0);
            parallel_for(R2, threads, [&](int64_t lo, int64_t hi, int t) {
                lo_t[t] = lo;
                hi_t[t] = hi;
                std::vector<int64_t> a2((size_t)C2, 0);
                std::vector<int32_t> t2;
                auto& oi = idx_t[t];
                auto& ow = w_t[t];
                for (int64_t n = lo; n < hi; ++n) {
                    t2.clear();
                    for (int64_t m = mptr[n]; m < mptr[n + 1]; ++m) {
                        const int32_t r = members[m];
                        for (int64_t e = lv.rows.ptr[r]; e < lv.rows.ptr[r + 1]; ++e) {
                            const int32_t cl = clab[lv.rows.idx[e]];
                            if (cl < 0) continue;
                            const int32_t cn = cid[cl];
                            if (nx.twin[cn] == (int32_t)n) continue;  // a cluster's edges to its own columns
                            if (a2[cn] == 0) t2.push_back(cn);
                            a2[cn] += lv.rows.w.empty() ? 1 : lv.rows.w[e];
                        }
                    }
                    for (int32_t cn : t2) {
                        oi.push_back(cn);
                        ow.push_back((int32_t)std::min<int64_t>(a2[cn], 0x7fffffff));
                        a2[cn] = 0;
                    }
                    nx.rows.ptr[(size_t)n + 1] = (int64_t)t2.size();  // degree for now, prefix-summed below
                }
            });
            for (int32_t n = 0; n < R2; ++n) nx.rows.ptr[(size_t)n + 1] += nx.rows.ptr[n];
            nx.rows.idx.resize((size_t)nx.rows.ptr[R2]);
            nx.rows.w.resize((size_t)nx.rows.ptr[R2]);
            for (int t = 0; t < threads; ++t) {
                if (idx_t[t].empty()) continue;
                const int64_t at = nx.rows.ptr[lo_t[t]];
                std::memcpy(nx.rows.idx.data() + at, idx_t[t].data(), idx_t[t].size() * sizeof(int32_t));
                std::memcpy(nx.rows.w.data() + at, w_t[t].data(), w_t[t].size() * sizeof(int32_t));
            }
        }
        transpose(nx.rows, nx.R, nx.C, nx.cols);
        lv = std::move(nx);
        cap *= opt.cap_growth > 1 ? opt.cap_growth : 4;
    }

    // ---- order: stable counting sorts from the finest level to the coarsest (LSD) = lexicographic by
    //      (coarsest label, ..., finest label, original row id)
    std::vector<int32_t> order((size_t)M), tmp((size_t)M);
    std::iota(order.begin(), order.end(), 0);
    for (const auto& lab : level_labels) {
        int32_t nl = 0;
        for (int32_t v : lab) nl = std::max(nl, v + 1);
        std::vector<int64_t> cnt((size_t)nl + 1, 0);
        for (int64_t i = 0; i < M; ++i) cnt[(size_t)lab[order[i]] + 1]++;
        for (int32_t l = 0; l < nl; ++l) cnt[l + 1] += cnt[l];
        for (int64_t i = 0; i < M; ++i) tmp[cnt[lab[order[i]]]++] = order[i];
        order.swap(tmp);
    }
    std::memcpy(perm, order.data(), (size_t)M * sizeof(int32_t));
    if (top_labels) {
        if (level_labels.empty()) std::iota(top_labels, top_labels + M, 0);
        else std::memcpy(top_labels, level_labels.back().data(), (size_t)M * sizeof(int32_t));
    }
    return 0;
}

// Distinct-column share of a processing order under an LRU of `window` B rows per slice (the model of
// the per-XCD L2 used in DESIGN.md): returns the fraction of non-zeros whose B row is already resident.
double simulate_l2_hits(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const int32_t* perm,
                        int slices, int64_t window, int64_t max_entries_per_slice) {
    const int64_t nnz = rowptr[M];
    if (nnz == 0 || K <= 0) return 0.0;
    int64_t simulated = 0;
    // exact LRU with a time-stamp array and a ring of (stamp, column) — an entry is live if its stamp is current
    int64_t hits = 0;
    std::vector<int64_t> cuts((size_t)slices + 1, M);
    cuts[0] = 0;
    {
        int64_t acc = 0;
        int s = 1;
        for (int64_t i = 0; i < M && s < slices; ++i) {
            const int32_t r = perm ? perm[i] : (int32_t)i;
            acc += rowptr[r + 1] - rowptr[r];
            while (s < slices && acc >= nnz * s / slices) cuts[s++] = i + 1;
        }
    }
    std::vector<int64_t> stamp((size_t)K);
    std::vector<std::pair<int64_t, int32_t>> ring;
    for (int s = 0; s < slices; ++s) {
        std::fill(stamp.begin(), stamp.end(), -1);
        ring.clear();
        size_t head = 0;
        int64_t live = 0, now = 0;
        for (int64_t i = cuts[s]; i < cuts[s + 1]; ++i) {
            if (max_entries_per_slice > 0 && now >= max_entries_per_slice) break;  // a prefix of the slice is a fair sample
            const int32_t r = perm ? perm[i] : (int32_t)i;
            simulated += rowptr[r + 1] - rowptr[r];
            for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                const int32_t c = colind[p];
                if (c < 0 || c >= K) continue;
                if (stamp[c] >= 0) hits++;
                else live++;
                stamp[c] = now;
                ring.emplace_back(now, c);
                ++now;
                while (live > window) {  // evict the least recently used live entry
                    const auto& e = ring[head++];
                    if (stamp[e.second] == e.first) {
                        stamp[e.second] = -1;
                        live--;
                    }
                }
            }
        }
    }
    return simulated > 0 ? (double)hits / (double)simulated : 0.0;
}

}  // namespace gespmm
