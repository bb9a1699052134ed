// reorder.h — plan-time row clustering and task tables (host side, no HIP). See reorder.cpp / plan.cpp.
#pragma once
#include <stdint.h>

namespace gespmm {

struct ClusterOptions {
    int threads = 0;       // 0: hardware concurrency (results do not depend on it)
    int max_levels = 0;    // 0: 6
    int sweeps = 0;        // 0: 5 label-propagation sweeps per level
    int64_t first_cap = 0; // 0: 256 original rows per label at level 0
    int cap_growth = 0;    // 0: x4 per level
    int stop_percent = 0;  // 0: 97 — a level that keeps more than this share of its nodes ends the hierarchy
};

struct ClusterStats {
    int levels = 0;
    int32_t clusters[16] = {0};  // row clusters after each level
};

// perm[i] = original row processed at position i. Returns 0, or -1 on bad arguments. All HOST pointers.
int cluster_rows(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const ClusterOptions& opt,
                 int32_t* perm, ClusterStats* stats, int32_t* top_labels = nullptr /* [M]: coarsest cluster of every row (studies) */);

// Model of the per-XCD L2: rows processed in `perm` order (NULL = storage order), cut into `slices`
// contiguous parts of equal non-zero count, each with an LRU of `window` B rows. Returns the share of
// non-zeros whose B row is resident when it is gathered.
// max_entries_per_slice > 0: only that many non-zeros at the head of every slice are simulated (large matrices).
double simulate_l2_hits(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const int32_t* perm,
                        int slices, int64_t window, int64_t max_entries_per_slice = 0);

}  // namespace gespmm
