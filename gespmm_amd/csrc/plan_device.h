// plan_device.h — the plan's analysis stage ON THE DEVICE (plan_device.hip): validation, row clustering
// (the algorithm of reorder.cpp, same labels), L2 model, row-permuted copy, task tables. All pointers are
// DEVICE pointers unless named *_host; everything is ordered on `st`; the functions synchronise `st` where
// they need a count on the host (a handful of times per call).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "reorder.h"

namespace gespmm {

// rowptr[0] == 0, rowptr non-decreasing, rowptr[M] == nnz, every colind in [0, K). *bad_host: 0 = fine,
// 1 = rowptr malformed, 2 = a column index out of range. *max_degree_host: longest row. *wedge_probe_host (may be NULL): the structure
// probe of plan_policy.cpp's cost estimate — share of 16 384 sampled wedges (two columns c1, c2 of one row) with c2 in row c1; -1 when the
// matrix is not square or too few wedges exist. One more small kernel, the same readback.
hipError_t device_validate_csr(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                               int32_t* max_degree_host, int32_t* bad_host, double* wedge_probe_host, hipStream_t st);

// Multi-level label propagation of reorder.cpp on the device: perm[i] = original row processed at position i.
// Same rules (snapshot half-sweeps, size caps, hash tie-breaks, twins), hence the same order as cluster_rows().
hipError_t device_cluster_rows(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr, const int32_t* colind,
                               const ClusterOptions& opt, int32_t* perm, ClusterStats* stats_host, hipStream_t st);

// rowptr_p[M+1], colind_p[nnz], src_begin[M] of the matrix with its rows in `perm` order.
hipError_t device_permute_csr(int64_t M, int64_t nnz, const int32_t* rowptr, const int32_t* colind, const int32_t* perm,
                              int32_t* rowptr_p, int32_t* colind_p, int32_t* src_begin, hipStream_t st);

// The L2 model of simulate_l2_hits() (rows of `rowptr`/`colind` in storage order — pass the permuted copy to
// judge an order): `slices` parts of equal non-zero count, an LRU of `window` B rows each; an unbiased estimate
// from `samples_per_slice` stratified accesses per slice whose LRU stack distance is computed exactly
// (all accesses when a slice has fewer). max_entries_per_slice > 0: only the head of every slice is modelled.
hipError_t device_l2_model(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr, const int32_t* colind, int slices,
                           int64_t window, int64_t max_entries_per_slice, int samples_per_slice, double* hits_host,
                           hipStream_t st);

// Greedy task cutting of plan.cpp for BOTH task tables of a plan at once (v = 0: wavefront tasks, 1: lane-group tasks):
// a task = consecutive rows, <= kMaxRowsPerWave of them, cost <= budget[v], at least one row; cost of a row =
// max(entries, row_floor[v]). tasks[v] = int4 {first row, #rows, CSR begin, CSR end} per task; tasks[0] is ONE hipMalloc
// block that holds both tables (free tasks[0] only).
hipError_t device_cut_tasks(int64_t M, const int32_t* rowptr_p, const int64_t budget[2], const int64_t row_floor[2],
                            int32_t* tasks[2], int32_t ntasks_host[2], hipStream_t st);

// Tables of spmm_staged.hip for the clustered matrix (rowptr_p / colind_p / val_p = the plan's row-permuted copy; val_p NULL:
// unweighted, the stream carries 1.0f): blocks of R rows, per block the <= H columns its entries use most often
// (>= 2 uses), `waves` tasks, and the interleaved {code, value} stream. staged_fraction = share of the entries whose B row
// comes from LDS. Deterministic (ties in column order). The four arrays are hipMalloc blocks owned by the caller (free_staging).
struct StagingTables {
    void* block = nullptr;        // the one allocation ev / tasks / hot_cols / nhot are parts of
    int32_t* ev = nullptr;        // 2 * (nnz_s + M + kStagedPad) words: entries + one row-end record per row (spmm_kernels.h)
    int32_t* hot_cols = nullptr;  // nblocks * H
    int32_t* nhot = nullptr;      // nblocks
    int32_t* tasks = nullptr;     // nblocks * waves int4
    int32_t nblocks = 0;
    int32_t waves = 0;            // wavefronts (tasks) per block
    int32_t slots = 0;            // staged rows per block (H)
    double staged_fraction = 0.0;
    // hub rows (device_split_long_rows): the staged kernel walks a copy of the row pointers in which they are EMPTY, and they
    // are handed to the streaming kernel's long-row pass as one-row tasks. NULL / 0: no such rows, the plan's own row pointers.
    int32_t* rowptr_s = nullptr;  // M + 1
    int32_t* ltasks = nullptr;    // nlong int4 {row, 1, CSR begin, CSR end} in the clustered matrix
    int32_t nlong = 0;
    int64_t nnz_s = 0;            // entries the staged kernel walks
};

// Rows of more than `limit` entries taken out of the clustered matrix: rowptr_s (those rows empty; kept in `t`), a compacted
// copy of the remaining column indices / values (*colind_s / *val_s: hipMalloc blocks for the caller to free once the staging
// tables are built; val_s NULL when val_p is) and the one-row tasks of the rows taken out.
hipError_t device_split_long_rows(int64_t M, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p, const float* val_p,
                                  int limit, StagingTables* t, int32_t** colind_s, float** val_s, hipStream_t st);
// perm (clustered position -> C row; NULL = identity) goes into the row-end records of the stream.
// `parts`: tasks per block (the wide kernel: one per wavefront = waves; the narrow kernel: one per lane group = waves x G)
hipError_t device_build_staging(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p,
                                const float* val_p, const int32_t* perm, int R, int H, int waves, int parts, StagingTables* out,
                                hipStream_t st, int64_t seg_rows = 0, bool slab_tables = false);
// (seg_rows > 0: blocks never straddle a segment of seg_rows rows, M a multiple of it; slab_tables: the tables of a column-slab view —
//  task word 0 = C row of the task's first row, row-end codes carry the C row of the next row: spmm_staged.hip's continuing launches)

// Column-slab view of the clustered matrix (plan_device.hip): P x M rows, slab-major. rowptr_v / src_v stay with the plan (value updates),
// colind_v / val_v / perm_v are what device_build_staging reads and go once the tables exist. sorted == 0: some row has descending
// columns — no view was made.
struct SlabView {
    int32_t* rowptr_v = nullptr;  // P * M + 1
    int32_t* src_v = nullptr;     // nnz: entry of the clustered matrix behind each entry of the view
    int32_t* colind_v = nullptr;  // nnz            (temporary)
    float* val_v = nullptr;       // nnz or NULL    (temporary)
    int32_t* perm_v = nullptr;    // P * M          (temporary)
    int32_t slabs = 0;
    int32_t max_row = 0;          // longest row of the view
    int32_t sorted = 0;
};
hipError_t device_build_slab_view(int64_t M, int64_t K, int64_t nnz, const int32_t* rowptr_p, const int32_t* colind_p, const float* val_p,
                                  const int32_t* perm, int P, SlabView* out, hipStream_t st);
hipError_t device_slab_set_values(const StagingTables& t, const SlabView& v, const float* val_p, int64_t Mv, int64_t nnz, hipStream_t st);
void free_slab_view(SlabView* v, bool temporaries_only);
// val_p: values in the clustered matrix's entry order (NULL: 1.0f); rowptr_p: its row pointers (used when hub rows were split off)
hipError_t device_staging_set_values(const StagingTables& t, const float* val_p, const int32_t* rowptr_p, int64_t M, int64_t nnz,
                                     hipStream_t st);
void free_staging(StagingTables* t);

// Frees the analysis arenas kept for the next plan (one per device, up to the limit below).
void release_cached_arena();
// Largest arena kept between plans, per device (default 1 GiB, or GESPMM_ARENA_CACHE_MB; negative: back to that default).
void set_arena_cache_limit(long long bytes);

}  // namespace gespmm
