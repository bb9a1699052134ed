// plan.cpp — gespmm_plan: the "analysis" stage in front of repeated SpMM calls on ONE sparse matrix.
//
// The reference launches its kernels straight on the caller's CSR (spmmWrapper, spmm_test.cu:456-492;
// spmm_cuda, pytorch-custom/spmm_kernel.cu:425-458) and has no such stage; vendor libraries do
// (rocsparse_spmm_stage_preprocess). A plan looks at the matrix ONCE — on the host, one synchronisation —
// and keeps what every later launch can reuse:
//
//   * the longest row (decides the long-row pass exactly instead of guessing from nnz and the mean degree);
//   * for dense graphs: the workspace with the per-row split points of the cache-blocked path;
//   * for sparse graphs whose B exceeds the L2s: a ROW-CLUSTERED copy of the matrix (reorder.cpp) and a task
//     table with an equal non-zero budget per wavefront. Rows that share neighbours are processed next to
//     each other, so the B rows they share are gathered from the XCD's L2 instead of crossing the fabric
//     again. Only the processing order changes: every row is still summed by one lane group in its own CSR
//     order and written to its own C row (through perm[]), so the result has the same bits as the plain call.
//
// The plan owns its device memory (permuted rowptr / colind / val, perm, tasks, workspace) and refers to the
// caller's arrays only while it is created (and in gespmm_plan_set_values).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/gespmm.h"
#include "auto_plan.h"
#include "plan.h"
#include "plan_device.h"
#include "plan_policy.h"
#include "reorder.h"
#include "select.h"
#include "spmm_kernels.h"

struct gespmm_plan {
    int64_t M = 0, K = 0, nnz = 0, N = 0;
    int variant = GESPMM_VARIANT_AUTO;
    int launch_flags = 0;
    int device = 0;
    const int32_t* rowptr = nullptr;  // caller's arrays (used only when the plan keeps the storage order)
    const int32_t* colind = nullptr;
    const float* val = nullptr;
    bool valued = false;
    int32_t max_degree = 0;
    bool reordered = false;
    bool identity_order = false;  // reordered, but the plan's copy is in the caller's order (a matrix that arrived clustered: the staged-rows kernel needs the plan's tables)
    void* d_block = nullptr;  // device analysis: perm / rowptr / colind / src_begin (/ val) are parts of this ONE allocation
    bool val_in_block = false;
    int32_t* d_rowptr = nullptr;
    int32_t* d_colind = nullptr;
    float* d_val = nullptr;
    int32_t* d_perm = nullptr;
    int32_t* d_src_begin = nullptr;
    int32_t* d_tasks = nullptr;
    int32_t ntasks = 0;
    int32_t* d_gtasks = nullptr;  // lane-group tasks of the segmented-stream kernel
    int32_t ngtasks = 0;
    bool gtasks_shared = false;   // d_gtasks points into the block of d_tasks (device analysis)
    // SDDMM through the plan (built on first use): edges in clustered order as COO with the ORIGINAL row ids, the
    // position of every edge in the caller's CSR, and a buffer for the results in clustered order
    int32_t* d_coo_row = nullptr;
    int32_t* d_coo_row_storage = nullptr;  // storage-order plans: row id of every edge (the COO form skips the row search)
    int32_t* d_edge_dst = nullptr;
    float* d_sddmm_tmp = nullptr;
    int32_t task_entries = 0;
    int kernel_choice = 0;         // GESPMM_PLAN_KERNEL_*
    gespmm::PlanFacts facts;       // what the policy functions (plan_policy.h) are asked with
    std::vector<int32_t> perm_host;  // filled by the host analysis, or on demand (gespmm_plan_get_order)
    bool cost_skipped = false;       // AUTO skipped the analysis: expected launches x estimated gain < estimated cost (plan_policy.cpp)
    double est_gain_us = 0.0, est_cost_us = 0.0;
    int analysis = 0;                // GESPMM_PLAN_ANALYSIS_*
    double model_seconds = 0.0;
    void* ws = nullptr;
    int64_t ws_bytes = 0;
    bool split_ready = false;
    int split_vec = 0;               // vector width (operand alignment) the kept split points were computed for
    gespmm::ClusterStats stats;
    double analysis_seconds = 0.0, cluster_seconds = 0.0;
    double hits_before = -1.0, hits_after = -1.0;
    // staged-rows kernel (spmm_staged.hip): tables for width N (plan_device.hip: device_build_staging)
    gespmm::StagingTables stg;
    double staging_seconds = 0.0;
    // column-slab tables (round 6; dense clustered matrices at N = 128): staged tables of the slab view, one launch per slab (plan_run)
    gespmm::StagingTables slab;
    gespmm::SlabView slab_view;
    double slab_seconds = 0.0;
    // padded-record kernel (spmm_records.hip): tables for width N (narrow widths, short rows)
    gespmm::RecordTables rec;
    double records_seconds = 0.0;
    // gespmm_plan_tune: measured kernel times on the caller's operands (us; < 0: candidate not available)
    // The measurement is valid for the plan's own width only: launches at p->N take tuned_kernel / tuned_vec, every other width keeps
    // the per-launch rules of plan_policy.cpp (kernel_choice stays what the creator asked for).
    bool tuned = false;
    int tuned_kernel = 0;  // GESPMM_PLAN_KERNEL_* that won (valid while `tuned`)
    int tuned_vec = 0;     // 1: the winner is the batch-stream kernel with 4 floats per lane (N <= 64)
    bool staging_kept_by_policy = false;  // keep_staged_tables() said yes at creation (else the tables exist only while tune measures them / if they won)
    double tune_us[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};  // batch-stream, segmented-stream, staged-rows, batch-stream with 4 floats per lane (N <= 64), padded records
    bool records_kept_by_policy = false;  // want_record_tables() / keep_record_tables() said yes at creation (else the tables exist only while tune measures them / if they won)
};

namespace {

__global__ void iota_kernel(int32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ void permute_values_kernel(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ src_begin,
                                      const float* __restrict__ val, float* __restrict__ val_p, int M, int nnz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int lo = 0, hi = M;  // rowptr_p[lo] <= p < rowptr_p[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr_p[mid] <= p) lo = mid;
        else hi = mid;
    }
    val_p[p] = val[src_begin[lo] + (p - rowptr_p[lo])];
}

__global__ void plan_edge_maps_kernel(const int32_t* __restrict__ rowptr_p, const int32_t* __restrict__ src_begin,
                                      const int32_t* __restrict__ perm, int32_t* __restrict__ coo_row,
                                      int32_t* __restrict__ edge_dst, int M, int nnz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int lo = 0, hi = M;  // rowptr_p[lo] <= p < rowptr_p[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr_p[mid] <= p) lo = mid;
        else hi = mid;
    }
    coo_row[p] = perm[lo];
    edge_dst[p] = src_begin[lo] + (p - rowptr_p[lo]);
}

__global__ void expand_rows_kernel(const int32_t* __restrict__ rowptr, int32_t* __restrict__ coo_row, int M, int nnz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int lo = 0, hi = M;  // rowptr[lo] <= p < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= p) lo = mid;
        else hi = mid;
    }
    coo_row[p] = lo;
}

__global__ void scatter_by_index_kernel(const float* __restrict__ src, const int32_t* __restrict__ dst_index,
                                        float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[dst_index[i]] = src[i];
}

void free_device(gespmm_plan* p) {
    gespmm::free_staging(&p->stg);
    gespmm::free_staging(&p->slab);
    gespmm::free_slab_view(&p->slab_view, false);
    gespmm::free_records(&p->rec);
    if (p->gtasks_shared) p->d_gtasks = nullptr;
    if (p->d_block) {  // the permuted copy is one block
        (void)hipFree(p->d_block);
        p->d_block = nullptr;
        p->d_rowptr = p->d_colind = p->d_perm = p->d_src_begin = nullptr;
        if (p->val_in_block) p->d_val = nullptr;
        p->val_in_block = false;
    }
    void* ptrs[] = {p->d_rowptr, p->d_colind, p->d_val, p->d_perm, p->d_src_begin, p->d_tasks, p->ws, p->d_gtasks, p->d_coo_row, p->d_edge_dst, p->d_sddmm_tmp, p->d_coo_row_storage};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    p->d_rowptr = p->d_colind = p->d_perm = p->d_src_begin = p->d_tasks = p->d_gtasks = p->d_coo_row = p->d_edge_dst = nullptr;
    p->d_sddmm_tmp = nullptr;
    p->d_coo_row_storage = nullptr;
    p->d_val = nullptr;
    p->ws = nullptr;
}


// Experiment knobs (scripts/plan_time.py): GESPMM_CLUSTER_LEVELS / _SWEEPS / _STOP / _CAP override the clustering defaults.
gespmm::ClusterOptions cluster_options_from_env() {
    gespmm::ClusterOptions o;
    if (const char* v = getenv("GESPMM_CLUSTER_LEVELS")) o.max_levels = atoi(v);
    if (const char* v = getenv("GESPMM_CLUSTER_SWEEPS")) o.sweeps = atoi(v);
    if (const char* v = getenv("GESPMM_CLUSTER_STOP")) o.stop_percent = atoi(v);
    if (const char* v = getenv("GESPMM_CLUSTER_CAP")) o.first_cap = atoi(v);
    return o;
}

template <typename T>
hipError_t upload(T** dst, const std::vector<T>& src, hipStream_t st) {
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, st);
    return e;
}

}  // namespace

namespace gespmm {
bool plan_is_clustered(const gespmm_plan* p) { return p && p->reordered; }
static std::atomic<bool> g_analysis_warm{false};
bool analysis_is_warm() { return g_analysis_warm.load(std::memory_order_relaxed); }
void mark_analysis_warm() { g_analysis_warm.store(true, std::memory_order_relaxed); }
}  // namespace gespmm

extern "C" {

int gespmm_cluster_rows(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int32_t threads,
                        int32_t* perm_out, int32_t* levels_out, int32_t* clusters_out /* [16] */) {
    if (M < 0 || K < 0 || (M > 0 && (!rowptr || !perm_out))) return GESPMM_EINVAL;
    gespmm::ClusterOptions opt;
    opt.threads = threads;
    gespmm::ClusterStats st;
    try {
        if (gespmm::cluster_rows(M, K, rowptr, colind, opt, perm_out, &st) != 0) return GESPMM_EINVAL;
    } catch (const std::bad_alloc&) {
        return GESPMM_ENOMEM;
    }
    if (levels_out) *levels_out = st.levels;
    if (clusters_out)
        for (int i = 0; i < 16; ++i) clusters_out[i] = st.clusters[i];
    return 0;
}

// Study hook (scripts/cluster_chain_study.py): the host clustering with its options and the coarsest cluster of every row.
int gespmm_cluster_rows_study(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int32_t max_levels, int32_t sweeps,
                              int32_t* perm_out, int32_t* top_label_out, int32_t* clusters_out /* [16] */) {
    if (M <= 0 || !rowptr || !perm_out) return GESPMM_EINVAL;
    gespmm::ClusterOptions opt = cluster_options_from_env();
    opt.max_levels = max_levels;
    opt.sweeps = sweeps;
    gespmm::ClusterStats st;
    if (gespmm::cluster_rows(M, K, rowptr, colind, opt, perm_out, &st, top_label_out) != 0) return GESPMM_EINVAL;
    if (clusters_out)
        for (int i = 0; i < 16; ++i) clusters_out[i] = st.clusters[i];
    return st.levels;
}

double gespmm_simulate_l2_hits(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, const int32_t* perm,
                               int32_t slices, int64_t window_rows) {
    if (M <= 0 || K <= 0 || !rowptr || slices < 1 || window_rows < 1) return 0.0;
    try {
        return gespmm::simulate_l2_hits(M, K, rowptr, colind, perm, slices, window_rows);
    } catch (const std::bad_alloc&) {
        return -1.0;
    }
}

// The device analysis by itself (DEVICE rowptr / colind, HOST outputs) — what tests compare with gespmm_cluster_rows.
int gespmm_device_cluster_rows(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                               int32_t* perm_out_host, int32_t* levels_out, int32_t* clusters_out /* [16] */, void* stream) {
    if (M < 0 || K < 0 || nnz < 0 || (M > 0 && (!rowptr || !perm_out_host)) || (nnz > 0 && !colind)) return GESPMM_EINVAL;
    if (M == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t max_deg = 0, bad = 0;
    hipError_t e = gespmm::device_validate_csr(rowptr, colind, M, K, nnz, &max_deg, &bad, nullptr, st);
    if (e != hipSuccess) return (int)e;
    if (bad) return GESPMM_EINVAL;
    int32_t* d_perm = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&d_perm), (size_t)M * 4);
    if (e != hipSuccess) return (int)e;
    gespmm::ClusterOptions opt;
    gespmm::ClusterStats stats;
    e = gespmm::device_cluster_rows(M, K, nnz, rowptr, colind, opt, d_perm, &stats, st);
    if (e == hipSuccess) e = hipMemcpyAsync(perm_out_host, d_perm, (size_t)M * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_perm);
    if (e != hipSuccess) return (int)e;
    if (levels_out) *levels_out = stats.levels;
    if (clusters_out)
        for (int i = 0; i < 16; ++i) clusters_out[i] = stats.clusters[i];
    return 0;
}

// The device L2 model by itself: DEVICE rowptr / colind, perm_host (HOST, may be NULL = storage order). Returns the
// modelled hit rate, or a negative value on error.
double gespmm_device_l2_model(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                              const int32_t* perm_host, int32_t slices, int64_t window_rows, int64_t max_entries_per_slice,
                              int32_t samples_per_slice, void* stream) {
    if (M <= 0 || K <= 0 || nnz <= 0 || !rowptr || !colind || slices < 1 || slices > 16 || window_rows < 1) return -1.0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double hits = -1.0;
    hipError_t e = hipSuccess;
    int32_t *d_perm = nullptr, *rp = nullptr, *ci = nullptr, *src = nullptr;
    if (perm_host) {
        e = hipMalloc(reinterpret_cast<void**>(&d_perm), (size_t)M * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&rp), ((size_t)M + 1) * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ci), (size_t)nnz * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&src), (size_t)M * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_perm, perm_host, (size_t)M * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = gespmm::device_permute_csr(M, nnz, rowptr, colind, d_perm, rp, ci, src, st);
    }
    if (e == hipSuccess)
        e = gespmm::device_l2_model(M, K, nnz, perm_host ? rp : rowptr, perm_host ? ci : colind, slices, window_rows,
                                    max_entries_per_slice, samples_per_slice > 0 ? samples_per_slice : 8192, &hits, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    for (void* q : {(void*)d_perm, (void*)rp, (void*)ci, (void*)src})
        if (q) (void)hipFree(q);
    return e == hipSuccess ? hits : -1.0;
}

// Test hook: the task table of a clustered plan (which = 0: wavefront tasks, 1: lane-group tasks) as int4 records in
// HOST memory; returns the number of tasks (<= capacity are copied) or a negative error.
int gespmm_plan_debug_tasks(const gespmm_plan* p, int32_t which, int32_t* out_host, int64_t capacity) {
    if (!p || which < 0 || which > 1) return GESPMM_EINVAL;
    if (!p->reordered) return 0;
    const int32_t n = which ? p->ngtasks : p->ntasks;
    const int32_t* d = which ? p->d_gtasks : p->d_tasks;
    const int64_t take = n < capacity ? n : capacity;
    if (take > 0 && out_host && hipMemcpy(out_host, d, (size_t)take * 16, hipMemcpyDeviceToHost) != hipSuccess) return GESPMM_EINVAL;
    return n;
}

// Tables of the staged-rows kernel for the plan's width (plan_device.hip). Hub rows (one wavefront would walk such a row
// alone) are taken out: the staged kernel sees them empty, the streaming kernel's long-row pass gets them as one-row tasks
// (plan_run). Leaves p->stg empty when there is nothing but hub rows.
static hipError_t build_staging_tables(gespmm_plan* p, hipStream_t st) {
    const auto ts = std::chrono::steady_clock::now();
    const int64_t M = p->M, K = p->K, nnz = p->nnz, N = p->N;
    gespmm::StagedShape shape = gespmm::staged_shape_any(N);  // (the block shape of whichever staged kernel serves the width)
    if (!getenv("GESPMM_STAGED_ROWS")) shape.rows = gespmm::staged_rows_for(p->facts, shape.rows, shape.waves);  // (by mean degree: plan_policy.cpp)
    hipError_t e = hipSuccess;
    const int32_t* rp_s = p->d_rowptr;
    const int32_t* ci_s = p->d_colind;
    const float* val_s = p->valued ? p->d_val : nullptr;
    int32_t* ci_tmp = nullptr;
    float* val_tmp = nullptr;
    int64_t nnz_s = nnz;
    if (p->max_degree > gespmm::kStagedMaxRow) {
        e = gespmm::device_split_long_rows(M, nnz, p->d_rowptr, p->d_colind, val_s, gespmm::kStagedMaxRow, &p->stg, &ci_tmp, &val_tmp, st);
        rp_s = p->stg.rowptr_s;
        ci_s = ci_tmp;
        val_s = val_tmp;
        nnz_s = p->stg.nnz_s;
    }
    if (e == hipSuccess && nnz_s > 0)
        e = gespmm::device_build_staging(M, K, nnz_s, rp_s, ci_s, val_s, p->d_perm, shape.rows, shape.slots, shape.waves,
                                         gespmm::staged_tasks_per_block(N), &p->stg, st);
    if (ci_tmp) (void)hipFree(ci_tmp);
    if (val_tmp) (void)hipFree(val_tmp);
    if (e == hipSuccess && !p->stg.ev) gespmm::free_staging(&p->stg);  // (nothing but hub rows)
    p->staging_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
    return e;
}

// Column-slab tables (plan_device.hip: device_build_slab_view): the clustered matrix as P ascending column ranges, staged tables per
// (block of rows, slab). Leaves p->slab empty when the matrix does not qualify (a row with descending columns: slab order would not be
// CSR order; a slab of some row beyond the staged kernel's row limit) — the plan's other kernels stay.
static hipError_t build_slab_tables(gespmm_plan* p, int P, hipStream_t st) {
    const auto ts = std::chrono::steady_clock::now();
    const int64_t M = p->M, K = p->K, nnz = p->nnz, N = p->N;
    gespmm::StagedShape shape = gespmm::staged_shape(N);
    static const int rows_env = getenv("GESPMM_SLAB_ROWS") ? atoi(getenv("GESPMM_SLAB_ROWS")) : 0;  // experiments
    if (rows_env > 0) shape.rows = rows_env;
    static const int lds_env = getenv("GESPMM_SLAB_LDS_KB") ? atoi(getenv("GESPMM_SLAB_LDS_KB")) : 0;  // experiments: 3 = three 48 KB blocks per CU
    if (N == 128 && shape.waves == 16 && shape.slots == 160 && lds_env == 3) shape.slots = 96;
    if (N != 128 || shape.waves != 16 || (shape.slots != 160 && shape.slots != 96) || P < 2 || nnz <= 0 || !gespmm::staged_serves(M, K, N) ||
        !gespmm::staged_stream_fits(M * P, nnz) || M * P >= (1ll << 30))
        return hipSuccess;
    hipError_t e = gespmm::device_build_slab_view(M, K, nnz, p->d_rowptr, p->d_colind, p->valued ? p->d_val : nullptr, p->d_perm, P,
                                                  &p->slab_view, st);
    if (e == hipSuccess && (!p->slab_view.sorted || p->slab_view.max_row > gespmm::kStagedMaxRow)) {
        gespmm::free_slab_view(&p->slab_view, false);
        return hipSuccess;
    }
    if (e == hipSuccess)
        e = gespmm::device_build_staging(M * P, K, nnz, p->slab_view.rowptr_v, p->slab_view.colind_v, p->slab_view.val_v, p->slab_view.perm_v,
                                         shape.rows, shape.slots, shape.waves, shape.waves, &p->slab, st, M, true);
    gespmm::free_slab_view(&p->slab_view, e == hipSuccess);  // (the view's column indices / values / row map are in the tables now)
    if (e != hipSuccess) gespmm::free_staging(&p->slab);
    p->slab_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
    return e;
}

static double record_slot_fill(const gespmm_plan* p) {  // share of the entry slots of the batches that carry an entry
    if (!p->rec.batches || p->rec.nbatches <= 0) return 0.0;
    return (double)p->nnz / ((double)p->rec.nbatches * (64 / p->rec.group) * gespmm::kRecordPiece);
}

// Tables of the padded-record kernel (spmm_records.hip) for the plan's width: the matrix in the order the plan processes it (its
// clustered copy, or the caller's arrays when the storage order was kept).
static hipError_t build_record_tables(gespmm_plan* p, hipStream_t st) {
    const auto ts = std::chrono::steady_clock::now();
    static const int env_rows = getenv("GESPMM_REC_BATCHES") ? atoi(getenv("GESPMM_REC_BATCHES")) : 0;  // experiments
    const int rows = env_rows > 0 ? env_rows : gespmm::records_batches_per_task(p->facts);
    const hipError_t e = gespmm::device_build_records(p->M, p->reordered ? p->d_rowptr : p->rowptr, p->reordered ? p->d_colind : p->colind,
                                                      p->valued ? (p->reordered ? p->d_val : p->val) : nullptr,
                                                      p->reordered ? p->d_perm : nullptr, rows, p->N, &p->rec, st);
    p->records_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
    return e;
}

// gespmm_plan_create_v2: `opt_bytes` = sizeof(gespmm_plan_options) as the CALLER was compiled with; fields beyond it take
// their defaults, bytes beyond what this library knows are ignored (gespmm.h, "Plan options and versions").
static int plan_create_impl(gespmm_plan** out, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M,
                            int64_t K, int64_t nnz, int64_t N, int variant, const gespmm_plan_options* opt, void* stream) {
    if (!out) return GESPMM_EINVAL;
    *out = nullptr;
    if (M < 0 || K < 0 || N < 0 || nnz < 0) return GESPMM_EINVAL;
    if (M > 0x7fffffffLL - 64 || K > 0x7fffffffLL || N > 0x7fffffffLL / 4 || nnz > 0x7fffffffLL - 4096) return GESPMM_ERANGE;
    if (variant < GESPMM_VARIANT_AUTO || variant >= GESPMM_NUM_VARIANTS) return GESPMM_EINVAL;
    if (M > 0 && !rowptr) return GESPMM_EINVAL;
    if (nnz > 0 && !colind) return GESPMM_EINVAL;
    const int reorder_mode = opt ? opt->reorder : GESPMM_PLAN_REORDER_AUTO;
    if (reorder_mode < 0 || reorder_mode > 2) return GESPMM_EINVAL;
    const int kernel_mode = opt ? opt->kernel : GESPMM_PLAN_KERNEL_AUTO;
    if (kernel_mode != GESPMM_PLAN_KERNEL_AUTO && kernel_mode != GESPMM_PLAN_KERNEL_STREAM && kernel_mode != GESPMM_PLAN_KERNEL_SEG_STREAM &&
        kernel_mode != GESPMM_PLAN_KERNEL_STAGED && kernel_mode != GESPMM_PLAN_KERNEL_RECORDS && kernel_mode != GESPMM_PLAN_KERNEL_STAGED_SLABS)
        return GESPMM_EINVAL;
    if (opt && opt->expected_launches < 0) return GESPMM_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const auto t_start = std::chrono::steady_clock::now();

    gespmm_plan* p = new (std::nothrow) gespmm_plan;
    if (!p) return GESPMM_ENOMEM;
    p->M = M;
    p->K = K;
    p->nnz = nnz;
    p->N = N;
    p->variant = variant;
    p->rowptr = rowptr;
    p->colind = colind;
    p->val = val;
    p->valued = val != nullptr;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) {
        delete p;
        return (int)e;
    }
    const int user_flags = opt ? opt->flags : 0;

    const int analysis = opt ? opt->analysis : GESPMM_PLAN_ANALYSIS_DEVICE;
    if (analysis != GESPMM_PLAN_ANALYSIS_DEVICE && analysis != GESPMM_PLAN_ANALYSIS_HOST) {
        delete p;
        return GESPMM_EINVAL;
    }
    p->analysis = analysis;
    const bool on_host = analysis == GESPMM_PLAN_ANALYSIS_HOST;

    try {
        // ---- one pass over the matrix on the device: rowptr monotone and consistent with nnz, every column index
        //      inside [0, K) (the kernels trust them), the longest row
        int32_t max_deg = 0, bad = 0;
        double wedge_probe = -1.0;
        const int reorder_auto = reorder_mode == GESPMM_PLAN_REORDER_AUTO;
        e = gespmm::device_validate_csr(rowptr, colind, M, K, nnz, &max_deg, &bad, (reorder_auto && !on_host) ? &wedge_probe : nullptr, st);
        if (e != hipSuccess) {
            delete p;
            return (int)e;
        }
        if (bad) {
            delete p;
            return GESPMM_EINVAL;  // rowptr does not describe nnz entries, or a column index is outside [0, K)
        }
        // ---- host analysis only (GESPMM_PLAN_ANALYSIS_HOST): the matrix comes to the host once
        std::vector<int32_t> h_rowptr, h_colind;
        if (on_host) {
            h_rowptr.assign((size_t)M + 1, 0);
            h_colind.resize((size_t)nnz);
            if (M > 0) e = hipMemcpyAsync(h_rowptr.data(), rowptr, ((size_t)M + 1) * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(h_colind.data(), colind, (size_t)nnz * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                delete p;
                return (int)e;
            }
        }
        p->max_degree = max_deg;
        // ---- the facts the policy is asked with (plan_policy.h): what a plain call would launch, the longest row, the options
        gespmm::PlanFacts& f = p->facts;
        f.M = M;
        f.K = K;
        f.nnz = nnz;
        f.N = N;
        f.variant = variant;
        f.max_degree = max_deg;
        f.reorder_mode = reorder_mode;
        f.kernel_choice = p->kernel_choice = opt ? opt->kernel : GESPMM_PLAN_KERNEL_AUTO;
        f.host_analysis = on_host;
        f.user_flags = user_flags;
        f.opt_task_entries = opt ? opt->task_entries : 0;
        f.opt_row_floor = opt ? opt->row_floor : 0;
        f.expected_launches = opt ? opt->expected_launches : 0;
        f.wedge_probe = wedge_probe;
        f.cold_start = !gespmm::analysis_is_warm();
        {
            gespmm::Selection sel;
            int max_vec = 4;
            while (max_vec > 1 && (N % max_vec) != 0) max_vec >>= 1;
            const int lr_flags = gespmm::long_row_flags(M, nnz, max_deg, user_flags);
            if (gespmm::resolve_geometry(M, K, N > 0 ? N : 1, nnz, variant, max_vec, 0, 0, 0, 0, 0, lr_flags, &sel) != 0) {
                delete p;
                return GESPMM_EINVAL;
            }
            f.sel_variant = sel.variant;
            f.slab_blocked = sel.geo.slab_blocked;
            f.tile_cols = (int64_t)sel.geo.group * sel.geo.vec * sel.geo.strips;
        }
        const gespmm::AnalysisDecision ad = gespmm::decide_analysis(f);
        p->launch_flags = ad.launch_flags;
        bool reorder = ad.analyse;
        p->cost_skipped = ad.cost_skipped;
        p->est_gain_us = ad.cost.gain_us;
        p->est_cost_us = ad.cost.cost_us;
        const bool dense_try = ad.dense_try;
        const int64_t model_window = ad.model_window, model_sample = ad.model_sample;

        static const bool timing = getenv("GESPMM_PLAN_TIMING") != nullptr;
        auto lap = [&](const char* what) {
            if (!timing) return;
            (void)hipStreamSynchronize(st);
            static thread_local std::chrono::steady_clock::time_point last;
            const auto now = std::chrono::steady_clock::now();
            if (what) fprintf(stderr, "[plan] %-22s %8.3f ms\n", what, std::chrono::duration<double>(now - last).count() * 1e3);
            last = now;
        };
        if (reorder && !on_host) {
            // ==================================================================== analysis on the device
            lap(nullptr);
            const auto tc = std::chrono::steady_clock::now();
            {   // one allocation for the plan's permuted copy of the matrix (a hipMalloc costs ~0.1 ms: five of them were 7 % of the analysis)
                auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
                const size_t b_perm = up((size_t)M * 4), b_rp = up(((size_t)M + 1) * 4), b_ci = up((size_t)(nnz > 0 ? nnz : 1) * 4),
                             b_src = up((size_t)M * 4), b_val = p->valued ? up((size_t)(nnz > 0 ? nnz : 1) * 4) : 0;
                e = hipMalloc(&p->d_block, b_perm + b_rp + b_ci + b_src + b_val);
                if (e == hipSuccess) {
                    char* base = reinterpret_cast<char*>(p->d_block);
                    p->d_perm = reinterpret_cast<int32_t*>(base);
                    p->d_rowptr = reinterpret_cast<int32_t*>(base + b_perm);
                    p->d_colind = reinterpret_cast<int32_t*>(base + b_perm + b_rp);
                    p->d_src_begin = reinterpret_cast<int32_t*>(base + b_perm + b_rp + b_ci);
                    if (p->valued) {
                        p->d_val = reinterpret_cast<float*>(base + b_perm + b_rp + b_ci + b_src);
                        p->val_in_block = true;
                    }
                }
            }
            gespmm::ClusterOptions copt = cluster_options_from_env();
            if (copt.max_levels <= 0) copt.max_levels = gespmm::cluster_levels_for(f);
            if (copt.sweeps <= 0) copt.sweeps = gespmm::cluster_sweeps_for(f);
            if (e == hipSuccess) e = gespmm::device_cluster_rows(M, K, nnz, rowptr, colind, copt, p->d_perm, &p->stats, st);
            p->cluster_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
            lap("cluster");
            if (e == hipSuccess)
                e = gespmm::device_permute_csr(M, nnz, rowptr, colind, p->d_perm, p->d_rowptr, p->d_colind, p->d_src_begin, st);
            lap("permute");
            const auto tm = std::chrono::steady_clock::now();
            // (the storage order is only judged against the clustered one — "already local, or hit by hubs: keep it" shows anywhere in a
            //  slice — so on matrices of >= 2^20 entries the first QUARTER of every slice is modelled: a quarter of the sort)
            const int64_t before_sample = nnz >= (1 << 20) ? std::max<int64_t>(nnz / 32, 1 << 15) : model_sample;
            const int model_points = gespmm::model_points_for(f);  // sampled accesses per slice
            if (e == hipSuccess)
                e = gespmm::device_l2_model(M, K, nnz, rowptr, colind, 8, model_window,
                                            model_sample > 0 ? std::min<int64_t>(model_sample, before_sample) : before_sample, model_points,
                                            &p->hits_before, st);
            if (e == hipSuccess)
                e = gespmm::device_l2_model(M, K, nnz, p->d_rowptr, p->d_colind, 8, model_window, model_sample, model_points,
                                            &p->hits_after, st);
            p->model_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tm).count();
            lap("l2 model x2");
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            if (!gespmm::keep_clustered_order(f, ad, p->hits_before, p->hits_after) && !dense_try && gespmm::storage_order_wants_plan_copy(f, p->hits_before)) {
                // The matrix ARRIVED in an order as good as the clustering's (a caller who keeps the graph by community): the staged-rows
                // kernel still needs the plan's own tables, so the plan copies the matrix in the IDENTITY order and goes on as if it had
                // clustered it (dropped again below if the tables are not kept: then nothing is paid per launch, as before)
                hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, p->d_perm, (int)M);
                e = hipGetLastError();
                if (e == hipSuccess)
                    e = gespmm::device_permute_csr(M, nnz, rowptr, colind, p->d_perm, p->d_rowptr, p->d_colind, p->d_src_begin, st);
                if (e != hipSuccess) {
                    free_device(p);
                    delete p;
                    return (int)e;
                }
                p->hits_after = p->hits_before;
                p->identity_order = true;
            } else if (!gespmm::keep_clustered_order(f, ad, p->hits_before, p->hits_after)) {
                reorder = false;  // the storage order (or the cache-blocked path) is as good: keep it and pay nothing per launch
                (void)hipFree(p->d_block);
                p->d_block = nullptr;
                p->d_perm = p->d_rowptr = p->d_colind = p->d_src_begin = nullptr;
                p->d_val = nullptr;
                p->val_in_block = false;
            }
        }
        if (reorder && !on_host) {
            if (dense_try) p->launch_flags |= GESPMM_FLAG_NO_SLAB_BLOCKED;  // a clustered dense graph runs the streaming kernels
            // ---- task tables (same greedy cut as the host path), values
            const gespmm::PlanKernelDecision kd = gespmm::choose_plan_kernel(f, p->hits_after);
            const int budget = kd.task_entries, gbudget = kd.group_task_entries;
            const int64_t row_floor = kd.row_floor;
            p->task_entries = budget;
            {
                const int64_t budgets[2] = {budget, gbudget}, floors[2] = {row_floor, 0};
                int32_t* tables[2] = {nullptr, nullptr};
                int32_t counts[2] = {0, 0};
                e = gespmm::device_cut_tasks(M, p->d_rowptr, budgets, floors, tables, counts, st);
                p->d_tasks = tables[0];
                p->d_gtasks = tables[1];
                p->ntasks = counts[0];
                p->ngtasks = counts[1];
                p->gtasks_shared = true;  // one block holds both tables: free d_tasks only
            }
            lap("tasks x2");
            if (e == hipSuccess && p->valued && nnz > 0) {  // (d_val is part of the plan's block)
                hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p->d_rowptr,
                                   p->d_src_begin, val, p->d_val, (int)M, (int)nnz);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the caller's `val` is not read after the call returns)
            lap("values");
            // ---- staged-rows kernel (choose_plan_kernel says when): the tables are built, and kept when enough entries find
            //      their B row staged
            if (e == hipSuccess && kd.build_staged) {
                e = build_staging_tables(p, st);
                if (e == hipSuccess && p->stg.ev && !gespmm::keep_staged_tables(f, p->stg.staged_fraction))
                    gespmm::free_staging(&p->stg);  // not enough reuse inside the blocks: the streaming kernels stay
                p->staging_kept_by_policy = p->stg.ev != nullptr;
                lap("staging tables");
            }
            // ---- column-slab tables (dense clustered matrices at N = 128: plan_policy.cpp slab_count_for / keep_slab_tables)
            if (e == hipSuccess && !p->identity_order) {
                const int P = gespmm::slab_count_for(f);
                if (P >= 2) {
                    e = build_slab_tables(p, P, st);
                    if (e == hipSuccess && p->slab.ev && !gespmm::keep_slab_tables(f, p->slab.staged_fraction)) {
                        gespmm::free_staging(&p->slab);
                        gespmm::free_slab_view(&p->slab_view, false);
                    }
                    lap("slab tables");
                }
            }
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            if (p->identity_order && !p->staging_kept_by_policy) {
                // the copy in storage order was made for the staged-rows kernel alone: without its tables the caller's arrays serve
                gespmm::free_staging(&p->stg);
                if (p->d_tasks) (void)hipFree(p->d_tasks);
                p->d_tasks = p->d_gtasks = nullptr;
                p->ntasks = p->ngtasks = 0;
                p->gtasks_shared = false;
                (void)hipFree(p->d_block);
                p->d_block = nullptr;
                p->d_perm = p->d_rowptr = p->d_colind = p->d_src_begin = nullptr;
                p->d_val = nullptr;
                p->val_in_block = false;
                p->identity_order = false;
            } else {
                p->reordered = true;
                if (kd.shallow_unroll) p->launch_flags |= GESPMM_FLAG_SHALLOW_UNROLL;
            }
            reorder = false;  // done: skip the host branch
        }

        if (reorder) {
            const auto tc = std::chrono::steady_clock::now();
            p->perm_host.resize((size_t)M);
            gespmm::ClusterOptions copt = cluster_options_from_env();
            if (copt.max_levels <= 0) copt.max_levels = gespmm::cluster_levels_for(f);
            if (copt.sweeps <= 0) copt.sweeps = gespmm::cluster_sweeps_for(f);
            copt.threads = opt ? opt->threads : 0;
            if (gespmm::cluster_rows(M, K, h_rowptr.data(), h_colind.data(), copt, p->perm_host.data(), &p->stats) != 0) {
                delete p;
                return GESPMM_EINVAL;
            }
            // (Moving the heavy rows to the front of each XCD slice, so that no long sequential chain starts late, was
            // measured: no effect on the community graph, 151 vs 137 us on the structureless one — hubs stay where the
            // clustering puts them, next to the rows that share their neighbours. profiles/r02/plan_hubs_first.log)
            p->cluster_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
            // A model of the XCD L2s says whether the new order is worth having (graphs whose storage order is
            // already local, or that have no structure to find, keep their order and pay nothing per launch).
            {
                p->hits_before = gespmm::simulate_l2_hits(M, K, h_rowptr.data(), h_colind.data(), nullptr, 8, model_window, model_sample);
                p->hits_after = gespmm::simulate_l2_hits(M, K, h_rowptr.data(), h_colind.data(), p->perm_host.data(), 8, model_window, model_sample);
                if (!gespmm::keep_clustered_order(f, ad, p->hits_before, p->hits_after)) reorder = false;
            }
        }
        if (reorder) {
            // ---- row-permuted copy + task table
            std::vector<int32_t> rp((size_t)M + 1), ci((size_t)nnz), src((size_t)M);
            rp[0] = 0;
            for (int64_t i = 0; i < M; ++i) {
                const int32_t r = p->perm_host[i];
                const int32_t b = h_rowptr[r], d = h_rowptr[r + 1] - b;
                src[i] = b;
                std::memcpy(ci.data() + rp[i], h_colind.data() + b, (size_t)d * 4);
                rp[i + 1] = rp[i] + d;
            }
            const gespmm::PlanKernelDecision kd = gespmm::choose_plan_kernel(f, p->hits_after);
            const int budget = kd.task_entries, gbudget = kd.group_task_entries;
            const int64_t row_floor = kd.row_floor;
            p->task_entries = budget;
            auto cost = [&](int64_t i2) { const int64_t d = rp[i2 + 1] - rp[i2]; return d > row_floor ? d : row_floor; };
            // batch-stream kernel: a task per WAVEFRONT; segmented-stream kernel: a task per lane GROUP (its time is
            // proportional to the entries it streams, so its tasks are cut by non-zeros alone, half the budget)
            auto cut_tasks = [&](int64_t budget_, bool floor_rows, std::vector<int32_t>& out_) {
                out_.reserve((size_t)(nnz / (budget_ > 0 ? budget_ : 1) + M / gespmm::kMaxRowsPerWave + 16) * 4);
                int64_t i2 = 0;
                while (i2 < M) {
                    const int64_t first = i2;
                    auto c2 = [&](int64_t r) { return floor_rows ? cost(r) : (int64_t)(rp[r + 1] - rp[r]); };
                    int64_t acc = c2(i2);
                    ++i2;
                    while (i2 < M && i2 - first < gespmm::kMaxRowsPerWave && acc + c2(i2) <= budget_) {
                        acc += c2(i2);
                        ++i2;
                    }
                    out_.push_back((int32_t)first);
                    out_.push_back((int32_t)(i2 - first));
                    out_.push_back(rp[first]);
                    out_.push_back(rp[i2]);
                }
            };
            std::vector<int32_t> tasks, gtasks;
            cut_tasks(budget, true, tasks);
            cut_tasks(gbudget, false, gtasks);
            p->ngtasks = (int32_t)(gtasks.size() / 4);
            p->ntasks = (int32_t)(tasks.size() / 4);
            e = upload(&p->d_rowptr, rp, st);
            if (e == hipSuccess) e = upload(&p->d_colind, ci, st);
            if (e == hipSuccess) e = upload(&p->d_perm, p->perm_host, st);
            if (e == hipSuccess) e = upload(&p->d_src_begin, src, st);
            if (e == hipSuccess) e = upload(&p->d_tasks, tasks, st);
            if (e == hipSuccess) e = upload(&p->d_gtasks, gtasks, st);
            if (e == hipSuccess && p->valued) e = hipMalloc(reinterpret_cast<void**>(&p->d_val), (size_t)(nnz > 0 ? nnz : 1) * 4);
            if (e == hipSuccess && p->valued && nnz > 0) {
                hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p->d_rowptr,
                                   p->d_src_begin, val, p->d_val, (int)M, (int)nnz);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host vectors go out of scope
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            p->reordered = true;
            if (kd.shallow_unroll) p->launch_flags |= GESPMM_FLAG_SHALLOW_UNROLL;
        } else {
            p->perm_host.clear();
        }
        // ---- scratch of the launches (split points / long-row partials), owned by the plan
        gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, p->launch_flags};
        const int64_t need = gespmm_csr_spmm_workspace_bytes(M, K, N, nnz, variant, &cfg);
        if (need > 0) {
            e = hipMalloc(&p->ws, (size_t)need);
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            p->ws_bytes = need;
        }
    } catch (const std::bad_alloc&) {
        free_device(p);
        delete p;
        return GESPMM_ENOMEM;
    }
    // ---- padded-record kernel (narrow widths, short rows): when asked for, or when the policy says so (plan_policy.cpp)
    if (gespmm::records_serves(M, K, N, p->max_degree) && nnz > 0 && gespmm::want_record_tables(p->facts, p->reordered ? p->hits_after : p->hits_before) &&
        !(p->stg.ev && p->staging_kept_by_policy)) {
        e = build_record_tables(p, st);
        if (e == hipErrorOutOfMemory) {  // (padding beyond the cap, or no memory: the other kernels serve the plan)
            gespmm::free_records(&p->rec);
            (void)hipGetLastError();
            e = hipSuccess;
        }
        if (e == hipSuccess && p->rec.batches && !gespmm::keep_record_tables(p->facts, record_slot_fill(p)))
            gespmm::free_records(&p->rec);  // too much padding (rows of very different lengths share tasks): the other kernels stay
        p->records_kept_by_policy = p->rec.batches != nullptr;
        if (e != hipSuccess) {
            free_device(p);
            delete p;
            return (int)e;
        }
    }
    p->analysis_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (p->reordered || p->hits_after >= 0.0) gespmm::mark_analysis_warm();  // (the analysis passes ran: their kernels are loaded now)
    *out = p;
    return 0;
}

// The un-versioned entry point: every header that shipped with it alone had a gespmm_plan_options of SEVEN int32 fields
// (reorder .. analysis — round 3's 0.1 header already carried `analysis`, and this symbol honoured it), so that is what it reads;
// anything appended later is reachable through gespmm_plan_create_v2 only.
int gespmm_plan_create(gespmm_plan** out, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M,
                       int64_t K, int64_t nnz, int64_t N, int variant, const gespmm_plan_options* opt, void* stream) {
    return gespmm_plan_create_v2(out, rowptr, colind, val, M, K, nnz, N, variant, opt, opt ? 7 * (int64_t)sizeof(int32_t) : 0, stream);
}

int gespmm_plan_create_v2(gespmm_plan** out, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M,
                          int64_t K, int64_t nnz, int64_t N, int variant, const gespmm_plan_options* opt, int64_t opt_bytes,
                          void* stream) {
    if (opt && (opt_bytes < 0 || opt_bytes % 4 != 0)) return GESPMM_EINVAL;
    gespmm_plan_options o;
    std::memset(&o, 0, sizeof o);  // every field's default is 0
    if (opt && opt_bytes > 0) std::memcpy(&o, opt, (size_t)(opt_bytes < (int64_t)sizeof o ? opt_bytes : (int64_t)sizeof o));
    return plan_create_impl(out, rowptr, colind, val, M, K, nnz, N, variant, opt ? &o : nullptr, stream);
}

static int plan_run(gespmm_plan* p, const float* B, float* C, int64_t N, int reduce, float empty, void* stream,
                    const gespmm::LaunchGuard* guard = nullptr) {
    if (!p || N < 0) return GESPMM_EINVAL;
    if (reduce == gespmm::kReduceMax && p->valued) return GESPMM_EINVAL;
    gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, p->launch_flags};
    void* ws = (N == p->N) ? p->ws : nullptr;  // another width: the library's pool serves the scratch
    const int64_t ws_bytes = (N == p->N) ? p->ws_bytes : 0;
    // the slab geometry (rows per slab) depends on the vector width the operands' alignment allows: split points kept from
    // a call with other alignment must not be reused
    int vec_now = 4;
    while (vec_now > 1 && ((N % vec_now) != 0 || (reinterpret_cast<uintptr_t>(B) % (4u * vec_now)) != 0 ||
                           (reinterpret_cast<uintptr_t>(C) % (4u * vec_now)) != 0))
        vec_now >>= 1;
    if (ws && p->split_ready && p->split_vec == vec_now) cfg.flags |= GESPMM_FLAG_REUSE_SPLIT;
    int rc;
    const bool variant_v4 = p->variant == GESPMM_VARIANT_AUTO || p->variant == GESPMM_VARIANT_CRC_CWM4 ||
                            p->variant == GESPMM_VARIANT_CRC_CWM8;
    // which kernel: the creator's choice (AUTO = the rules of plan_policy.cpp, per launch width) — or, at the plan's own width,
    // what gespmm_plan_tune measured
    const bool use_tuned = p->tuned && N == p->N;
    const int kchoice = use_tuned ? p->tuned_kernel : p->kernel_choice;
    gespmm::PlanFacts facts = p->facts;
    facts.kernel_choice = kchoice;
    // staged-rows kernels: the tables exist (the plan decided at creation), same width, 16-byte operands. Which kernel walks them follows
    // from the width and the reducer (spmm_kernels.h: staged_kernel_class): lane groups at N = 16 / 32 / 64, the tuned shapes at N = 128 and
    // 256 * 2^t, the general kernel at every other width and for the max reducer (round 6)
    const int sclass = gespmm::staged_kernel_class(p->M, p->K, N, reduce);
    const bool shape_ok = sclass != gespmm::kStagedGeneral ||
                          (p->stg.waves == gespmm::staged_gen_shape(N).waves && p->stg.slots == gespmm::staged_gen_shape(N).slots);
    const bool staged = p->reordered && p->stg.ev && N == p->N && sclass != gespmm::kStagedNone && shape_ok && variant_v4 &&
                        (use_tuned ? kchoice == GESPMM_PLAN_KERNEL_STAGED
                                   : ((kchoice == GESPMM_PLAN_KERNEL_AUTO && p->staging_kept_by_policy) || kchoice == GESPMM_PLAN_KERNEL_STAGED)) &&
                        (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    // padded-record kernel: the tables exist (plan's width), sum reducer, 16-byte operands
    if (p->rec.batches && N == p->N && reduce == gespmm::kReduceSum && variant_v4 && !(use_tuned && kchoice != GESPMM_PLAN_KERNEL_RECORDS) &&
        (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && !(staged && kchoice == GESPMM_PLAN_KERNEL_STAGED)) {
        if (!B || !C) return GESPMM_EINVAL;
        if (guard && guard->word == nullptr) return 0;  // (dry run: one kernel, guardable)
        return (int)gespmm::launch_spmm_records(p->rec, B, C, N, p->launch_flags, guard, reinterpret_cast<hipStream_t>(stream));
    }
    // column-slab tables: one launch of the staged-rows kernel per slab, the second and later ones continuing from C (sum reducer, the
    // plan's width, 16-byte operands; an explicit other kernel or a tuned plan keeps its choice)
    if (p->reordered && p->slab.ev && N == p->N && reduce == gespmm::kReduceSum && variant_v4 && !use_tuned &&
        (kchoice == GESPMM_PLAN_KERNEL_AUTO || kchoice == GESPMM_PLAN_KERNEL_STAGED_SLABS) && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        if (!B || !C) return GESPMM_EINVAL;
        if (guard) return gespmm::kNotGuardable;  // (several launches)
        hipStream_t hst = reinterpret_cast<hipStream_t>(stream);
        const int P = p->slab_view.slabs, nb = p->slab.nblocks / P;
        for (int s = 0; s < P; ++s) {
            gespmm::StagedArgs sa = {p->slab_view.rowptr_v, p->slab.ev, nullptr, p->slab.tasks, p->slab.hot_cols, p->slab.nhot, B, C, nb,
                                     p->slab.waves, p->slab.slots, 0, nullptr, 0, 0, 0.0f, nullptr, 0, s * nb, s > 0 ? 1 : 0};
            rc = (int)gespmm::launch_spmm_staged(sa, p->M, p->K, N, hst);
            if (rc != 0) return rc;
        }
        return 0;
    }
    if (staged) {
        if (!B || !C) return GESPMM_EINVAL;
        gespmm::StagedArgs sa = {p->stg.rowptr_s ? p->stg.rowptr_s : p->d_rowptr, p->stg.ev, p->d_perm, p->stg.tasks, p->stg.hot_cols,
                                 p->stg.nhot, B, C, p->stg.nblocks, p->stg.waves, p->stg.slots, 0, nullptr, 0, 0, 0.0f,
                                 guard ? guard->word : nullptr, guard ? guard->want : 0};
        if (guard && p->stg.nlong > 0) return gespmm::kNotGuardable;  // (hub rows take a second launch and the long-row pass)
        if (guard && guard->word == nullptr) return 0;                 // (dry run: one kernel, guardable)
        hipStream_t hst = reinterpret_cast<hipStream_t>(stream);
        if (sclass == gespmm::kStagedTuned) rc = (int)gespmm::launch_spmm_staged(sa, p->M, p->K, N, hst);
        else if (sclass == gespmm::kStagedNarrow) rc = (int)gespmm::launch_spmm_staged_narrow(sa, p->M, p->K, N, hst);
        else rc = (int)gespmm::launch_spmm_staged_gen(sa, p->M, p->K, N, reduce, empty, hst);
        if (rc == 0 && p->stg.nlong > 0) {
            // hub rows (written as empty rows above): one-row tasks through the batch-stream kernel, whose long-row pass splits
            // them — under GESPMM_FLAG_STRICT_ORDER each is one lane group's chain instead, as everywhere else
            gespmm::PlanLaunch pl = {p->stg.ltasks, p->stg.nlong, p->d_perm, nullptr, 0, false};
            gespmm_launch_cfg lcfg = cfg;
            lcfg.flags = (lcfg.flags | GESPMM_FLAG_BATCH_STREAM | GESPMM_FLAG_NO_SLAB_BLOCKED) & ~GESPMM_FLAG_REUSE_SPLIT;
            if (!(lcfg.flags & GESPMM_FLAG_STRICT_ORDER)) lcfg.flags |= GESPMM_FLAG_SPLIT_LONG_ROWS;
            rc = gespmm::run_spmm(p->d_rowptr, p->d_colind, p->valued ? p->d_val : nullptr, B, C, p->M, p->K, N, p->nnz, p->variant,
                                  &lcfg, reduce, empty, stream, ws, ws_bytes, &pl);
        }
        return rc;
    }
    if (p->reordered) {
        // (which streaming kernel: prefer_segmented; which lane geometry at narrow widths: narrow_vec4 — plan_policy.cpp)
        const bool seg = gespmm::prefer_segmented(facts, p->hits_after, N);
        const bool vec4 = use_tuned ? p->tuned_vec == 1 : gespmm::narrow_vec4(facts, p->hits_after, N);
        gespmm::PlanLaunch pl = {p->d_tasks, p->ntasks, p->d_perm, p->d_gtasks, p->ngtasks, seg && !vec4};
        rc = gespmm::run_spmm(p->d_rowptr, p->d_colind, p->valued ? p->d_val : nullptr, B, C, p->M, p->K, N, p->nnz,
                              vec4 ? GESPMM_VARIANT_CRC_CWM4 : p->variant, &cfg, reduce, empty, stream, ws, ws_bytes, &pl, guard);
    } else {
        rc = gespmm::run_spmm(p->rowptr, p->colind, p->valued ? p->val : nullptr, B, C, p->M, p->K, N, p->nnz, p->variant,
                              &cfg, reduce, empty, stream, ws, ws_bytes, nullptr, guard);
    }
    if (rc == 0 && ws) {
        p->split_ready = true;
        p->split_vec = vec_now;
    }
    return rc;
}

int gespmm_plan_spmm_f32(gespmm_plan* plan, const float* B, float* C, int64_t N, void* stream) {
    return plan_run(plan, B, C, N, gespmm::kReduceSum, 0.0f, stream);
}

int gespmm_plan_spmm_max_f32(gespmm_plan* plan, const float* B, float* C, int64_t N, float empty_value, void* stream) {
    return plan_run(plan, B, C, N, gespmm::kReduceMax, empty_value, stream);
}

// Which kernel, MEASURED: the candidates of a clustered plan — batch-stream, segmented-stream and (at the plan's width) staged-rows —
// run on the caller's operands, `reps` launches each between a pair of events; the fastest becomes the plan's kernel. Every
// candidate produces the same bits, so C holds the product afterwards whatever wins. The static rules of plan_policy.cpp
// stay the default; this is for callers that would rather pay a few launches than trust a threshold (the hold-out audit,
// profiles/r04/holdout_audit.log, is where the rules and the measurement are compared).
int gespmm_plan_tune(gespmm_plan* p, const float* B, float* C, int64_t N, int32_t reps, void* stream) {
    if (!p || N <= 0 || !B || !C) return GESPMM_EINVAL;
    if (N != p->N) return GESPMM_EINVAL;  // the tables are made for one width
    // a storage-order plan has one launch path, and the caller's explicit choice stands: nothing to measure, but C = A * B as promised
    // (... and so does a plan that kept column-slab tables: they are not one of the candidates below, and the rule that kept them —
    //  half of the entries staged on a matrix of mean degree >= 96 — is far from the margin: x0.73-0.81 wherever it fires, slab_density.log)
    if (!p->reordered || p->kernel_choice != GESPMM_PLAN_KERNEL_AUTO || p->slab.ev)
        return plan_run(p, B, C, N, gespmm::kReduceSum, 0.0f, stream);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return GESPMM_EINVAL;
    (void)hipGetLastError();
    if (reps <= 0) reps = 3;
    if (reps > 50) reps = 50;
    hipError_t e = hipSuccess;
    const bool v4 = p->variant == GESPMM_VARIANT_AUTO || p->variant == GESPMM_VARIANT_CRC_CWM4 || p->variant == GESPMM_VARIANT_CRC_CWM8;
    if (!p->stg.ev && p->analysis == GESPMM_PLAN_ANALYSIS_DEVICE && v4 && p->nnz > 0 && gespmm::staged_serves_any(p->M, p->K, p->N) && gespmm::staged_stream_fits(p->M, p->nnz)) {
        e = build_staging_tables(p, st);  // (built for the occasion: kept only if the staged-rows kernel wins)
        if (e != hipSuccess) {
            gespmm::free_staging(&p->stg);
            return (int)e;
        }
    }
    if (!p->rec.batches && v4 && p->nnz > 0 && gespmm::records_serves(p->M, p->K, p->N, p->max_degree)) {
        e = build_record_tables(p, st);  // (built for the occasion too)
        if (e != hipSuccess) {
            gespmm::free_records(&p->rec);
            (void)hipGetLastError();
            e = hipSuccess;
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (!p->staging_kept_by_policy) gespmm::free_staging(&p->stg);
        if (!p->records_kept_by_policy) gespmm::free_records(&p->rec);
        return (int)e;
    }
    const int cand[5] = {GESPMM_PLAN_KERNEL_STREAM, GESPMM_PLAN_KERNEL_SEG_STREAM, GESPMM_PLAN_KERNEL_STAGED, GESPMM_PLAN_KERNEL_STREAM,
                         GESPMM_PLAN_KERNEL_RECORDS};
    const bool was_tuned = p->tuned;
    const int was_kernel = p->tuned_kernel, was_vec = p->tuned_vec;
    int best = -1, rc = 0;
    p->tuned = true;  // (plan_run below launches the candidate through the tuned path)
    for (int c = 0; c < 5 && rc == 0; ++c) {
        p->tune_us[c] = -1.0;
        if (c == 1 && !p->d_gtasks) continue;
        if (c == 4 && !(p->rec.batches && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0)) continue;
        if (c == 2 && !(p->stg.ev && (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && v4)) continue;
        if (c == 3 && !(p->variant == GESPMM_VARIANT_AUTO && N <= 64 && N % 4 == 0)) continue;
        p->tuned_kernel = cand[c];
        p->tuned_vec = c == 3 ? 1 : 0;
        rc = plan_run(p, B, C, N, gespmm::kReduceSum, 0.0f, stream);  // warm: code objects, split points, L2 state
        if (rc == 0) rc = (int)hipEventRecord(e0, st);
        for (int r = 0; r < reps && rc == 0; ++r) rc = plan_run(p, B, C, N, gespmm::kReduceSum, 0.0f, stream);
        if (rc == 0) rc = (int)hipEventRecord(e1, st);
        if (rc == 0) rc = (int)hipEventSynchronize(e1);
        float ms = 0.0f;
        if (rc == 0) rc = (int)hipEventElapsedTime(&ms, e0, e1);
        if (rc != 0) break;
        p->tune_us[c] = (double)ms * 1e3 / reps;
        if (best < 0 || p->tune_us[c] < p->tune_us[best]) best = c;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != 0 || best < 0) {
        // a candidate failed: the plan is what it was before the call — and tables the policy had not kept do not stay behind
        // (~16 bytes per entry, and an AUTO launch would otherwise take the staged-rows kernel against the policy)
        p->tuned = was_tuned;
        p->tuned_kernel = was_kernel;
        p->tuned_vec = was_vec;
        if (!p->staging_kept_by_policy && !(was_tuned && was_kernel == GESPMM_PLAN_KERNEL_STAGED)) gespmm::free_staging(&p->stg);
        if (!p->records_kept_by_policy && !(was_tuned && was_kernel == GESPMM_PLAN_KERNEL_RECORDS)) gespmm::free_records(&p->rec);
        return rc;
    }
    p->tuned_kernel = cand[best];
    p->tuned_vec = best == 3 ? 1 : 0;
    // tables of a kernel that lost are not kept (~16 bytes per entry): launches at p->N take the winner, other widths never use the
    // staged-rows kernel, and a later tune rebuilds them (above) if it is asked again
    if (best != 2 && p->stg.ev) {
        gespmm::free_staging(&p->stg);
        p->staging_kept_by_policy = false;
    }
    if (best != 4 && p->rec.batches) {
        gespmm::free_records(&p->rec);
        p->records_kept_by_policy = false;
    }
    if (best != 2) rc = plan_run(p, B, C, N, gespmm::kReduceSum, 0.0f, stream);  // (C is the winner's product either way: same bits)
    return rc;
}

// SDDMM on the plan's pattern: out[e] = <D1[row(e), :], D2[col(e), :]> for every edge e of the CALLER's CSR (out in the
// caller's edge order). A clustered plan walks the edges in its own order — the rows of D2 that neighbouring rows share
// are then found in L2, as in the SpMM — and scatters the results back; each dot product is the same lane butterfly as in
// gespmm_sddmm_{coo,csr}_f32, so the bits are the same.
int gespmm_plan_sddmm_f32(gespmm_plan* p, const float* D1, const float* D2, float* out, int64_t N, void* stream) {
    if (!p || N < 0) return GESPMM_EINVAL;
    if (p->nnz == 0) return 0;
    if (!out || (N > 0 && (!D1 || !D2))) return GESPMM_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // which form: sddmm_route (plan_policy.cpp) — 0 CSR call, 1 COO on row ids expanded ONCE (same lane butterfly per edge, same
    // bits), 2 the plan's clustered edge order + scatter
    hipError_t e = hipSuccess;
    const int route = gespmm::sddmm_route(p->facts, p->reordered, p->hits_after, N);
    if (route != 2) {
        if (route == 1) {
            if (!p->d_coo_row_storage) {
                int32_t* rows = nullptr;
                e = hipMalloc(reinterpret_cast<void**>(&rows), (size_t)p->nnz * 4);
                if (e != hipSuccess) return (int)e;
                hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, st, p->rowptr, rows,
                                   (int)p->M, (int)p->nnz);
                e = hipGetLastError();
                if (e != hipSuccess) {
                    (void)hipFree(rows);
                    return (int)e;
                }
                p->d_coo_row_storage = rows;
            }
            return (int)gespmm::launch_sddmm(p->d_coo_row_storage, false, p->colind, D1, D2, out, p->M, p->nnz, N, 0, st);
        }
        return gespmm_sddmm_csr_f32(p->rowptr, p->colind, D1, D2, out, p->M, p->nnz, N, stream);
    }
    if (!p->d_coo_row) {
        const size_t bytes = (size_t)p->nnz * 4;
        // all three buffers or none: a half-built set must not survive into the next call
        int32_t *coo = nullptr, *dst = nullptr;
        float* tmp = nullptr;
        e = hipMalloc(reinterpret_cast<void**>(&coo), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dst), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tmp), bytes);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(plan_edge_maps_kernel, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, st, p->d_rowptr,
                               p->d_src_begin, p->d_perm, coo, dst, (int)p->M, (int)p->nnz);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            if (coo) (void)hipFree(coo);
            if (dst) (void)hipFree(dst);
            if (tmp) (void)hipFree(tmp);
            return (int)e;
        }
        p->d_coo_row = coo;
        p->d_edge_dst = dst;
        p->d_sddmm_tmp = tmp;
    }
    e = gespmm::launch_sddmm(p->d_coo_row, false, p->d_colind, D1, D2, p->d_sddmm_tmp, p->M, p->nnz, N, 0, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(scatter_by_index_kernel, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, st, p->d_sddmm_tmp,
                       p->d_edge_dst, out, (int)p->nnz);
    return (int)hipGetLastError();
}

int gespmm_plan_set_values(gespmm_plan* p, const float* val, void* stream) {
    if (!p) return GESPMM_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!p->reordered) {
        p->val = val;
        p->valued = val != nullptr;
        if (p->rec.batches) return (int)gespmm::device_records_set_values(p->rec, p->M, p->rowptr, p->colind, val, st);
        return 0;
    }
    if (!val) {
        p->valued = false;
        if (p->rec.batches) {
            const hipError_t er = gespmm::device_records_set_values(p->rec, p->M, p->d_rowptr, p->d_colind, nullptr, st);
            if (er != hipSuccess) return (int)er;
        }
        if (p->slab.ev) {
            const hipError_t es = gespmm::device_slab_set_values(p->slab, p->slab_view, nullptr, p->M * p->slab_view.slabs, p->nnz, st);
            if (es != hipSuccess) return (int)es;
        }
        if (p->stg.ev) return (int)gespmm::device_staging_set_values(p->stg, nullptr, p->d_rowptr, p->M, p->nnz, st);  // the stream carries 1.0f
        return 0;
    }
    if (!p->d_val) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_val), (size_t)(p->nnz > 0 ? p->nnz : 1) * 4);
        if (e != hipSuccess) return (int)e;
    }
    p->valued = true;
    if (p->nnz == 0) return 0;
    hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, st, p->d_rowptr,
                       p->d_src_begin, val, p->d_val, (int)p->M, (int)p->nnz);
    if (p->stg.ev) {
        const hipError_t es = gespmm::device_staging_set_values(p->stg, p->d_val, p->d_rowptr, p->M, p->nnz, st);
        if (es != hipSuccess) return (int)es;
    }
    if (p->slab.ev) {
        const hipError_t es = gespmm::device_slab_set_values(p->slab, p->slab_view, p->d_val, p->M * p->slab_view.slabs, p->nnz, st);
        if (es != hipSuccess) return (int)es;
    }
    if (p->rec.batches) {
        const hipError_t er = gespmm::device_records_set_values(p->rec, p->M, p->d_rowptr, p->d_colind, p->d_val, st);
        if (er != hipSuccess) return (int)er;
    }
    return (int)hipGetLastError();
}

int gespmm_plan_get_order(const gespmm_plan* p, int32_t* perm_host) {
    if (!p || (p->M > 0 && !perm_host)) return GESPMM_EINVAL;
    if (p->reordered && p->perm_host.size() != (size_t)p->M) {  // device analysis: the order lives on the device
        if (hipMemcpy(perm_host, p->d_perm, (size_t)p->M * 4, hipMemcpyDeviceToHost) != hipSuccess) return GESPMM_EINVAL;
    } else if (p->reordered) std::memcpy(perm_host, p->perm_host.data(), (size_t)p->M * 4);
    else
        for (int64_t i = 0; i < p->M; ++i) perm_host[i] = (int32_t)i;
    return p->reordered ? 1 : 0;
}

int gespmm_plan_describe(const gespmm_plan* p, char* out, int64_t capacity) {
    if (!p || !out || capacity <= 0) return GESPMM_EINVAL;
    char what[256] = "";
    gespmm::PlanFacts facts = p->facts;  // (what a launch at the plan's own width does: the tuned choice, if there is one)
    if (p->tuned) facts.kernel_choice = p->tuned_kernel;
    const bool vec4d = p->reordered && (p->tuned ? p->tuned_vec == 1 : gespmm::narrow_vec4(facts, p->hits_after, p->N));
    const bool seg = p->reordered && p->d_gtasks && !vec4d && gespmm::prefer_segmented(facts, p->hits_after, p->N);
    const bool staged_d = p->stg.ev && (p->tuned ? p->tuned_kernel == GESPMM_PLAN_KERNEL_STAGED
                                                 : (p->kernel_choice == GESPMM_PLAN_KERNEL_STAGED ||
                                                    (p->kernel_choice == GESPMM_PLAN_KERNEL_AUTO && p->staging_kept_by_policy)));
    gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, p->launch_flags | (p->reordered ? ((seg ? GESPMM_FLAG_SEG_STREAM : GESPMM_FLAG_BATCH_STREAM) | GESPMM_FLAG_NO_SLAB_BLOCKED) : 0)};
    gespmm_describe_launch(p->M, p->K, p->N, p->nnz, vec4d ? GESPMM_VARIANT_CRC_CWM4 : p->variant, &cfg, what, sizeof what);
    int n;
    if (p->reordered) {
        char lv[128] = "";
        int off = 0;
        for (int i = 0; i < p->stats.levels && i < 16 && off < 100; ++i)
            off += snprintf(lv + off, sizeof lv - (size_t)off, "%s%d", i ? ">" : "", p->stats.clusters[i]);
        char kern[420];
        const bool slab_d = p->slab.ev && !p->tuned && (p->kernel_choice == GESPMM_PLAN_KERNEL_AUTO || p->kernel_choice == GESPMM_PLAN_KERNEL_STAGED_SLABS) &&
                            (p->variant == GESPMM_VARIANT_AUTO || p->variant >= GESPMM_VARIANT_CRC_CWM4);
        if (slab_d)
            snprintf(kern, sizeof kern, "kernel=staged-slabs slabs=%d blocks=%d rows_in_lds<=%d staged_entries=%.3f tables=%.4fs (max / other widths: %s)",
                     p->slab_view.slabs, p->slab.nblocks, p->slab.slots, p->slab.staged_fraction, p->slab_seconds, what);
        else if (staged_d && (p->variant == GESPMM_VARIANT_AUTO || p->variant >= GESPMM_VARIANT_CRC_CWM4))
            snprintf(kern, sizeof kern, "kernel=staged-rows blocks=%d rows_in_lds<=%d staged_entries=%.3f hub_rows=%d tables=%.4fs (max / other widths: %s)",
                     p->stg.nblocks, gespmm::staged_shape_any(p->N).slots, p->stg.staged_fraction, p->stg.nlong, p->staging_seconds, what);
        else snprintf(kern, sizeof kern, "%s", what);
        if (p->rec.batches && !(p->tuned && p->tuned_kernel != GESPMM_PLAN_KERNEL_RECORDS) && !(staged_d && p->kernel_choice == GESPMM_PLAN_KERNEL_STAGED))
            snprintf(kern, sizeof kern, "kernel=padded-records tasks=%d batches_per_task>=%d batches=%d slot_fill=%.3f tables=%.4fs (max / other widths: %s)",
                     p->rec.ntasks, p->rec.target_batches, p->rec.nbatches,
                     record_slot_fill(p), p->records_seconds, what);
        char tuned[200] = "";
        if (p->tuned)
            snprintf(tuned, sizeof tuned, " tuned[us: batch-stream=%.1f segmented-stream=%.1f staged-rows=%.1f batch-stream-V4=%.1f padded-records=%.1f]",
                     p->tune_us[0], p->tune_us[1], p->tune_us[2], p->tune_us[3], p->tune_us[4]);
        n = snprintf(out, (size_t)capacity,
                     "order=%s levels=%d clusters=%s tasks=%d task_entries=%d group_tasks=%d max_degree=%d probe=%.3f l2_model=%.3f->%.3f "
                     "analysis=%.4fs on the %s (clustering %.4fs)%s | %s",
                     p->identity_order ? "storage(plan copy: as local as the clustering)" : "clustered", p->stats.levels, lv, p->ntasks, p->task_entries,
                     p->ngtasks, p->max_degree, p->facts.wedge_probe, p->hits_before,
                     p->hits_after, p->analysis_seconds, p->analysis == GESPMM_PLAN_ANALYSIS_HOST ? "host" : "device", p->cluster_seconds, tuned, kern);
    } else {
        char why[200] = "";
        if (p->cost_skipped)
            snprintf(why, sizeof why, " (analysis skipped: est. gain %.1f us x %d launches < est. cost %.0f us; wedge probe %.4f)", p->est_gain_us,
                     p->facts.expected_launches > 0 ? p->facts.expected_launches : gespmm::kDefaultExpectedLaunches, p->est_cost_us,
                     p->facts.wedge_probe);
        char kern[420];
        if (p->rec.batches)
            snprintf(kern, sizeof kern, "kernel=padded-records tasks=%d batches_per_task>=%d batches=%d slot_fill=%.3f tables=%.4fs (max / other widths: %s)", p->rec.ntasks,
                     p->rec.target_batches, p->rec.nbatches, record_slot_fill(p), p->records_seconds, what);
        else snprintf(kern, sizeof kern, "%s", what);
        n = snprintf(out, (size_t)capacity, "order=storage max_degree=%d l2_model=%.3f->%.3f analysis=%.4fs%s | %s",
                     p->max_degree, p->hits_before, p->hits_after, p->analysis_seconds, why, kern);
    }
    if (n < 0) return GESPMM_EINVAL;
    return n < capacity ? n : (int)capacity - 1;
}

void gespmm_release_cached_memory(void) { gespmm::release_cached_arena(); }

void gespmm_set_cached_memory_limit(int64_t bytes) { gespmm::set_arena_cache_limit((long long)bytes); }

void gespmm_plan_destroy(gespmm_plan* p) {
    if (!p) return;
    free_device(p);
    delete p;
}

}  // extern "C"

namespace gespmm {
int plan_spmm_guarded(gespmm_plan* plan, const float* B, float* C, int64_t N, int reduce, float empty, void* stream, const LaunchGuard* guard) {
    return plan_run(plan, B, C, N, reduce, empty, stream, guard);
}
}  // namespace gespmm
