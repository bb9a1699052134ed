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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/gespmm.h"
#include "plan.h"
#include "plan_device.h"
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
    // task-outer kernel (spmm_outer.hip): one 544-byte record per task
    int32_t* d_orecs = nullptr;
    int32_t* d_orec_src = nullptr;
    int32_t norec = 0;
    double orec_dup = 0.0;
    // SDDMM through the plan (built on first use): edges in clustered order as COO with the ORIGINAL row ids, the
    // position of every edge in the caller's CSR, and a buffer for the results in clustered order
    int32_t* d_coo_row = nullptr;
    int32_t* d_coo_row_storage = nullptr;  // storage-order plans: row id of every edge (the COO form skips the row search)
    int32_t* d_edge_dst = nullptr;
    float* d_sddmm_tmp = nullptr;
    int32_t task_entries = 0;
    // LDS-staged-rows kernel (spmm_ldsrow.hip): one 640-byte record per task
    int32_t* d_recs = nullptr;
    int32_t* d_rec_src = nullptr;  // per record entry: position of its value in the CALLER's val array (-1: none)
    int32_t nrec = 0;
    double rec_dup = 0.0;          // non-zeros per distinct B row, averaged over the records
    int kernel_choice = 0;         // GESPMM_PLAN_KERNEL_*
    std::vector<int32_t> perm_host;  // filled by the host analysis, or on demand (gespmm_plan_get_order)
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
};

namespace {

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
    if (p->gtasks_shared) p->d_gtasks = nullptr;
    void* ptrs[] = {p->d_rowptr, p->d_colind, p->d_val, p->d_perm, p->d_src_begin, p->d_tasks, p->ws, p->d_recs, p->d_rec_src, p->d_gtasks, p->d_coo_row, p->d_edge_dst, p->d_sddmm_tmp, p->d_orecs, p->d_orec_src, p->d_coo_row_storage};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    p->d_rowptr = p->d_colind = p->d_perm = p->d_src_begin = p->d_tasks = p->d_recs = p->d_rec_src = p->d_gtasks = p->d_coo_row = p->d_edge_dst = nullptr;
    p->d_sddmm_tmp = nullptr;
    p->d_coo_row_storage = nullptr;
    p->d_orecs = p->d_orec_src = nullptr;
    p->d_val = nullptr;
    p->ws = nullptr;
}


__global__ void scatter_record_values_kernel(const int32_t* __restrict__ rec_src, const float* __restrict__ val,
                                             int32_t* __restrict__ recs, int64_t nslots) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const int32_t src = rec_src[i];
    if (src < 0) return;
    const int64_t rec = i / gespmm::kRecEntries, k = i % gespmm::kRecEntries;
    reinterpret_cast<float*>(recs)[rec * gespmm::kRecWords + gespmm::kRecOffVal + k] = val[src];
}

// Cut the row-permuted matrix into the records of spmm_ldsrow.hip: consecutive rows are packed while they fit
// (<= 32 rows, <= `target` (<= 64) entries, <= 32 DISTINCT columns); a row that does not fit a record by itself
// becomes a chain of records over consecutive pieces of its entries.
struct RecordBuilder {
    std::vector<int32_t> recs, src;
    int64_t entries = 0, distinct = 0;
    int32_t nrec() const { return (int32_t)(recs.size() / gespmm::kRecWords); }
    int32_t* open() {
        recs.resize(recs.size() + gespmm::kRecWords, 0);
        src.resize(src.size() + gespmm::kRecEntries, -1);
        return recs.data() + recs.size() - gespmm::kRecWords;
    }
};

void build_records(int64_t M, int64_t K, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci,
                   const std::vector<int32_t>& src_begin, const std::vector<int32_t>& perm, int target, RecordBuilder& rb) {
    using namespace gespmm;
    if (target <= 0 || target > kRecEntries) target = kRecEntries;
    std::vector<int32_t> stamp((size_t)K, -1), slot_of((size_t)K, 0), seen((size_t)K, -1), seen_alone((size_t)K, -1);
    int32_t stamp_id = 0;
    auto set_byte = [](int32_t* rec, int byte_off, int value) {
        reinterpret_cast<uint8_t*>(rec)[byte_off] = (uint8_t)value;
    };
    int64_t i = 0;
    while (i < M) {
        // distinct columns of row i alone
        auto row_distinct = [&](int64_t r) {
            int d = 0;
            for (int32_t p = rp[r]; p < rp[r + 1]; ++p)
                if (seen_alone[ci[p]] != (int32_t)r) {
                    seen_alone[ci[p]] = (int32_t)r;
                    ++d;
                }
            return d;
        };
        const int32_t deg = rp[i + 1] - rp[i];
        if (deg > kRecEntries || (deg > kRecDistinct && row_distinct(i) > kRecDistinct)) {
            // ---- long row: chain of records = one unit
            const size_t first_word = rb.recs.size();
            int nseg = 0;
            int32_t p = rp[i];
            while (p < rp[i + 1]) {
                int32_t* rec = rb.open();
                const size_t src_base = rb.src.size() - kRecEntries;
                ++stamp_id;
                int nent = 0, ndist = 0;
                while (p < rp[i + 1] && nent < kRecEntries) {
                    const int32_t c = ci[p];
                    if (stamp[c] != stamp_id) {
                        if (ndist == kRecDistinct) break;
                        stamp[c] = stamp_id;
                        slot_of[c] = ndist;
                        rec[kRecOffDcol + ndist] = c;
                        ++ndist;
                    }
                    set_byte(rec, kRecOffSlotBytes + nent, slot_of[c]);
                    rb.src[src_base + nent] = src_begin[i] + (p - rp[i]);
                    ++nent;
                    ++p;
                }
                rec[0] = 1;
                rec[1] = nent;
                rec[2] = ndist;
                rec[3] = 3;  // continues from the previous record and into the next (fixed up below)
                rec[kRecOffCrow] = perm[i];
                set_byte(rec, kRecOffRpBytes + 0, 0);
                set_byte(rec, kRecOffRpBytes + 1, nent);
                rb.entries += nent;
                rb.distinct += ndist;
                ++nseg;
            }
            rb.recs[first_word + 3] &= ~1;                                        // first: nothing before it
            rb.recs[first_word + (size_t)(nseg - 1) * kRecWords + 3] &= ~2;       // last: nothing after it
            ++i;
            continue;
        }
        // ---- ordinary record: pack consecutive rows
        int32_t* rec = rb.open();
        const size_t src_base = rb.src.size() - kRecEntries;
        ++stamp_id;
        int nrows = 0, nent = 0, ndist = 0;
        while (i < M && nrows < kRecRows) {
            const int32_t d = rp[i + 1] - rp[i];
            if (nrows > 0 && nent + d > target) break;
            if (d > kRecEntries) break;
            // new distinct columns this row would add (row-local duplicates counted once)
            int add = 0;
            for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
                const int32_t c = ci[p];
                if (stamp[c] != stamp_id && seen[c] != (int32_t)i) {
                    seen[c] = (int32_t)i;
                    ++add;
                }
            }
            if (ndist + add > kRecDistinct) break;  // (never the record's first row: that one was checked to fit alone)
            set_byte(rec, kRecOffRpBytes + nrows, nent);
            rec[kRecOffCrow + nrows] = perm[i];
            for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
                const int32_t c = ci[p];
                if (stamp[c] != stamp_id) {
                    stamp[c] = stamp_id;
                    slot_of[c] = ndist;
                    rec[kRecOffDcol + ndist] = c;
                    ++ndist;
                }
                set_byte(rec, kRecOffSlotBytes + nent, slot_of[c]);
                rb.src[src_base + nent] = src_begin[i] + (p - rp[i]);
                ++nent;
            }
            ++nrows;
            ++i;
        }
        set_byte(rec, kRecOffRpBytes + nrows, nent);
        rec[0] = nrows;
        rec[1] = nent;
        rec[2] = ndist;
        rec[3] = 0;
        rb.entries += nent;
        rb.distinct += ndist;
    }
}

__global__ void scatter_outer_values_kernel(const int32_t* __restrict__ rec_src, const float* __restrict__ val,
                                            int32_t* __restrict__ recs, int64_t nslots) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const int32_t src = rec_src[i];
    if (src < 0) return;
    const int64_t rec = i / gespmm::kOutEntries, k = i % gespmm::kOutEntries;
    reinterpret_cast<float*>(recs)[rec * gespmm::kOutWords + gespmm::kOutOffVal + k] = val[src];
}

// Records of spmm_outer.hip: consecutive rows of the clustered order are packed (<= 8 rows, <= `target` (<= 64) entries,
// <= 32 distinct columns) as long as each of them has strictly ascending columns; the record lists the sorted union of the
// columns and, column by column, the (row, value) pairs that use it. A row with unsorted or repeated columns is a record
// of its own whose "columns" are its entries in CSR order; a row that does not fit one record is a chain of such records.
struct OuterBuilder {
    std::vector<int32_t> recs, src;
    int64_t entries = 0, distinct = 0;
    int32_t nrec() const { return (int32_t)(recs.size() / gespmm::kOutWords); }
    int32_t* open() {
        recs.resize(recs.size() + gespmm::kOutWords, 0);
        src.resize(src.size() + gespmm::kOutEntries, -1);
        return recs.data() + recs.size() - gespmm::kOutWords;
    }
};

void build_outer_records(int64_t M, int64_t K, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci,
                         const std::vector<int32_t>& src_begin, const std::vector<int32_t>& perm, int target, OuterBuilder& ob) {
    using namespace gespmm;
    (void)K;
    if (target <= 0 || target > kOutEntries) target = kOutEntries;
    auto set_byte = [](int32_t* rec, int byte_off, int value) { reinterpret_cast<uint8_t*>(rec)[byte_off] = (uint8_t)value; };
    auto ascending = [&](int64_t r) {
        for (int32_t p = rp[r] + 1; p < rp[r + 1]; ++p)
            if (ci[p] <= ci[p - 1]) return false;
        return true;
    };
    std::vector<std::pair<int32_t, int32_t>> cols;  // (column, position in the permuted CSR)
    int64_t i = 0;
    while (i < M) {
        const int32_t deg = rp[i + 1] - rp[i];
        if (!ascending(i) || deg > kOutDistinct) {
            // ---- a record (or chain) of its own: entries in CSR order, one "column" slot per entry
            const size_t first_word = ob.recs.size();
            int nseg = 0;
            int32_t p = rp[i];
            do {
                int32_t* rec = ob.open();
                const size_t sb = ob.src.size() - kOutEntries;
                int n = 0;
                while (p < rp[i + 1] && n < kOutDistinct) {
                    rec[kOutOffDcol + n] = ci[p];
                    set_byte(rec, kOutOffCptrBytes + n, n);
                    set_byte(rec, kOutOffRowBytes + n, 0);
                    ob.src[sb + n] = src_begin[i] + (p - rp[i]);
                    ++n;
                    ++p;
                }
                set_byte(rec, kOutOffCptrBytes + n, n);
                rec[0] = 1;
                rec[1] = n;
                rec[2] = n;
                rec[3] = 3;
                rec[kOutOffCrow] = perm[i];
                ob.entries += n;
                ob.distinct += n;
                ++nseg;
            } while (p < rp[i + 1]);
            ob.recs[first_word + 3] &= ~1;
            ob.recs[first_word + (size_t)(nseg - 1) * kOutWords + 3] &= ~2;
            ++i;
            continue;
        }
        // ---- pack consecutive sorted rows
        const int64_t first = i;
        int nrows = 0, nent = 0;
        cols.clear();
        while (i < M && nrows < kOutRows) {
            const int32_t d = rp[i + 1] - rp[i];
            if (d > kOutDistinct || !ascending(i)) break;
            if (nrows > 0 && nent + d > target) break;
            // distinct columns if this row joins: merge count against the sorted union so far
            std::vector<std::pair<int32_t, int32_t>> trial = cols;
            for (int32_t p = rp[i]; p < rp[i + 1]; ++p) trial.emplace_back(ci[p], (int32_t)(i - first));
            std::sort(trial.begin(), trial.end());
            int nd = 0;
            for (size_t t = 0; t < trial.size(); ++t)
                if (t == 0 || trial[t].first != trial[t - 1].first) ++nd;
            if (nd > kOutDistinct) break;
            cols.swap(trial);
            nent += d;
            ++nrows;
            ++i;
        }
        // cols = (column, row) pairs sorted by column then row: exactly the walk order of the kernel
        int32_t* rec = ob.open();
        const size_t sb = ob.src.size() - kOutEntries;
        int nd = 0;
        for (size_t t = 0; t < cols.size(); ++t) {
            if (t == 0 || cols[t].first != cols[t - 1].first) {
                rec[kOutOffDcol + nd] = cols[t].first;
                set_byte(rec, kOutOffCptrBytes + nd, (int)t);
                ++nd;
            }
            const int32_t r = cols[t].second;
            set_byte(rec, kOutOffRowBytes + (int)t, r);
            // position of (row, column) in the permuted CSR: the row is sorted, so a binary search finds it
            const int64_t row = first + r;
            const int32_t* lo = std::lower_bound(ci.data() + rp[row], ci.data() + rp[row + 1], cols[t].first);
            ob.src[sb + t] = src_begin[row] + (int32_t)(lo - (ci.data() + rp[row]));
        }
        set_byte(rec, kOutOffCptrBytes + nd, (int)cols.size());
        for (int r = 0; r < nrows; ++r) rec[kOutOffCrow + r] = perm[first + r];
        rec[0] = nrows;
        rec[1] = nent;
        rec[2] = nd;
        rec[3] = 0;
        ob.entries += nent;
        ob.distinct += nd;
    }
}

bool columns_in_range(const int32_t* colind, int64_t nnz, int64_t K) {
    for (int64_t p = 0; p < nnz; ++p)
        if ((uint32_t)colind[p] >= (uint64_t)K) return false;
    return true;
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

// Non-zeros per wavefront task. Storage-order launches take ~12 KB of gathered B per task (select.cpp);
// clustered plans take ~20 KB (40 entries at N = 128, 32 at N >= 256, 80 at N = 64: profiles/r02/plan_task_size_final.log — at
// N = 128 anything from 40 to 80 entries runs within 1 %, and the smaller task keeps fewer rows in flight per XCD: fabric bytes
// 1.48x algorithmic at 40 entries, 1.54x at 48, 1.62x at 56, plan_task_size_traffic.log), see the caller for the L2-hit case.
int default_task_entries(int64_t N) {
    const int64_t row_bytes = 4 * (N < 256 ? N : 256);
    int64_t t = (20 << 10) / (row_bytes > 0 ? row_bytes : 4);
    if (t < 32) t = 32;
    if (t > 96) t = 96;  // narrow rows (N = 32: 128-byte rows) are latency-bound per row pair: the plain path's 96 entries
    return (int)t;
}

// Records of the two opt-in kernels (LDS-staged rows, task-outer), cut on the host from the row-permuted matrix and uploaded.
hipError_t build_and_upload_records(gespmm_plan* p, const gespmm_plan_options* opt, const std::vector<int32_t>& rp,
                                    const std::vector<int32_t>& ci, const std::vector<int32_t>& src, const float* val,
                                    hipStream_t st) {
    hipError_t e = hipSuccess;
    const int64_t M = p->M, K = p->K;
    const int target = (opt && opt->task_entries > 0) ? opt->task_entries : 0;
    if (p->launch_flags & GESPMM_FLAG_SPLIT_LONG_ROWS || K <= 0) return e;  // not for matrices that need the long-row pass
    if (p->kernel_choice == GESPMM_PLAN_KERNEL_LDS_ROWS) {
        RecordBuilder rb;
        build_records(M, K, rp, ci, src, p->perm_host, target, rb);
        p->nrec = rb.nrec();
        p->rec_dup = rb.distinct > 0 ? (double)rb.entries / (double)rb.distinct : 0.0;
        e = upload(&p->d_recs, rb.recs, st);
        if (e == hipSuccess) e = upload(&p->d_rec_src, rb.src, st);
        if (e == hipSuccess && p->valued && p->nrec > 0) {
            const int64_t nslots = (int64_t)p->nrec * gespmm::kRecEntries;
            hipLaunchKernelGGL(scatter_record_values_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st,
                               p->d_rec_src, val, p->d_recs, nslots);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // rb goes out of scope
    } else if (p->kernel_choice == GESPMM_PLAN_KERNEL_OUTER) {  // opt-in: measured level with the batch-stream kernel, not ahead
        OuterBuilder ob;
        build_outer_records(M, K, rp, ci, src, p->perm_host, target, ob);
        p->norec = ob.nrec();
        p->orec_dup = ob.distinct > 0 ? (double)ob.entries / (double)ob.distinct : 0.0;
        e = upload(&p->d_orecs, ob.recs, st);
        if (e == hipSuccess) e = upload(&p->d_orec_src, ob.src, st);
        if (e == hipSuccess && p->valued && p->norec > 0) {
            const int64_t nslots = (int64_t)p->norec * gespmm::kOutEntries;
            hipLaunchKernelGGL(scatter_outer_values_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st,
                               p->d_orec_src, val, p->d_orecs, nslots);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // ob goes out of scope
    }
    return e;
}

}  // namespace

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

double gespmm_simulate_l2_hits(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, const int32_t* perm,
                               int32_t slices, int64_t window_rows) {
    if (M <= 0 || K <= 0 || !rowptr || slices < 1 || window_rows < 1) return 0.0;
    try {
        return gespmm::simulate_l2_hits(M, K, rowptr, colind, perm, slices, window_rows);
    } catch (const std::bad_alloc&) {
        return -1.0;
    }
}

// Test hook (HOST pointers): the records build_records() cuts from a matrix in the given processing order, so the
// CPU test-suite can interpret them against its CPU checker. *recs_out is malloc'ed; free() it.
int gespmm_debug_build_records(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, const int32_t* perm,
                               int32_t target, int32_t** recs_out, int32_t** src_out, int32_t* nrec_out) {
    if (!rowptr || !perm || !recs_out || !nrec_out || M < 0 || K <= 0) return GESPMM_EINVAL;
    if (M > 0 && !columns_in_range(colind, rowptr[M], K)) return GESPMM_EINVAL;  // the builders index scratch by column
    try {
        std::vector<int32_t> rp((size_t)M + 1, 0), src((size_t)M, 0), pv(perm, perm + M);
        for (int64_t i = 0; i < M; ++i) rp[i + 1] = rp[i] + (rowptr[perm[i] + 1] - rowptr[perm[i]]);
        std::vector<int32_t> ci((size_t)rp[M]);
        for (int64_t i = 0; i < M; ++i) {
            src[i] = rowptr[perm[i]];
            std::memcpy(ci.data() + rp[i], colind + src[i], (size_t)(rp[i + 1] - rp[i]) * 4);
        }
        RecordBuilder rb;
        build_records(M, K, rp, ci, src, pv, target, rb);
        *nrec_out = rb.nrec();
        *recs_out = (int32_t*)malloc(rb.recs.size() * 4 + 4);
        if (!*recs_out) return GESPMM_ENOMEM;
        std::memcpy(*recs_out, rb.recs.data(), rb.recs.size() * 4);
        if (src_out) {
            *src_out = (int32_t*)malloc(rb.src.size() * 4 + 4);
            if (!*src_out) return GESPMM_ENOMEM;
            std::memcpy(*src_out, rb.src.data(), rb.src.size() * 4);
        }
    } catch (const std::bad_alloc&) {
        return GESPMM_ENOMEM;
    }
    return 0;
}

// The same hook for the records of the task-outer kernel (spmm_outer.hip): nrec x 136 int32, src nrec x 64.
int gespmm_debug_build_outer_records(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, const int32_t* perm,
                                     int32_t target, int32_t** recs_out, int32_t** src_out, int32_t* nrec_out) {
    if (!rowptr || !perm || !recs_out || !nrec_out || M < 0 || K <= 0) return GESPMM_EINVAL;
    if (M > 0 && !columns_in_range(colind, rowptr[M], K)) return GESPMM_EINVAL;  // the builders index scratch by column
    try {
        std::vector<int32_t> rp((size_t)M + 1, 0), src((size_t)M, 0), pv(perm, perm + M);
        for (int64_t i = 0; i < M; ++i) rp[i + 1] = rp[i] + (rowptr[perm[i] + 1] - rowptr[perm[i]]);
        std::vector<int32_t> ci((size_t)rp[M]);
        for (int64_t i = 0; i < M; ++i) {
            src[i] = rowptr[perm[i]];
            std::memcpy(ci.data() + rp[i], colind + src[i], (size_t)(rp[i + 1] - rp[i]) * 4);
        }
        OuterBuilder ob;
        build_outer_records(M, K, rp, ci, src, pv, target, ob);
        *nrec_out = ob.nrec();
        *recs_out = (int32_t*)malloc(ob.recs.size() * 4 + 4);
        if (!*recs_out) return GESPMM_ENOMEM;
        std::memcpy(*recs_out, ob.recs.data(), ob.recs.size() * 4);
        if (src_out) {
            *src_out = (int32_t*)malloc(ob.src.size() * 4 + 4);
            if (!*src_out) return GESPMM_ENOMEM;
            std::memcpy(*src_out, ob.src.data(), ob.src.size() * 4);
        }
    } catch (const std::bad_alloc&) {
        return GESPMM_ENOMEM;
    }
    return 0;
}

// The device analysis by itself (DEVICE rowptr / colind, HOST outputs) — what tests compare with gespmm_cluster_rows.
int gespmm_device_cluster_rows(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                               int32_t* perm_out_host, int32_t* levels_out, int32_t* clusters_out /* [16] */, void* stream) {
    if (M < 0 || K < 0 || nnz < 0 || (M > 0 && (!rowptr || !perm_out_host)) || (nnz > 0 && !colind)) return GESPMM_EINVAL;
    if (M == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t max_deg = 0, bad = 0;
    hipError_t e = gespmm::device_validate_csr(rowptr, colind, M, K, nnz, &max_deg, &bad, st);
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

int gespmm_plan_create(gespmm_plan** out, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M,
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
    int user_flags = opt ? opt->flags : 0;

    int analysis = opt ? opt->analysis : GESPMM_PLAN_ANALYSIS_DEVICE;
    if (const char* v = getenv("GESPMM_PLAN_ANALYSIS")) analysis = (v[0] == 'h') ? GESPMM_PLAN_ANALYSIS_HOST : GESPMM_PLAN_ANALYSIS_DEVICE;  // debugging aid
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
        e = gespmm::device_validate_csr(rowptr, colind, M, K, nnz, &max_deg, &bad, st);
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
        // long-row pass: decided by the longest row (the plain entry points have to guess)
        const int64_t mean = (M > 0) ? (nnz + M - 1) / M : 0;
        int64_t threshold = 32 * mean;
        if (threshold < gespmm::kLongRowThreshold) threshold = gespmm::kLongRowThreshold;
        if (!(user_flags & (GESPMM_FLAG_STRICT_ORDER | GESPMM_FLAG_SPLIT_LONG_ROWS)))
            user_flags |= (max_deg > threshold) ? GESPMM_FLAG_SPLIT_LONG_ROWS : GESPMM_FLAG_STRICT_ORDER;
        p->launch_flags = user_flags;

        // ---- what would a plain call launch? (the cache-blocked path keeps the storage order)
        gespmm::Selection sel;
        int max_vec = 4;
        while (max_vec > 1 && (N % max_vec) != 0) max_vec >>= 1;
        if (gespmm::resolve_geometry(M, K, N > 0 ? N : 1, nnz, variant, max_vec, 0, 0, 0, 0, 0, user_flags, &sel) != 0) {
            delete p;
            return GESPMM_EINVAL;
        }
        const bool stream_family = sel.variant >= GESPMM_VARIANT_CRC && sel.variant <= GESPMM_VARIANT_CRC_CWM8 &&
                                   !sel.geo.slab_blocked;
        const int64_t tile_cols = (int64_t)sel.geo.group * sel.geo.vec * sel.geo.strips;
        const int64_t b_bytes = K * 4 * (N < tile_cols ? N : tile_cols);
        bool reorder = false;
        if (reorder_mode == GESPMM_PLAN_REORDER) reorder = stream_family && M > 1 && nnz > 0;
        else if (reorder_mode == GESPMM_PLAN_REORDER_AUTO)
            // B beyond the L2s (below that every order hits), enough rows to cluster
            reorder = stream_family && M >= (1 << 14) && nnz >= M && b_bytes > (8ll << 20) && mean <= 96 &&
                      nnz <= (1ll << 28);
        // Dense graphs (the plain call's cache-blocked path): worth clustering only when they have STRONG community structure — a
        // reddit-sized graph with planted communities modelled at 0.71-0.77 hits runs 3.0 vs 4.0 ms at N = 128 (1.6 vs 2.2 at 64,
        // 6.5 vs 8.3 at 256) through a clustered plan; modelled at 0.37-0.50 the cache-blocked path wins (4.1 vs 4.9 ms), and on the
        // structureless stand-in by 2x (profiles/r03/dense_community_audit.log). AUTO runs the analysis and keeps the clustered order
        // only from 0.65 on; otherwise the tables are dropped and the cache-blocked path stays.
        bool dense_try = false;
        if (reorder_mode == GESPMM_PLAN_REORDER_AUTO && !reorder && !on_host && (sel.geo.slab_blocked || mean > 96) &&
            sel.variant >= GESPMM_VARIANT_CRC && sel.variant <= GESPMM_VARIANT_CRC_CWM8 && M >= (1 << 14) && nnz >= M &&
            nnz <= (1ll << 28) && b_bytes > (8ll << 20)) {
            reorder = true;
            dense_try = true;
        }
        // the model of the XCD L2s: window = B rows that 3 MiB hold; matrices beyond 2^25 non-zeros: the first 2^22
        // non-zeros of each of the 8 slices are the sample
        const int64_t model_sample = nnz <= (1ll << 25) ? 0 : (1ll << 22);
        const int64_t model_rowb = 4 * (N < tile_cols ? N : tile_cols);
        const int64_t model_window = (3ll << 20) / (model_rowb > 0 ? model_rowb : 4);

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
            e = hipMalloc(reinterpret_cast<void**>(&p->d_perm), (size_t)M * 4);
            gespmm::ClusterOptions copt = cluster_options_from_env();
            if (e == hipSuccess) e = gespmm::device_cluster_rows(M, K, nnz, rowptr, colind, copt, p->d_perm, &p->stats, st);
            p->cluster_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
            lap("cluster");
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_rowptr), ((size_t)M + 1) * 4);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_colind), (size_t)(nnz > 0 ? nnz : 1) * 4);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_src_begin), (size_t)M * 4);
            if (e == hipSuccess)
                e = gespmm::device_permute_csr(M, nnz, rowptr, colind, p->d_perm, p->d_rowptr, p->d_colind, p->d_src_begin, st);
            lap("permute");
            const auto tm = std::chrono::steady_clock::now();
            if (e == hipSuccess)
                e = gespmm::device_l2_model(M, K, nnz, rowptr, colind, 8, model_window, model_sample, 4096, &p->hits_before, st);
            if (e == hipSuccess)
                e = gespmm::device_l2_model(M, K, nnz, p->d_rowptr, p->d_colind, 8, model_window, model_sample, 4096,
                                            &p->hits_after, st);
            p->model_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tm).count();
            lap("l2 model x2");
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            if (reorder_mode == GESPMM_PLAN_REORDER_AUTO && (p->hits_after < p->hits_before + 0.05 || (dense_try && p->hits_after < 0.65))) {
                reorder = false;  // the storage order (or the cache-blocked path) is as good: keep it and pay nothing per launch
                (void)hipFree(p->d_perm);
                (void)hipFree(p->d_rowptr);
                (void)hipFree(p->d_colind);
                (void)hipFree(p->d_src_begin);
                p->d_perm = p->d_rowptr = p->d_colind = p->d_src_begin = nullptr;
            }
        }
        if (reorder && !on_host) {
            if (dense_try) p->launch_flags |= GESPMM_FLAG_NO_SLAB_BLOCKED;  // a clustered dense graph runs the streaming kernels
            // ---- task tables (same greedy cut as the host path), values, optional records
            int budget = (opt && opt->task_entries > 0) ? opt->task_entries : default_task_entries(N);
            if (!(opt && opt->task_entries > 0) && budget < 5 * mean) budget = (int)(5 * mean < 512 ? 5 * mean : 512);
            const int floor_opt = opt ? opt->row_floor : 0;
            const int64_t row_floor = floor_opt < 0 ? 0 : (floor_opt > 0 ? floor_opt : 8);
            p->task_entries = budget;
            int gbudget = budget / 2 > 16 ? budget / 2 : 16;
            if (opt && opt->task_entries > 0) gbudget = opt->task_entries / 2 > 4 ? opt->task_entries / 2 : 4;
            else if (mean < 16) gbudget = 16;
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
            if (e == hipSuccess && p->valued) e = hipMalloc(reinterpret_cast<void**>(&p->d_val), (size_t)(nnz > 0 ? nnz : 1) * 4);
            if (e == hipSuccess && p->valued && nnz > 0) {
                hipLaunchKernelGGL(permute_values_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, p->d_rowptr,
                                   p->d_src_begin, val, p->d_val, (int)M, (int)nnz);
                e = hipGetLastError();
            }
            p->kernel_choice = opt ? opt->kernel : GESPMM_PLAN_KERNEL_AUTO;
            const bool want_recs = (p->kernel_choice == GESPMM_PLAN_KERNEL_LDS_ROWS || p->kernel_choice == GESPMM_PLAN_KERNEL_OUTER) &&
                                   !(p->launch_flags & GESPMM_FLAG_SPLIT_LONG_ROWS) && K > 0;
            if (e == hipSuccess && want_recs) {
                // the two opt-in record kernels cut their records on the host: the permuted matrix travels once
                std::vector<int32_t> rp((size_t)M + 1), ci((size_t)nnz), src((size_t)M);
                p->perm_host.resize((size_t)M);
                e = hipMemcpyAsync(rp.data(), p->d_rowptr, ((size_t)M + 1) * 4, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess && nnz > 0) e = hipMemcpyAsync(ci.data(), p->d_colind, (size_t)nnz * 4, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipMemcpyAsync(src.data(), p->d_src_begin, (size_t)M * 4, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipMemcpyAsync(p->perm_host.data(), p->d_perm, (size_t)M * 4, hipMemcpyDeviceToHost, st);
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                if (e == hipSuccess) e = build_and_upload_records(p, opt, rp, ci, src, val, st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            lap("values");
            // ---- staged-rows kernel: worth its tables where a block of clustered rows uses the same B rows again and again
            // (profiles/r03/staged_rows.log, staged_degree_sweep.log; products-shaped communities: 3.0 vs 3.9 ms at N = 128, 5.8 vs
            // 7.8 ms at N = 256). At N = 128 a row is half of what a load instruction could carry and short rows are level at best
            // (com-Amazon-shaped communities: 108 vs 106 us; mean degree 8: 239 vs 236 us; from 12 on: 9-15 % ahead) — AUTO asks for
            // mean degree >= 12; at N = 256 short rows win as well (com-Amazon-shaped: 196 vs 208 us; mean degree 8: 416 vs 450 us).
            // One wavefront walks a row's entries one after the other, so hub rows stay with the streaming kernels.
            {
                const int H = gespmm::staged_rows_per_block_lds(N);
                const bool fits = H > 0 && nnz > 0 && (uint64_t)K * (uint64_t)N * 4ull < 0xFFFF0000ull;
                const bool want = p->kernel_choice == GESPMM_PLAN_KERNEL_STAGED ||
                                  (p->kernel_choice == GESPMM_PLAN_KERNEL_AUTO && mean >= (N >= 256 ? 5 : 12) && p->hits_after >= 0.40 &&
                                   nnz >= (1 << 20) &&
                                   (variant == GESPMM_VARIANT_AUTO || variant == GESPMM_VARIANT_CRC_CWM4 ||
                                    variant == GESPMM_VARIANT_CRC_CWM8));
                if (e == hipSuccess && fits && want) {
                    const auto ts = std::chrono::steady_clock::now();
                    // hub rows (one wavefront would walk such a row alone) are taken out: the staged kernel sees them empty, the
                    // streaming kernel's long-row pass gets them as one-row tasks (plan_run)
                    const int32_t* rp_s = p->d_rowptr;
                    const int32_t* ci_s = p->d_colind;
                    const float* val_s = p->valued ? p->d_val : nullptr;
                    int32_t* ci_tmp = nullptr;
                    float* val_tmp = nullptr;
                    int64_t nnz_s = nnz;
                    if (p->max_degree > gespmm::kStagedMaxRow) {
                        e = gespmm::device_split_long_rows(M, nnz, p->d_rowptr, p->d_colind, val_s, gespmm::kStagedMaxRow, &p->stg,
                                                           &ci_tmp, &val_tmp, st);
                        rp_s = p->stg.rowptr_s;
                        ci_s = ci_tmp;
                        val_s = val_tmp;
                        nnz_s = p->stg.nnz_s;
                    }
                    if (e == hipSuccess && nnz_s > 0)
                        e = gespmm::device_build_staging(M, K, nnz_s, rp_s, ci_s, val_s, p->d_perm, gespmm::staged_block_rows(N), H,
                                                         &p->stg, st);
                    if (ci_tmp) (void)hipFree(ci_tmp);
                    if (val_tmp) (void)hipFree(val_tmp);
                    if (e == hipSuccess && !p->stg.ev) gespmm::free_staging(&p->stg);  // (nothing but hub rows)
                    p->staging_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
                    if (e == hipSuccess && p->kernel_choice == GESPMM_PLAN_KERNEL_AUTO && p->stg.staged_fraction < 0.40)
                        gespmm::free_staging(&p->stg);  // not enough reuse inside the blocks: the streaming kernels stay
                    lap("staging tables");
                }
            }
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            p->reordered = true;
            if (p->hits_after >= 0.40 && N <= 128 && mean <= 8 && nnz >= (1 << 20) && !(opt && opt->flags & 0x20000))
                p->launch_flags |= GESPMM_FLAG_SHALLOW_UNROLL;  // see the host branch below
            reorder = false;  // done: skip the host branch
        }

        if (reorder) {
            const auto tc = std::chrono::steady_clock::now();
            p->perm_host.resize((size_t)M);
            gespmm::ClusterOptions copt = cluster_options_from_env();
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
                if (reorder_mode == GESPMM_PLAN_REORDER_AUTO && p->hits_after < p->hits_before + 0.05) reorder = false;
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
            // task size: ~32 KB of gathered B per wavefront (plan_task_size.log, plan_unroll_geometry.log)
            int budget = (opt && opt->task_entries > 0) ? opt->task_entries : default_task_entries(N);
            // ... but never fewer than ~5 rows of mean length per task (products-shaped graphs, degree 50: 256-entry tasks
            // at N = 32 run 1.48 ms, 96-entry tasks 2.33 ms)
            if (!(opt && opt->task_entries > 0) && budget < 5 * mean) budget = (int)(5 * mean < 512 ? 5 * mean : 512);
            const int floor_opt = opt ? opt->row_floor : 0;
            const int64_t row_floor = floor_opt < 0 ? 0 : (floor_opt > 0 ? floor_opt : 8);
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
            // (short rows: 16 entries per lane group — profiles/r02/plan_seg_task_size.log: 139 us at 16, 146 at 24, 150 at 32
            // on the com-Amazon stand-in)
            int gbudget = budget / 2 > 16 ? budget / 2 : 16;
            if (opt && opt->task_entries > 0) gbudget = opt->task_entries / 2 > 4 ? opt->task_entries / 2 : 4;
            else if (mean < 16) gbudget = 16;
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
            p->kernel_choice = opt ? opt->kernel : GESPMM_PLAN_KERNEL_AUTO;
            if (e == hipSuccess) e = build_and_upload_records(p, opt, rp, ci, src, val, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // the host vectors go out of scope
            if (e != hipSuccess) {
                free_device(p);
                delete p;
                return (int)e;
            }
            p->reordered = true;
            // Where the clustered order is modelled to hit L2 (>= 40 % of the gathers) four B rows in flight per lane group
            // beat eight at up to 128 columns (com-Amazon-shaped communities, N = 128: 105 vs 114 us, N = 64: 48 vs 60 us;
            // at 256+ columns and on the structureless graph eight stay ahead) — profiles/r02/plan_unroll_geometry.log
            // (short rows only: degree-50 rows want the depth — products-shaped communities, N = 32: 525 vs 365 us)
            if (p->hits_after >= 0.40 && N <= 128 && mean <= 8 && nnz >= (1 << 20) && !(opt && opt->flags & 0x20000))
                p->launch_flags |= GESPMM_FLAG_SHALLOW_UNROLL;
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
    p->analysis_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    *out = p;
    return 0;
}

// Which streaming kernel a clustered plan launches (AUTO rule + the caller's choice).
//   segmented-stream (one continuous gather stream per lane group): ahead of the batch kernel on clustered matrices with
//     longer rows at one column tile (products-shaped communities, N = 128: 3.95 vs 4.37 ms; N = 16: 1.06 vs 1.17 ms; N = 32:
//     1.35 vs 1.38), behind at N = 64 (2.25 vs 2.03);
//   batch-stream otherwise — on short rows the two are within 2 % of each other at N >= 128 (com-Amazon stand-ins: 138.5 vs
//     140.1 us at 128, 267.7 vs 261.8 at 256, 574.6 vs 570.4 at 512) and the batch kernel is far ahead below (N = 64: 61 vs 91 us)
//     and on small graphs (pubmed N = 128: 9.6 vs 13.2 us) — and whenever long rows are split (run_spmm decides that).
//   (profiles/r02/plan_seg_widths.log; an earlier version of the planned batch kernel carried runtime plan / persistent-task
//   branches and lost 5 % to the segmented kernel on short rows — see plain_path_regression.log.)
static bool plan_prefers_segmented(const gespmm_plan* p, int64_t N) {
    if (p->kernel_choice == GESPMM_PLAN_KERNEL_SEG_STREAM) return true;
    if (p->kernel_choice != GESPMM_PLAN_KERNEL_AUTO) return false;
    const int64_t mean_deg = p->M > 0 ? p->nnz / p->M : 0;
    // (dense clustered graphs, mean degree in the hundreds: segmented also at N = 256 and 512 — 6.36 vs 7.03 ms and 15.5 vs 16.8 ms on the
    // reddit-sized community graph, profiles/r03/dense_community_audit.log)
    return p->nnz >= (1 << 20) && N % 4 == 0 && mean_deg >= 16 && p->hits_after >= 0.40 &&
           (N <= 32 || (N > 64 && N <= 128) || (mean_deg >= 128 && N > 64 && N <= 512));
}

static int plan_run(gespmm_plan* p, const float* B, float* C, int64_t N, int reduce, float empty, void* stream) {
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
    bool lds_rows = p->reordered && p->d_recs && p->nrec > 0 && gespmm::ldsrow_group_width(N) > 0 && variant_v4 &&
                    (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    // opt-in (GESPMM_PLAN_KERNEL_LDS_ROWS): 118-128 us vs 108-113 us for the batch-stream kernel on the clustered bench
    // graph — its sums are instruction-issue bound at the 8 wavefronts per CU its LDS footprint allows
    if (lds_rows && p->kernel_choice != GESPMM_PLAN_KERNEL_LDS_ROWS) lds_rows = false;
    // staged-rows kernel: its tables exist (the plan decided at creation), same width, sum reducer, 16-byte operands
    const bool staged = p->reordered && p->stg.ev && N == p->N && reduce == gespmm::kReduceSum && variant_v4 && !lds_rows &&
                        p->kernel_choice != GESPMM_PLAN_KERNEL_OUTER && (reinterpret_cast<uintptr_t>(B) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    if (staged) {
        if (!B || !C) return GESPMM_EINVAL;
        gespmm::StagedArgs sa = {p->stg.rowptr_s ? p->stg.rowptr_s : p->d_rowptr, p->stg.ev, p->d_perm, p->stg.tasks, p->stg.hot_cols,
                                 p->stg.nhot, B, C, p->stg.nblocks};
        rc = (int)gespmm::launch_spmm_staged(sa, N, reinterpret_cast<hipStream_t>(stream));
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
    const int oV = gespmm::outer_vec_width(N);
    bool outer = p->reordered && p->d_orecs && p->norec > 0 && oV > 0 && variant_v4 && !lds_rows &&
                 (reinterpret_cast<uintptr_t>(B) % (4u * oV)) == 0 && (reinterpret_cast<uintptr_t>(C) % (4u * oV)) == 0;
    if (outer && p->kernel_choice != GESPMM_PLAN_KERNEL_OUTER) outer = false;
    if (outer) {
        if (!B || !C) return GESPMM_EINVAL;
        gespmm::OuterArgs oa;
        oa.recs = p->d_orecs;
        oa.B = B;
        oa.C = C;
        oa.nrec = p->norec;
        oa.N = (int32_t)N;
        oa.ntile = oa.nblk = 0;
        oa.empty = empty;
        const bool idx64 = (uint64_t)p->K * (uint64_t)N * 4ull >= (1ull << 32);
        rc = (int)gespmm::launch_spmm_outer(oa, p->valued, idx64, reduce, reinterpret_cast<hipStream_t>(stream));
    } else if (lds_rows) {
        if (!B || !C) return GESPMM_EINVAL;
        gespmm::LdsRowArgs la;
        la.recs = p->d_recs;
        la.B = B;
        la.C = C;
        la.nrec = p->nrec;
        la.N = (int32_t)N;
        la.ntile = la.nblk = 0;
        la.empty = empty;
        static const int dbg = getenv("GESPMM_LDSROW_DEBUG") ? atoi(getenv("GESPMM_LDSROW_DEBUG")) : 0;
        la.debug = dbg;
        const bool idx64 = (uint64_t)p->K * (uint64_t)N * 4ull >= (1ull << 32);
        rc = (int)gespmm::launch_spmm_ldsrow(la, p->valued, idx64, reduce, reinterpret_cast<hipStream_t>(stream));
    } else if (p->reordered) {
        // (which streaming kernel: plan_prefers_segmented)
        const bool seg = plan_prefers_segmented(p, N);
        gespmm::PlanLaunch pl = {p->d_tasks, p->ntasks, p->d_perm, p->d_gtasks, p->ngtasks, seg};
        rc = gespmm::run_spmm(p->d_rowptr, p->d_colind, p->valued ? p->d_val : nullptr, B, C, p->M, p->K, N, p->nnz,
                              p->variant, &cfg, reduce, empty, stream, ws, ws_bytes, &pl);
    } else {
        rc = gespmm::run_spmm(p->rowptr, p->colind, p->valued ? p->val : nullptr, B, C, p->M, p->K, N, p->nnz, p->variant,
                              &cfg, reduce, empty, stream, ws, ws_bytes, nullptr);
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

// SDDMM on the plan's pattern: out[e] = <D1[row(e), :], D2[col(e), :]> for every edge e of the CALLER's CSR (out in the
// caller's edge order). A clustered plan walks the edges in its own order — the rows of D2 that neighbouring rows share
// are then found in L2, as in the SpMM — and scatters the results back; each dot product is the same lane butterfly as in
// gespmm_sddmm_{coo,csr}_f32, so the bits are the same.
int gespmm_plan_sddmm_f32(gespmm_plan* p, const float* D1, const float* D2, float* out, int64_t N, void* stream) {
    if (!p || N < 0) return GESPMM_EINVAL;
    if (p->nnz == 0) return 0;
    if (!out || (N > 0 && (!D1 || !D2))) return GESPMM_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // The clustered walk pays a scatter pass at the end: worth it where the order is modelled to hit L2 for >= 40 % of the
    // gathers and the rows are >= 256 bytes (com-Amazon-shaped communities, N = 128: 114 vs 151 us COO / 167 us CSR; on the
    // structureless graph or at N = 41 it is equal or slower — profiles/r02/sddmm_plan.log). Otherwise: the plain CSR
    // form on the caller's arrays (which must therefore still be alive).
    hipError_t e = hipSuccess;
    if (!p->reordered || p->hits_after < 0.40 || N < 64) {
        // Short rows: the COO form on row ids expanded ONCE (the CSR form spends a row search per wavefront: 4-18 % on
        // com-Amazon-shaped patterns, profiles/r03/sddmm_audit.log); same lane butterfly per edge, same bits. Long rows
        // keep the CSR call, whose row-walking / cache-blocked forms need no row ids at all.
        if (p->M > 0 && p->nnz / p->M < 32) {
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
        return 0;
    }
    if (!val) {
        p->valued = false;
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
    if (p->d_orecs && p->norec > 0) {
        const int64_t nslots = (int64_t)p->norec * gespmm::kOutEntries;
        hipLaunchKernelGGL(scatter_outer_values_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st,
                           p->d_orec_src, val, p->d_orecs, nslots);
    }
    if (p->d_recs && p->nrec > 0) {
        const int64_t nslots = (int64_t)p->nrec * gespmm::kRecEntries;
        hipLaunchKernelGGL(scatter_record_values_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st,
                           p->d_rec_src, val, p->d_recs, nslots);
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
    const bool seg = p->reordered && p->d_gtasks && plan_prefers_segmented(p, p->N);
    gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, p->launch_flags | (p->reordered ? ((seg ? GESPMM_FLAG_SEG_STREAM : GESPMM_FLAG_BATCH_STREAM) | GESPMM_FLAG_NO_SLAB_BLOCKED) : 0)};
    gespmm_describe_launch(p->M, p->K, p->N, p->nnz, p->variant, &cfg, what, sizeof what);
    int n;
    if (p->reordered) {
        char lv[128] = "";
        int off = 0;
        for (int i = 0; i < p->stats.levels && i < 16 && off < 100; ++i)
            off += snprintf(lv + off, sizeof lv - (size_t)off, "%s%d", i ? ">" : "", p->stats.clusters[i]);
        char kern[420];
        const int W = gespmm::ldsrow_group_width(p->N);
        const bool lds = p->d_recs && p->nrec > 0 && W > 0 && p->kernel_choice == GESPMM_PLAN_KERNEL_LDS_ROWS &&
                         (p->variant == GESPMM_VARIANT_AUTO || p->variant >= GESPMM_VARIANT_CRC_CWM4);
        const int oV = gespmm::outer_vec_width(p->N);
        const bool outer = !lds && p->d_orecs && p->norec > 0 && oV > 0 &&
                           p->kernel_choice == GESPMM_PLAN_KERNEL_OUTER &&
                           (p->variant == GESPMM_VARIANT_AUTO || p->variant >= GESPMM_VARIANT_CRC_CWM4);
        if (outer) snprintf(kern, sizeof kern, "kernel=task-outer V=%d records=%d nnz_per_distinct_row=%.2f", oV, p->norec, p->orec_dup);
        else if (lds) snprintf(kern, sizeof kern, "kernel=lds-rows V=4 W=%d records=%d nnz_per_distinct_row=%.2f", W, p->nrec, p->rec_dup);
        else if (p->stg.ev && !outer && !lds && (p->variant == GESPMM_VARIANT_AUTO || p->variant >= GESPMM_VARIANT_CRC_CWM4))
            snprintf(kern, sizeof kern, "kernel=staged-rows blocks=%d rows_in_lds<=%d staged_entries=%.3f hub_rows=%d tables=%.4fs (max / other widths: %s)",
                     p->stg.nblocks, gespmm::staged_rows_per_block_lds(p->N), p->stg.staged_fraction, p->stg.nlong, p->staging_seconds, what);
        else snprintf(kern, sizeof kern, "%s", what);
        n = snprintf(out, (size_t)capacity,
                     "order=clustered levels=%d clusters=%s tasks=%d task_entries=%d group_tasks=%d max_degree=%d l2_model=%.3f->%.3f "
                     "analysis=%.4fs on the %s (clustering %.4fs) | %s",
                     p->stats.levels, lv, p->ntasks, p->task_entries, p->ngtasks, p->max_degree, p->hits_before, p->hits_after,
                     p->analysis_seconds, p->analysis == GESPMM_PLAN_ANALYSIS_HOST ? "host" : "device", p->cluster_seconds, kern);
    } else {
        n = snprintf(out, (size_t)capacity, "order=storage max_degree=%d l2_model=%.3f->%.3f analysis=%.3fs | %s",
                     p->max_degree, p->hits_before, p->hits_after, p->analysis_seconds, what);
    }
    if (n < 0) return GESPMM_EINVAL;
    return n < capacity ? n : (int)capacity - 1;
}

void gespmm_release_cached_memory(void) { gespmm::release_cached_arena(); }

void gespmm_plan_destroy(gespmm_plan* p) {
    if (!p) return;
    free_device(p);
    delete p;
}

}  // extern "C"
