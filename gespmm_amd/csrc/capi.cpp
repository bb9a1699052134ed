// capi.cpp — the extern "C" boundary declared in include/gespmm.h.
//
// Argument validation, variant -> launch-geometry selection, and the hand-off to
// the HIP launchers. Nothing here touches device memory; the only HIP calls are
// the kernel launches themselves. There is deliberately no CPU fallback: on a
// machine without a HIP device the launch fails and its hipError_t is returned.

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/gespmm.h"
#include "select.h"
#include "auto_plan.h"
#include "plan.h"
#include "spmm_kernels.h"

namespace {

using gespmm::Geometry;
using gespmm::SpmmArgs;

inline bool aligned_to(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_common(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, const float* C,
                 int64_t M, int64_t K, int64_t N, int64_t nnz) {
    if (M < 0 || K < 0 || N < 0 || nnz < -1) return GESPMM_EINVAL;
    // CSR positions are int32 and the kernels look up to a few tiles past a row's end before clamping
    if (M > 0x7fffffffLL - 64 || K > 0x7fffffffLL || N > 0x7fffffffLL / 4 || nnz > 0x7fffffffLL - 4096)
        return GESPMM_ERANGE;
    if (M == 0 || N == 0) return 0;  // nothing to do; pointers may be null
    if (!rowptr || !C) return GESPMM_EINVAL;
    if ((nnz != 0) && (!colind || !B)) return GESPMM_EINVAL;
    if (!aligned_to(rowptr, 4) || !aligned_to(colind, 4) || !aligned_to(val, 4) || !aligned_to(B, 4) ||
        !aligned_to(C, 4))
        return GESPMM_EALIGN;
    return 0;
}

}  // namespace

namespace gespmm {

// The one place every SpMM entry point ends up in (plan.cpp included: `pl` carries the plan's task table).
int run_spmm(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C, int64_t M,
             int64_t K, int64_t N, int64_t nnz, int variant, const gespmm_launch_cfg* cfg, int reduce, float empty,
             void* stream, void* ws, int64_t ws_bytes, const PlanLaunch* pl, const LaunchGuard* guard) {
    const int rc = check_common(rowptr, colind, val, B, C, M, K, N, nnz);
    if (rc != 0) return rc;
    if (M == 0 || N == 0) return 0;
    if (variant < GESPMM_VARIANT_AUTO || variant >= GESPMM_NUM_VARIANTS) return GESPMM_EINVAL;
    if (reduce == gespmm::kReduceMax && (val != nullptr || variant == GESPMM_VARIANT_PARREDUCE ||
                                         variant == GESPMM_VARIANT_NAIVE))
        return GESPMM_EINVAL;
    if (cfg && (cfg->rows_per_wave < 0 || cfg->rows_per_wave > gespmm::kMaxRowsPerWave || cfg->slab_rows < 0))
        return GESPMM_EINVAL;

    // Vector width is limited by what both B and C rows can be addressed with.
    int max_vec = 4;
    while (max_vec > 1 && ((N % max_vec) != 0 || !aligned_to(B, 4u * max_vec) || !aligned_to(C, 4u * max_vec)))
        max_vec >>= 1;

    gespmm::Selection sel;
    int flags = 0;
    if (cfg) flags = cfg->flags;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {
        // Under stream capture the paths that need a stream-ordered temporary are switched off: a
        // captured allocation becomes a mem-alloc graph node, and replaying those costs seconds per
        // launch on this runtime (measured: 14 s per GCN epoch on reddit-like instead of 20 ms).
        // The streaming kernels need no workspace and give the same bits (slab path) or the strict
        // CSR-order chain (long-row pass).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (ws != nullptr && ws_bytes > 0) {
            // the caller's workspace: nothing is allocated whether capturing or not (a workspace that
            // turns out too small falls back to the pool — gespmm_csr_spmm_workspace_bytes is an upper bound)
        } else if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            flags |= gespmm::kFlagNoSlabBlocked | gespmm::kFlagStrictOrder;
            flags &= ~(gespmm::kFlagSlabBlocked | gespmm::kFlagSplitLongRows);
        } else {
            (void)hipGetLastError();
        }
    }
    if (pl) {  // task tables exist for the two streaming kernels only
        flags |= gespmm::kFlagNoSlabBlocked;
        flags &= ~(gespmm::kFlagSlabBlocked | gespmm::kFlagSegStream | gespmm::kFlagBatchStream);
        flags |= (pl->prefer_segmented && pl->gtasks) ? gespmm::kFlagSegStream : gespmm::kFlagBatchStream;
        flags &= ~gespmm::kFlagAllowReassoc;  // a plan runs the CRC family only (the parallel-reduction variant has no task table)
    }
    const int src = gespmm::resolve_geometry(M, K, N, nnz, variant, max_vec, cfg ? cfg->vec : 0,
                                             cfg ? cfg->strips : 0, cfg ? cfg->group : 0,
                                             cfg ? cfg->rows_per_wave : 0, cfg ? cfg->slab_rows : 0, flags, &sel);
    if (src != 0) return src;
    sel.geo.reduce = reduce;
    if (pl && (sel.variant == GESPMM_VARIANT_NAIVE || sel.variant == GESPMM_VARIANT_PARREDUCE)) return GESPMM_EINVAL;

    SpmmArgs a;
    a.rowptr = rowptr;
    a.colind = colind;
    a.val = val;
    a.B = B;
    a.C = C;
    a.M = (int32_t)M;
    a.N = (int32_t)N;
    a.nblk = 0;
    a.ntile = 0;
    a.flags = flags | (sel.geo.sc1_store ? gespmm::kFlagSc1Store : 0);
    a.empty = empty;
    a.long_row = 0;
    a.lr_hdr = nullptr;
    a.lr_rows = nullptr;
    a.lr_chunks = nullptr;
    a.lr_chunk = a.lr_max_rows = a.lr_max_chunks = 0;
    a.row_begin = nullptr;
    a.row_end = nullptr;
    a.accumulate = 0;
    a.tasks = pl ? pl->tasks : nullptr;
    a.perm = pl ? pl->perm : nullptr;
    a.ntasks = pl ? pl->ntasks : 0;
    a.gtasks = pl ? pl->gtasks : nullptr;
    a.ngtasks = pl ? pl->ngtasks : 0;
    a.guard = guard ? guard->word : nullptr;
    a.guard_want = guard ? guard->want : 0;
    // a guard covers ONE kernel: the two streaming kernels without the long-row pass
    if (guard && (sel.variant == GESPMM_VARIANT_PARREDUCE || sel.variant == GESPMM_VARIANT_NAIVE || sel.geo.slab_blocked ||
                  sel.geo.split_long_rows))
        return gespmm::kNotGuardable;
    if (guard && guard->word == nullptr) return 0;  // dry run (auto_plan.cpp): would this call be one guardable kernel? nothing is launched

    hipError_t e;
    a.rpw = sel.geo.rows_per_group;
    if (sel.variant == GESPMM_VARIANT_PARREDUCE) e = gespmm::launch_spmm_parreduce(a, sel.geo, st);
    else if (sel.variant == GESPMM_VARIANT_NAIVE)
        e = gespmm::launch_spmm_naive(a, sel.geo, st);
    else if (sel.geo.slab_blocked) {
        // rows per lane group and launch: 2 measured best on reddit-like at N = 64..256 (hub rows make
        // 8-row tasks a long tail: 45 % average occupancy in the PMC run), profiles/r01/slab_task_size.log
        a.rpw = (cfg && cfg->rows_per_wave > 0) ? cfg->rows_per_wave : 2;
        e = gespmm::launch_spmm_slabblocked(a, sel.geo, ws, (size_t)(ws_bytes > 0 ? ws_bytes : 0), st);
    } else {
        bool seg = sel.geo.segmented;
        if (flags & gespmm::kFlagBatchStream) seg = false;
        if ((flags & gespmm::kFlagSegStream) && !sel.geo.split_long_rows) seg = true;
        // operands that are only 4- or 8-byte aligned resolve to a narrower vector; two strips of < 4 floats have no
        // segmented-stream instantiation: a plan then runs its wavefront task table (any N and alignment stay legal)
        if (seg && pl && sel.geo.strips == 2 && sel.geo.vec < 4) seg = false;
        if (seg) e = gespmm::launch_spmm_segstream(a, sel.geo, st);
        else {
            a.rpw = sel.geo.rows_per_wave;
            a.long_row = sel.geo.split_long_rows ? sel.geo.long_row_threshold : 0;
            if (sel.geo.split_long_rows)
                e = gespmm::launch_spmm_stream_with_longrows(a, sel.geo, nnz, ws, (size_t)(ws_bytes > 0 ? ws_bytes : 0), st);
            else
                e = gespmm::launch_spmm_stream(a, sel.geo, st);
        }
    }
    return (int)e;
}

}  // namespace gespmm

namespace {
int run_spmm(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C, int64_t M,
             int64_t K, int64_t N, int64_t nnz, int variant, const gespmm_launch_cfg* cfg, int reduce, float empty,
             void* stream, void* ws = nullptr, int64_t ws_bytes = 0) {
    return gespmm::run_spmm(rowptr, colind, val, B, C, M, K, N, nnz, variant, cfg, reduce, empty, stream, ws, ws_bytes,
                            nullptr);
}
}  // namespace

extern "C" {

const char* gespmm_version(void) {
    static char buf[64];
    if (!buf[0]) snprintf(buf, sizeof buf, "gespmm %d.%d (gfx950)", GESPMM_VERSION_MAJOR, GESPMM_VERSION_MINOR);
    return buf;
}

const char* gespmm_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case GESPMM_EINVAL: return "gespmm: invalid argument";
        case GESPMM_EALIGN: return "gespmm: pointer not 4-byte aligned";
        case GESPMM_ERANGE: return "gespmm: size exceeds int32 CSR addressing";
        case GESPMM_EIO: return "gespmm: file not found or unreadable";
        case GESPMM_EFORMAT: return "gespmm: could not process Matrix Market banner or size line";
        case GESPMM_ENOMEM: return "gespmm: host allocation failed";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "gespmm: unknown error code";
}

int gespmm_csr_spmm_f32(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C,
                        int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, void* stream) {
    if (gespmm::auto_plan_enabled()) {  // (gespmm_set_auto_plan: off by default — one relaxed load)
        int rc = 0;
        if (check_common(rowptr, colind, val, B, C, M, K, N, nnz) == 0 && variant >= GESPMM_VARIANT_AUTO && variant < GESPMM_NUM_VARIANTS &&
            variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE &&  // (a plan runs the CRC family)
            gespmm::auto_plan_try(rowptr, colind, val, B, C, M, K, N, nnz, variant, gespmm::kReduceSum, 0.0f, stream, &rc))
            return rc;
    }
    return run_spmm(rowptr, colind, val, B, C, M, K, N, nnz, variant, nullptr, gespmm::kReduceSum, 0.0f, stream);
}

int gespmm_csr_spmm_f32_cfg(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B,
                            float* C, int64_t M, int64_t K, int64_t N, int64_t nnz, int variant,
                            const gespmm_launch_cfg* cfg, void* stream) {
    return run_spmm(rowptr, colind, val, B, C, M, K, N, nnz, variant, cfg, gespmm::kReduceSum, 0.0f, stream);
}

int gespmm_csr_spmm_max_f32(const int32_t* rowptr, const int32_t* colind, const float* B, float* C, int64_t M,
                            int64_t K, int64_t N, int64_t nnz, float empty_value, int variant, void* stream) {
    if (gespmm::auto_plan_enabled() && variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE) {
        int rc = 0;
        if (check_common(rowptr, colind, nullptr, B, C, M, K, N, nnz) == 0 && variant >= GESPMM_VARIANT_AUTO && variant < GESPMM_NUM_VARIANTS &&
            gespmm::auto_plan_try(rowptr, colind, nullptr, B, C, M, K, N, nnz, variant, gespmm::kReduceMax, empty_value, stream, &rc))
            return rc;
    }
    return run_spmm(rowptr, colind, nullptr, B, C, M, K, N, nnz, variant, nullptr, gespmm::kReduceMax, empty_value,
                    stream);
}

int64_t gespmm_csr_spmm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant,
                                        const gespmm_launch_cfg* cfg) {
    if (M < 0 || K < 0 || N < 0 || nnz < -1) return GESPMM_EINVAL;
    if (variant < GESPMM_VARIANT_AUTO || variant >= GESPMM_NUM_VARIANTS) return GESPMM_EINVAL;
    if (M == 0 || N == 0) return 0;
    // the geometry depends on the alignment of B and C, unknown here: take the largest need
    size_t need = 0;
    for (int max_vec = 1; max_vec <= 4; max_vec *= 2) {
        if (N % max_vec != 0) break;
        gespmm::Selection sel;
        if (gespmm::resolve_geometry(M, K, N, nnz, variant, max_vec, cfg ? cfg->vec : 0, cfg ? cfg->strips : 0,
                                     cfg ? cfg->group : 0, cfg ? cfg->rows_per_wave : 0, cfg ? cfg->slab_rows : 0,
                                     cfg ? cfg->flags : 0, &sel) != 0)
            return GESPMM_EINVAL;
        size_t b = 0;
        if (sel.variant == GESPMM_VARIANT_PARREDUCE || sel.variant == GESPMM_VARIANT_NAIVE) b = 0;
        else if (sel.geo.slab_blocked) b = gespmm::slabblocked_workspace_bytes(M, sel.geo);
        else if (sel.geo.split_long_rows) b = gespmm::longrows_workspace_bytes(nnz, N, sel.geo.long_row_threshold);
        if (b > need) need = b;
    }
    return (int64_t)need;
}

int gespmm_csr_spmm_f32_ws(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C,
                           int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, const gespmm_launch_cfg* cfg,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (workspace_bytes < 0 || (workspace_bytes > 0 && workspace == nullptr)) return GESPMM_EINVAL;
    if (cfg == nullptr && gespmm::auto_plan_enabled()) {  // (no launch knobs: the stateless call with the caller's scratch — what the torch op makes)
        int rc = 0;
        if (check_common(rowptr, colind, val, B, C, M, K, N, nnz) == 0 && variant >= GESPMM_VARIANT_AUTO && variant < GESPMM_NUM_VARIANTS &&
            variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE &&
            gespmm::auto_plan_try(rowptr, colind, val, B, C, M, K, N, nnz, variant, gespmm::kReduceSum, 0.0f, stream, &rc))
            return rc;
    }
    return run_spmm(rowptr, colind, val, B, C, M, K, N, nnz, variant, cfg, gespmm::kReduceSum, 0.0f, stream, workspace,
                    workspace_bytes);
}

int gespmm_select_variant(int64_t M, int64_t nnz, int64_t N) { return gespmm::auto_variant(M, nnz, N); }

int gespmm_describe_launch(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, const gespmm_launch_cfg* cfg,
                           char* out, int64_t capacity) {
    if (!out || capacity <= 0 || M < 0 || K < 0 || N < 0 || nnz < -1) return GESPMM_EINVAL;
    if (variant < GESPMM_VARIANT_AUTO || variant >= GESPMM_NUM_VARIANTS) return GESPMM_EINVAL;
    int max_vec = 4;
    while (max_vec > 1 && (N % max_vec) != 0) max_vec >>= 1;
    gespmm::Selection sel;
    const int flags = cfg ? cfg->flags : 0;
    const int rc = gespmm::resolve_geometry(M, K, N, nnz, variant, max_vec, cfg ? cfg->vec : 0, cfg ? cfg->strips : 0,
                                            cfg ? cfg->group : 0, cfg ? cfg->rows_per_wave : 0, cfg ? cfg->slab_rows : 0,
                                            flags, &sel);
    if (rc != 0) return rc;
    const gespmm::Geometry& g = sel.geo;
    const char* idx = g.idx64 ? "idx64" : "idx32";
    int n;
    if (sel.variant == GESPMM_VARIANT_PARREDUCE)
        n = snprintf(out, (size_t)capacity, "variant=5 kernel=parallel-reduction W=%d %s", g.group, idx);
    else if (sel.variant == GESPMM_VARIANT_NAIVE)
        n = snprintf(out, (size_t)capacity, "variant=0 kernel=naive V=%d S=%d W=%d %s", g.vec, g.strips, g.group, idx);
    else if (g.slab_blocked)
        n = snprintf(out, (size_t)capacity, "variant=%d kernel=slab-blocked V=%d S=%d W=%d slab_rows=%d slabs=%lld %s",
                     sel.variant, g.vec, g.strips, g.group, g.slab_rows,
                     (long long)((K + g.slab_rows - 1) / g.slab_rows), idx);
    else {
        bool seg = g.segmented;
        if (flags & gespmm::kFlagBatchStream) seg = false;
        if ((flags & gespmm::kFlagSegStream) && !g.split_long_rows) seg = true;
        char tail[64] = "";
        if (!seg && g.split_long_rows) snprintf(tail, sizeof tail, " long_rows>%d chunk=%d", g.long_row_threshold, gespmm::kLongRowChunk);
        const char* st = (g.sc1_store || (flags & gespmm::kFlagSc1Store)) ? " c_stores=sc1" : "";
        if (seg)
            n = snprintf(out, (size_t)capacity, "variant=%d kernel=segmented-stream V=%d S=%d W=%d rows_per_group=%d %s%s",
                         sel.variant, g.vec, g.strips, g.group, g.rows_per_group, idx, st);
        else
            n = snprintf(out, (size_t)capacity, "variant=%d kernel=batch-stream V=%d S=%d W=%d rows_per_wave=%d %s%s%s",
                         sel.variant, g.vec, g.strips, g.group, g.rows_per_wave, idx, tail, st);
    }
    if (n < 0) return GESPMM_EINVAL;
    return n < capacity ? n : (int)capacity - 1;
}

// DGL hands over neither nnz nor the number of source nodes. For large graphs the 4-byte read of
// indptr[m] (a stream synchronisation — the patch's CustomCsrmm synchronises the stream right after the
// kernel anyway, binary_reduce_sum.cu:358) buys the dense-graph and long-row paths: reddit-shaped,
// N=128: 8.3 -> 4.2 ms. The number of source nodes is taken as m for the slab count only (columns beyond
// it fall into the last slab — same result), offsets into B stay 64-bit. Not on a capturing stream.
// Rows from which the DGL entry points read nnz back (a stream synchronisation); < 0 = never. Process-wide, set
// once at start-up by the integrator (gespmm_dgl_set_readback_rows).
static std::atomic<int64_t> g_dgl_readback_rows{1 << 15};

int gespmm_dgl_set_readback_rows(int64_t rows) {
    g_dgl_readback_rows.store(rows, std::memory_order_relaxed);
    return 0;
}

static int dgl_csrmm(int m, int n, const int32_t* indptr, const int32_t* indices, const float* B, float* C, int reduce,
                     float empty, void* stream) {
    int64_t nnz = -1, K = 0x7fffffffLL;
    gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (gespmm::auto_plan_enabled() && m > 0 && n > 0 && indptr && indices && B && C) {
        // (the fingerprint's read-back is the synchronisation this entry point performs anyway; K and nnz come out of it)
        int rc = 0;
        if (gespmm::auto_plan_try(indptr, indices, nullptr, B, C, m, 0, n, -1, GESPMM_VARIANT_AUTO, reduce, empty, stream, &rc)) return rc;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    const int64_t rb_rows = g_dgl_readback_rows.load(std::memory_order_relaxed);
    if (!capturing && rb_rows >= 0 && m >= rb_rows && indptr) {
        int32_t last = -1;
        if (hipMemcpyAsync(&last, indptr + m, sizeof last, hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess && last >= 0) {
            nnz = last;
            K = m;
            cfg.flags = GESPMM_FLAG_FORCE_IDX64;
        }
    }
    (void)hipGetLastError();
    return run_spmm(indptr, indices, nullptr, B, C, m, K, n, nnz, GESPMM_VARIANT_AUTO, &cfg, reduce, empty, stream);
}

int gespmm_dgl_csrmm_sum_f32(int m, int n, const int32_t* indptr, const int32_t* indices, const float* B, float* C,
                             void* stream) {
    return dgl_csrmm(m, n, indptr, indices, B, C, gespmm::kReduceSum, 0.0f, stream);
}

int gespmm_dgl_csrmm_max_f32(int m, int n, const int32_t* indptr, const int32_t* indices, const float* B, float* C,
                             void* stream) {
    return dgl_csrmm(m, n, indptr, indices, B, C, gespmm::kReduceMax, -10000.0f, stream);
}

int gespmm_sddmm_coo_f32(const int32_t* rowind, const int32_t* colind, const float* D1, const float* D2, float* out,
                         int64_t nnz, int64_t N, void* stream) {
    if (nnz < 0 || N < 0) return GESPMM_EINVAL;
    if (nnz > 0x7fffffffLL || N > 0x7fffffffLL / 4) return GESPMM_ERANGE;
    if (nnz == 0) return 0;
    if (!rowind || !colind || !out || (N > 0 && (!D1 || !D2))) return GESPMM_EINVAL;
    if (!aligned_to(rowind, 4) || !aligned_to(colind, 4) || !aligned_to(D1, 4) || !aligned_to(D2, 4) ||
        !aligned_to(out, 4))
        return GESPMM_EALIGN;
    return (int)gespmm::launch_sddmm(rowind, false, colind, D1, D2, out, 0, nnz, N, 0,
                                     reinterpret_cast<hipStream_t>(stream));
}

int gespmm_sddmm_csr_f32(const int32_t* rowptr, const int32_t* colind, const float* D1, const float* D2, float* out,
                         int64_t M, int64_t nnz, int64_t N, void* stream) {
    if (M < 0 || nnz < 0 || N < 0) return GESPMM_EINVAL;
    if (M > 0x7fffffffLL - 1 || nnz > 0x7fffffffLL || N > 0x7fffffffLL / 4) return GESPMM_ERANGE;
    if (nnz == 0) return 0;
    if (!rowptr || !colind || !out || (N > 0 && (!D1 || !D2))) return GESPMM_EINVAL;
    if (!aligned_to(rowptr, 4) || !aligned_to(colind, 4) || !aligned_to(D1, 4) || !aligned_to(D2, 4) ||
        !aligned_to(out, 4))
        return GESPMM_EALIGN;
    return (int)gespmm::launch_sddmm(rowptr, true, colind, D1, D2, out, M, nnz, N, 0,
                                     reinterpret_cast<hipStream_t>(stream));
}

int gespmm_baseline_atomic_scatter_f32(const int32_t* rowptr, const int32_t* colind, const float* in, float* out,
                                       int64_t M, int64_t K, int64_t N, int64_t nnz, void* stream) {
    if (M < 0 || K < 0 || N < 0 || nnz < 0) return GESPMM_EINVAL;
    if (M > 0x7fffffffLL - 1 || K > 0x7fffffffLL || N > 0x7fffffffLL / 4 || nnz > 0x7fffffffLL) return GESPMM_ERANGE;
    if (K == 0 || N == 0) return 0;
    if (!out || (nnz > 0 && (!rowptr || !colind || !in))) return GESPMM_EINVAL;
    return (int)gespmm::launch_atomic_scatter(rowptr, colind, in, out, M, K, N, nnz,
                                              reinterpret_cast<hipStream_t>(stream));
}

int gespmm_baseline_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
    if (n < 0) return GESPMM_EINVAL;
    if (n == 0) return 0;
    if (!src || !dst) return GESPMM_EINVAL;
    if (!aligned_to(src, 4) || !aligned_to(dst, 4)) return GESPMM_EALIGN;
    return (int)gespmm::launch_copy(src, dst, n, reinterpret_cast<hipStream_t>(stream));
}

int64_t gespmm_csr2csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
    if (M < 0 || K < 0 || nnz < 0) return GESPMM_EINVAL;
    return gespmm::csr2csc_workspace_bytes(M, K, nnz);
}

int gespmm_csr2csc_f32(const int32_t* rowptr, const int32_t* colind, const float* csr_val, int32_t* colptr,
                       int32_t* rowind, float* csc_val, int64_t M, int64_t K, int64_t nnz, void* workspace,
                       void* stream) {
    if (M < 0 || K < 0 || nnz < 0) return GESPMM_EINVAL;
    if (M > 0x7fffffffLL - 1 || K > 0x7fffffffLL - 1 || nnz > 0x7fffffffLL) return GESPMM_ERANGE;
    if (!rowptr || !colptr) return GESPMM_EINVAL;
    if (nnz > 0 && (!colind || !rowind || !workspace)) return GESPMM_EINVAL;
    if ((csr_val == nullptr) != (csc_val == nullptr)) return GESPMM_EINVAL;
    return (int)gespmm::launch_csr2csc(rowptr, colind, csr_val, colptr, rowind, csc_val, M, K, nnz, workspace,
                                       reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
