/*
 * gespmm.h — C ABI of libgespmm.so, the MI355X (gfx950) GE-SpMM hot path.
 *
 * This is the drop-in boundary for the ONE path this repository accelerates:
 * CSR x dense row-product SpMM (fp32 values, int32 indices), plus the SDDMM and
 * CSR->CSC helpers the reference's PyTorch op exposes next to it.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * the reference repository hgyhungry/ge-spmm):
 *
 *   gespmm_csr_spmm_f32      <- spmmWrapper()                 spmm_test.cu:456-492
 *                               spmm_cuda()                   pytorch-custom/spmm_kernel.cu:425-458
 *                               spmm_cuda_no_edge_value()     pytorch-custom/spmm_kernel.cu:175-207
 *                               XTopoCsrmm<float>()           dgl-custom/binary_reduce_sum.cu:310-335
 *   gespmm_csr_spmm_max_f32  <- XTopoCsrmmmax<float>()        dgl-custom/binary_reduce_max.cu:182-207
 *   gespmm_dgl_csrmm_{sum,max}_f32  <- the same two, with exactly their argument lists
 *   gespmm_csr_spmm_workspace_bytes / gespmm_csr_spmm_f32_ws  <- (new) caller-owned scratch, no reference counterpart
 *   gespmm_select_variant    <- the N-based 3-way dispatch    pytorch-custom/spmm_kernel.cu:186-206,437-457
 *   gespmm_sddmm_coo_f32     <- sddmm_cuda_coo()              pytorch-custom/sddmm.cu:427-457
 *   gespmm_sddmm_csr_f32     <- sddmm_cuda_csr()              pytorch-custom/sddmm.cu:459-484
 *   gespmm_csr2csc_f32       <- csr2csc_cuda()/csr2cscKernel  pytorch-custom/spmm_kernel.cu:381-476
 *   gespmm_mtx_read[_cached] / _free  <- readMtx<float>()     util/util.hpp:286-333 (+ mmio.hpp:215,308)
 *   gespmm_coo_to_csr        <- inline COO->CSR               spmm_test.cu:557-581
 *   gespmm_row_partition     <- (new; north_star multi-GPU)   no reference counterpart
 *   gespmm_plan_*            <- (new) analysis stage in front of repeated launches; no reference counterpart
 *   gespmm_init              <- the 200 empty warm-up launches spmm_test.cu:720-721 (same role: start-up cost outside the timed loop)
 *   gespmm_set_auto_plan / gespmm_auto_plan_* <- (new) plans for callers that keep no state: spmmWrapper spmm_test.cu:456-492,
 *                               spmm_cuda spmm_kernel.cu:425-458, CustomCsrmm dgl-custom/binary_reduce_sum.cu:338-360
 *   gespmm_cluster_rows / gespmm_simulate_l2_hits <- (new) the plan's host-side row clustering and its L2 model
 *   gespmm_baseline_atomic_scatter_f32 <- Gunrock app's edge map  gunrock-test/app/spmm/spmm_enactor.cuh:92-105
 *   gespmm_baseline_copy_f32 <- (new) streaming-copy yardstick for the roofline record; no reference counterpart
 *
 * Conventions (all device entry points):
 *   - every pointer is a DEVICE pointer owned by the caller, except where a
 *     parameter is documented as host memory (the loader / partitioner);
 *   - dense matrices are row-major with leading dimension N (B is K x N, C is M x N);
 *   - C is fully overwritten (alpha = 1, beta = 0, no accumulate, no pre-zeroing);
 *   - the call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the legacy default stream, which is what the reference launches on);
 *   - the return value is a hipError_t cast to int (0 = success) or one of the
 *     negative GESPMM_E* codes below; the library never calls exit();
 *   - re-entrant; the only global state is cached, immutable device properties and one memory
 *     pool per device for stream-ordered temporaries (created on first use);
 *   - the device that owns `stream` and the pointers is the calling thread's current device;
 *   - index RANGES are the caller's responsibility, as in the reference: rowptr must be non-decreasing with
 *     rowptr[M] = nnz and every colind entry must lie in [0, K). The device kernels do not check (a check would cost
 *     a pass over the matrix per call); the host-side entry points (loader, COO->CSR, plans) do.
 *
 * There is NO CPU fallback: without a HIP device every compute entry point
 * returns the HIP error (hipErrorNoDevice = 100).
 */
#ifndef GESPMM_H_
#define GESPMM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GESPMM_VERSION_MAJOR 0
#define GESPMM_VERSION_MINOR 3

/* Negative return codes (positive values are hipError_t). */
#define GESPMM_EINVAL   (-1)  /* bad argument (null pointer, negative size, unknown variant) */
#define GESPMM_EALIGN   (-2)  /* pointer not 4-byte aligned */
#define GESPMM_ERANGE   (-3)  /* size exceeds what int32 CSR indices can address */
#define GESPMM_EIO      (-4)  /* loader: file not found / unreadable */
#define GESPMM_EFORMAT  (-5)  /* loader: bad MatrixMarket banner or size line */
#define GESPMM_ENOMEM   (-6)  /* host allocation failed */

/*
 * Kernel variants. Numbers 0..4 keep the reference's `method` numbering
 * (spmm_test.cu:456-492): 0 naive, 1 CRC, 2/3/4 CRC+CWM with coarsening factor
 * 2/4/8. 5 is the parallel-reduction variant (lanes over nnz) that north_star
 * asks for and the reference only has in dgSPARSE. -1 lets the library choose.
 *
 * Variants 0-4 accumulate each output element in ONE fp32 register in ascending
 * CSR position with one fused multiply-add per non-zero (the arithmetic the
 * reference's device kernels perform); they are bit-identical to each other.
 * Variant 5 changes the summation order and is tolerance-checked.
 *
 * Load balance: in matrices with >= 2^23 non-zeros (or >= 2^20 at mean degree >= 8), rows longer than
 * max(2048, 32 x mean degree) entries (RMAT hubs) are skipped by the main kernel and computed in
 * 2048-entry CHUNKS spread over the whole chip: a workgroup sums one chunk (its lane groups take the
 * chunk's 64-entry tiles round-robin, partial rows added in fixed group order), a combine kernel adds
 * a row's chunk partials in chunk order — bit-reproducible from run to run, but a re-association of
 * the strict chain (within the 1e-4 tolerance, not bit-for-bit). GESPMM_FLAG_STRICT_ORDER (through
 * gespmm_csr_spmm_f32_cfg) keeps every row a strict chain. A gespmm_plan decides from the longest row
 * it actually saw instead of these size rules.
 */
#define GESPMM_VARIANT_AUTO       (-1)
#define GESPMM_VARIANT_NAIVE        0
#define GESPMM_VARIANT_CRC          1
#define GESPMM_VARIANT_CRC_CWM2     2
#define GESPMM_VARIANT_CRC_CWM4     3
#define GESPMM_VARIANT_CRC_CWM8     4
#define GESPMM_VARIANT_PARREDUCE    5
#define GESPMM_NUM_VARIANTS         6

const char* gespmm_version(void);
/* Human-readable string for any return code of this library. */
const char* gespmm_error_string(int code);

/*
 * C[M x N] = A[M x K] * B[K x N], A in CSR (rowptr[M+1], colind[nnz], val[nnz]).
 * val == NULL means A == 1 on its pattern (the "topo"/no_edge_value kernels).
 * nnz may be passed as -1 when unknown (it feeds the variant heuristic and sizes the
 * long-row pass's temporary: with -1 every row keeps the strict chain).
 * Temporaries (dense-graph cache blocking, long-row pass) are stream-ordered allocations
 * from a pool the library owns; on a stream under capture both paths are switched off, so
 * a captured call never allocates.
 */
int gespmm_csr_spmm_f32(const int32_t* rowptr, const int32_t* colind, const float* val,
                        const float* B, float* C,
                        int64_t M, int64_t K, int64_t N, int64_t nnz,
                        int variant, void* stream);

/*
 * Same product with the `max` reducer over the row's neighbours (unweighted):
 * C[r,c] = max_p B[colind[p], c]; rows without non-zeros yield `empty_value`
 * (the reference hard-codes -10000, binary_reduce_max.cu:22-24).
 */
int gespmm_csr_spmm_max_f32(const int32_t* rowptr, const int32_t* colind,
                            const float* B, float* C,
                            int64_t M, int64_t K, int64_t N, int64_t nnz,
                            float empty_value, int variant, void* stream);

/* The variant `GESPMM_VARIANT_AUTO` resolves to for this shape (pure host logic). */
int gespmm_select_variant(int64_t M, int64_t nnz, int64_t N);

/*
 * Explicit launch geometry, for tuning sweeps and tests. Fields set to 0 mean
 * "library default for this variant and N".
 *   vec      floats per lane per strip (1, 2 or 4) — contiguous, one vector load
 *   strips   strips per lane (1 or 2); coarsening factor = vec * strips
 *   group    lanes cooperating on one row (4..64, power of two); a 64-lane
 *            wavefront therefore carries 64/group rows
 *   rows_per_wave  consecutive rows one lane group streams through (1..32). With
 *            GESPMM_FLAG_BATCH_STREAM: rows per wavefront (rounded to a multiple of 64/group)
 *   flags    GESPMM_FLAG_* bits
 */
typedef struct gespmm_launch_cfg {
    int32_t vec;
    int32_t strips;
    int32_t group;
    int32_t rows_per_wave;
    int32_t slab_rows;   /* cache-blocked path: B rows per column slab (0 = ~6 MB worth) */
    int32_t flags;
} gespmm_launch_cfg;

#define GESPMM_FLAG_NO_XCD_REMAP   0x1  /* plain blockIdx -> row-block mapping */
#define GESPMM_FLAG_NT_STORE       0x2  /* non-temporal stores of C */
#define GESPMM_FLAG_SC1_STORE      0x8000 /* streaming kernels: C stored with system scope (written through instead of kept in the
                                            XCD's L2). Neutral where B exceeds the L2s; with B L2-resident and C not, +20 % when the
                                            reuse of B is skewed, -13 % when it is uniform (profiles/r02/l2_resident_store_scope.log).
                                            The cache-blocked path always stores this way */
#define GESPMM_FLAG_FORCE_IDX64    0x4  /* 64-bit B offsets even when K*N*4 < 2^32 */
#define GESPMM_FLAG_SHALLOW_UNROLL 0x10 /* gather 4 instead of 8 B rows per step (fewer VGPRs) */
#define GESPMM_FLAG_BATCH_STREAM   0x20 /* force the batch-stream kernel (rows walked 64/group at a time) */
#define GESPMM_FLAG_STRICT_ORDER   0x100 /* never split long rows: every row is one strict CSR-order chain */
#define GESPMM_FLAG_SPLIT_LONG_ROWS 0x200 /* run the long-row pass regardless of matrix size */
#define GESPMM_FLAG_SLAB_BLOCKED   0x400 /* force the cache-blocked path (one launch per column slab of B) */
#define GESPMM_FLAG_NO_SLAB_BLOCKED 0x800 /* never use it */
#define GESPMM_FLAG_REUSE_SPLIT     0x2000 /* gespmm_csr_spmm_f32_ws only: `workspace` already holds the split points an earlier call
                                             wrote for the SAME rowptr/colind/N/cfg (cache-blocked path): skip the scan. The caller
                                             vouches that the graph did not change; ignored on every other path */
#define GESPMM_FLAG_ALLOW_REASSOCIATION 0x1000 /* AUTO may pick the parallel-reduction variant (N <= 16, dense rows) */
#define GESPMM_FLAG_SEG_STREAM     0x80 /* force the segmented-stream kernel (AUTO takes it for short rows over an L2-resident B and
                                           for mean degree <= 3; ignored where long rows are split) */

int gespmm_csr_spmm_f32_cfg(const int32_t* rowptr, const int32_t* colind, const float* val,
                            const float* B, float* C,
                            int64_t M, int64_t K, int64_t N, int64_t nnz,
                            int variant, const gespmm_launch_cfg* cfg, void* stream);

/*
 * The same product with a CALLER-PROVIDED temporary (cuSPARSE/rocSPARSE style: query, allocate,
 * run). Two paths of the library need scratch memory — column-slab cache blocking for dense
 * graphs (per-row split points) and the chunked long-row pass (partial rows). The plain entry
 * points above take it stream-ordered from a pool the library owns; this one uses
 * `workspace` (device memory, 16-byte aligned, at least gespmm_csr_spmm_workspace_bytes() for
 * the same shape/variant/cfg) and never allocates — which also keeps both paths available on a
 * stream that is being captured into a HIP graph. workspace_bytes == 0 behaves like
 * gespmm_csr_spmm_f32_cfg. A framework binding passes memory from the framework's own
 * caching allocator here.
 */
int64_t gespmm_csr_spmm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant,
                                        const gespmm_launch_cfg* cfg /* may be NULL */);
int gespmm_csr_spmm_f32_ws(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C,
                           int64_t M, int64_t K, int64_t N, int64_t nnz, int variant,
                           const gespmm_launch_cfg* cfg /* may be NULL */, void* workspace, int64_t workspace_bytes,
                           void* stream);

/*
 * What a call with these arguments would launch, as one line of text (host-only; no device work): kernel
 * family, vector width V, strips S, lanes per row W, task size, and for the two special paths the slab /
 * long-row parameters — e.g. "variant=3 kernel=batch-stream V=4 S=1 W=32 rows_per_wave=4 idx32" or
 * "variant=3 kernel=slab-blocked V=4 S=1 W=32 slab_rows=12288 slabs=19 idx32". B and C are assumed
 * 16-byte aligned. Returns the length written (excluding the NUL), or a negative code.
 */
int gespmm_describe_launch(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant,
                           const gespmm_launch_cfg* cfg /* may be NULL */, char* out, int64_t capacity);


/*
 * The DGL kernel patch's entry points (dgl-custom/binary_reduce_sum.cu:310-335 XTopoCsrmm<float>,
 * binary_reduce_max.cu:182-207 XTopoCsrmmmax<float>) with their argument list — (m, n, indptr,
 * indices, B, C) plus the stream DGL keeps in its RuntimeConfig: C[m x n] = sum / max over the
 * row's neighbours of B[neighbour, :], A == 1 on the CSR pattern. DGL passes neither the number
 * of source nodes nor nnz at that point: 64-bit offsets into B are used; for graphs of 32768 rows
 * or more nnz is read back from indptr[m] (a stream synchronisation, as the patch does after its
 * kernel) so that dense graphs and hub rows get their dedicated paths, smaller graphs keep the
 * strict CSR-order chain for every row. Rows without neighbours give 0 (sum) or -10000 (max: the patch's
 * max_init, binary_reduce_max.cu:22-24). Returns 0 or an error code as above (the patch returns
 * its cudaError the same way).
 * The read-back is a policy, not part of the product: gespmm_dgl_set_readback_rows(rows) moves the threshold
 * (process-wide; rows < 0 = never synchronise: every call is then fully asynchronous and every row a strict chain;
 * 0 = always). On a stream under capture the entry points never read back, whatever the threshold.
 */
int gespmm_dgl_set_readback_rows(int64_t rows);
int gespmm_dgl_csrmm_sum_f32(int m, int n, const int32_t* indptr, const int32_t* indices, const float* B, float* C,
                             void* stream);
int gespmm_dgl_csrmm_max_f32(int m, int n, const int32_t* indptr, const int32_t* indices, const float* B, float* C,
                             void* stream);

/*
 * SDDMM: out[e] = sum_j D1[row(e), j] * D2[col(e), j], e in pattern order.
 * COO: row(e) = rowind[e].  CSR: row(e) = the row whose [rowptr[r], rowptr[r+1]) holds e.
 * D1 is M x N, D2 is K x N, row-major; out has nnz floats.
 */
int gespmm_sddmm_coo_f32(const int32_t* rowind, const int32_t* colind,
                         const float* D1, const float* D2, float* out,
                         int64_t nnz, int64_t N, void* stream);
int gespmm_sddmm_csr_f32(const int32_t* rowptr, const int32_t* colind,
                         const float* D1, const float* D2, float* out,
                         int64_t M, int64_t nnz, int64_t N, void* stream);

/*
 * CSR (M x K) -> CSC on the device: fills colptr[K+1], rowind[nnz] and, when
 * csr_val != NULL, csc_val[nnz]. Entries inside one column keep ascending row
 * order (stable), which is what makes backward SpMM on the CSC arrays
 * deterministic. `workspace` must hold gespmm_csr2csc_workspace_bytes() bytes.
 */
int64_t gespmm_csr2csc_workspace_bytes(int64_t M, int64_t K, int64_t nnz);
int gespmm_csr2csc_f32(const int32_t* rowptr, const int32_t* colind, const float* csr_val,
                       int32_t* colptr, int32_t* rowind, float* csc_val,
                       int64_t M, int64_t K, int64_t nnz,
                       void* workspace, void* stream);

/*
 * ------------------------------------------------------------------ plans (analysis stage)
 *
 * Repeated products with ONE sparse matrix (200 timed launches in spmm_test.cu:754-762; every layer of every
 * epoch in gcn_custom.py:118-143) can be prepared once. The reference has no such stage — its kernels walk the
 * rows in storage order on every launch; vendor SpMMs have one (rocsparse_spmm_stage_preprocess). A plan
 *   - reads the matrix ONCE, on the device (validation, clustering, L2 model, task cutting are device passes;
 *     gespmm_plan_create synchronises `stream` a handful of times; GESPMM_PLAN_ANALYSIS_HOST keeps the host form),
 *   - decides the long-row pass from the longest row it actually saw,
 *   - keeps the scratch of the cache-blocked path (dense graphs) so its split scan runs once,
 *   - and, for sparse graphs whose B exceeds the L2s, keeps a ROW-CLUSTERED copy of the matrix plus a task
 *     table with an equal non-zero budget per wavefront: rows that share neighbours are processed next to each
 *     other, so the B rows they share come from the XCD's L2 instead of crossing the fabric again.
 * The order in which rows are PROCESSED is the only thing a plan changes: each row is still one fp32 chain in
 * its own CSR order written to its own row of C, so gespmm_plan_spmm_f32 returns the same bits as
 * gespmm_csr_spmm_f32: the long-row pass (a re-association of rows beyond 32x the mean degree / 2048 entries) runs under a
 * plan exactly where the plain call runs it — matrices of >= 2^23 non-zeros, or >= 2^20 with mean degree >= 8 — and, because
 * the plan has SEEN the longest row, only when such a row exists. GESPMM_FLAG_SPLIT_LONG_ROWS / _STRICT_ORDER in
 * gespmm_plan_options.flags force it on / off (as they do for the plain call).
 *
 * rowptr / colind / val are DEVICE pointers. A plan that keeps the storage order (small or dense matrices, or
 * reorder = GESPMM_PLAN_NO_REORDER) keeps referring to them, so they must outlive the plan; a clustered plan owns
 * copies. After changing the VALUES call gespmm_plan_set_values; a changed pattern needs a new plan.
 * One plan serves one stream at a time.
 */
typedef struct gespmm_plan gespmm_plan;

#define GESPMM_PLAN_REORDER_AUTO 0  /* cluster when B exceeds the L2s and the model of the L2s predicts a gain */
#define GESPMM_PLAN_REORDER      1  /* always cluster (streaming-kernel family only) */
#define GESPMM_PLAN_NO_REORDER   2  /* keep the storage order */

typedef struct gespmm_plan_options {
    int32_t reorder;       /* GESPMM_PLAN_REORDER_* */
    int32_t task_entries;  /* work per wavefront task of a clustered plan in non-zeros, 0 = default for N */
    int32_t row_floor;     /* a row counts as at least this many non-zeros when tasks are cut (the kernel spends one
                              memory round trip per row pair however short the rows are), 0 = default, -1 = none */
    int32_t threads;       /* host threads for the clustering, 0 = all (the result does not depend on it) */
    int32_t flags;         /* GESPMM_FLAG_* applied to every launch (e.g. GESPMM_FLAG_STRICT_ORDER) */
    int32_t kernel;        /* GESPMM_PLAN_KERNEL_*: which kernel a clustered plan launches */
    int32_t analysis;      /* GESPMM_PLAN_ANALYSIS_*: where the clustering / L2 model / task cutting run (since 0.2) */
    int32_t expected_launches; /* products this plan is expected to serve, 0 = 200 (the reference's ITER, spmm_test.cu:714; its GCN runs 200
                              epochs, gcn_custom.py:134). reorder = AUTO weighs the analysis against them: clustering is skipped when the
                              estimated gain per launch x launches does not pay for its estimated time, and plans with launches x N < 100 000
                              cluster three levels deep instead of six (since 0.3, gespmm_plan_create_v2 only) */
} gespmm_plan_options;
/*
 * Plan options and versions. Every field's default is 0 and fields are only ever APPENDED. gespmm_plan_create is the
 * un-versioned symbol: it reads the SEVEN fields every header that shipped with it alone had (reorder .. analysis) and nothing
 * beyond them, whatever header the caller was built with. gespmm_plan_create_v2 takes sizeof(gespmm_plan_options) as the caller's compiler saw it (`opt_bytes`): fields the
 * caller does not have take their defaults, bytes this library does not know are ignored.
 */

#define GESPMM_PLAN_ANALYSIS_DEVICE 0  /* on the device (default): no copy of the matrix leaves HBM */
#define GESPMM_PLAN_ANALYSIS_HOST   1  /* round-2 path: the matrix is copied to the host and clustered there (same order) */

#define GESPMM_PLAN_KERNEL_AUTO     0  /* batch-stream kernel on the task table; segmented-stream for products-shaped rows */
#define GESPMM_PLAN_KERNEL_STREAM   1  /* batch-stream kernel on the task table */
#define GESPMM_PLAN_KERNEL_SEG_STREAM 3 /* segmented-stream kernel on a task table per lane group */
/* (values 2 and 4 belonged to two opt-in kernels of rounds 2-3 — LDS-staged task rows, task-outer — that never beat the
   streaming kernels; their sources and logs are under profiles/r02/experiments/. gespmm_plan_create answers GESPMM_EINVAL.) */
#define GESPMM_PLAN_KERNEL_STAGED 5     /* scalar-stream walk + the most used B rows of every block of 96 / 64 clustered rows staged in LDS
                                          (N = 128, 256, 512 or 1024 — 256-column tiles bound to XCDs beyond 256; sum reducer, device
                                          analysis; rows beyond 2048 entries go to the long-row pass). AUTO takes it for clustered
                                          matrices with mean degree >= 12 (N = 128) / >= 5 (wider) when >= 60 % (N = 128) / 42 % of
                                          the entries find their B row staged (csrc/plan_policy.cpp: hold-out audit) */
#define GESPMM_PLAN_KERNEL_RECORDS 6    /* padded-record kernel (since 0.3, round 6): 4 <= N <= 64, rows of <= 1024 entries, sum reducer,
                                          max(M, K) * N * 4 < 4 GB — the plan lays its copy of the matrix out as batches of 8-slot row pieces
                                          (one coalesced load per lane, all 8 gathers of a piece in flight at once); AUTO takes it for short
                                          rows at narrow widths; other launches of the plan fall back to the streaming kernels */

#define GESPMM_PLAN_KERNEL_STAGED_SLABS 7 /* (since 0.3, round 6) the staged-rows kernel over P ascending COLUMN ranges of the clustered matrix, one
                                          staging list per (block of rows, range) and one launch per range, the second and later ones continuing
                                          from the partial sums in C — for dense clustered matrices (a reddit-shaped community refers to ~100 000
                                          B rows; 160 LDS slots per block cover a seventh of its entries, per range two thirds). N = 128, sum
                                          reducer, rows with non-decreasing columns (range order == CSR order: the same bits), no slab of a row
                                          beyond 2048 entries — else the plan keeps its other kernels. AUTO takes it at mean degree >= 96 when
                                          >= 50 % of the entries find their B row staged */

int gespmm_plan_create(gespmm_plan** plan, const int32_t* rowptr, const int32_t* colind, const float* val /* may be NULL */,
                       int64_t M, int64_t K, int64_t nnz, int64_t N /* width the plan is tuned for */, int variant,
                       const gespmm_plan_options* opt /* may be NULL */, void* stream);
int gespmm_plan_create_v2(gespmm_plan** plan, const int32_t* rowptr, const int32_t* colind, const float* val, int64_t M,
                          int64_t K, int64_t nnz, int64_t N, int variant, const gespmm_plan_options* opt /* may be NULL */,
                          int64_t opt_bytes /* sizeof(gespmm_plan_options) in the caller */, void* stream);
/* C[M x N] = A * B through the plan; any N is legal (scratch and task size are tuned for the plan's N). */
int gespmm_plan_spmm_f32(gespmm_plan* plan, const float* B, float* C, int64_t N, void* stream);
/*
 * Kernel choice by MEASUREMENT instead of by rule: runs the candidates of a clustered plan (batch-stream, segmented-stream,
 * staged-rows where the width is served, and at N <= 64 the batch-stream kernel with 4 floats per lane) `reps` (0 = 3) times each on these operands, synchronously, and fixes the plan
 * on the fastest; C holds the product afterwards (every candidate gives the same bits). N must be the plan's width. A no-op
 * for storage-order plans and plans created with an explicit kernel. Not under stream capture. gespmm_plan_describe reports
 * the measured times.
 */
int gespmm_plan_tune(gespmm_plan* plan, const float* B, float* C, int64_t N, int32_t reps, void* stream);
/* max reducer (unweighted plans only), see gespmm_csr_spmm_max_f32 */
int gespmm_plan_spmm_max_f32(gespmm_plan* plan, const float* B, float* C, int64_t N, float empty_value, void* stream);
/* SDDMM on the plan's pattern, out[nnz] in the caller's CSR edge order (same bits as gespmm_sddmm_csr_f32); a clustered
 * plan whose order is modelled to hit L2 walks the edges in its own order (shared rows of D2 come from L2) and scatters the
 * results back; otherwise the call is gespmm_sddmm_csr_f32 on the arrays the plan was made from (keep them alive). */
int gespmm_plan_sddmm_f32(gespmm_plan* plan, const float* D1, const float* D2, float* out, int64_t N, void* stream);
/* New values on the unchanged pattern (val in the caller's CSR order, device memory; NULL = A == 1). */
int gespmm_plan_set_values(gespmm_plan* plan, const float* val, void* stream);
/* perm_host[i] = row processed at position i (HOST memory, M entries). Returns 1 if clustered, 0 if storage order. */
int gespmm_plan_get_order(const gespmm_plan* plan, int32_t* perm_host);
/* One line of text: order, cluster hierarchy, tasks, modelled L2 hit rate before -> after, analysis time, launch. */
int gespmm_plan_describe(const gespmm_plan* plan, char* out, int64_t capacity);
void gespmm_plan_destroy(gespmm_plan* plan);
/* The analysis stage keeps its scratch arena for the next plan — one per device, up to a limit of 1 GiB by default
 * (GESPMM_ARENA_CACHE_MB overrides the default; a products-sized analysis takes ~10 GB, freed when the plan is made unless the
 * limit says otherwise). gespmm_set_cached_memory_limit changes the limit (bytes; negative = default), gespmm_release_cached_memory
 * gives the kept arenas back. */
void gespmm_set_cached_memory_limit(int64_t bytes);
void gespmm_release_cached_memory(void);

/*
 * Warm-up (since 0.3). The FIRST plan a process builds pays ~29 ms on top of its analysis: the analysis kernels are loaded on first use
 * and the scratch arena is allocated (profiles/r06/plan_cold.log). A process that follows the reference's protocol — one process per
 * matrix, 200 launches (run_test.sh:5-11) — cannot amortise that, so the driver and the Python layer call gespmm_init before they time
 * anything: it builds and destroys plans on a small built-in matrix (every analysis pass, the staging tables, one product through each
 * kernel family) and reserves the arena a matrix of (rows_hint, nnz_hint) will ask for (0 / 0: nothing reserved; within the limit of
 * gespmm_set_cached_memory_limit). Idempotent per device; returns 0 or an error code. Until it (or a first plan) has run, a plan made
 * with reorder = AUTO adds the cold cost to its cost rule (gespmm_plan_policy_query.cold_start).
 */
int gespmm_init(int64_t rows_hint, int64_t nnz_hint, void* stream);
/* 1 if an analysis of a matrix of this shape could pay for itself inside `expected_launches` (0 = 200) products once the library is warm
 * (the cost rule with the most structure a probe can report), else 0: for such a matrix there is nothing to warm up for — a plan keeps
 * its storage order either way. Host only. The Python layer asks this before it spends gespmm_init's ~60 ms on a pubmed-sized graph. */
int gespmm_plan_wants_warmup(int64_t M, int64_t K, int64_t nnz, int64_t N, int32_t expected_launches);

/*
 * Plan reuse behind the STATELESS entry points (since 0.3; off by default). The reference's callers keep no state between products
 * (spmmWrapper spmm_test.cu:456-492, spmm_cuda spmm_kernel.cu:425-458, DGL's CustomCsrmm binary_reduce_sum.cu:338-360), so they cannot
 * hold a gespmm_plan. gespmm_set_auto_plan(k), k >= 1: gespmm_csr_spmm_f32, gespmm_csr_spmm_max_f32 and gespmm_dgl_csrmm_{sum,max}_f32
 * keep a small cache (8 entries, least recently used) keyed on (device, rowptr, colind, M, K, N, valued, variant, reducer); the k-th
 * call with one key builds a plan (synchronously), later calls run through it. Pointer identity is not pattern identity: every call
 * that uses a cached plan first fingerprints ALL of rowptr / colind / val on the device (one small kernel + a 32-byte read-back = one
 * stream synchronisation per call — the price of the switch: ~25 us on a com-Amazon-sized graph, where the plan saves ~60). A pattern
 * changed in place drops the plan, changed values are re-permuted. Where the plan's launch and the plain launch are one kernel each (no
 * long-row pass, no cache blocking) the calls after the plan's creation do not synchronise at all: the fingerprint is compared ON THE
 * DEVICE, the plan's kernel and the plain kernel are both enqueued behind the result (one of them runs, the other's workgroups leave at
 * once), and the host learns of a change from a record the check leaves in pinned memory, at the next call. Never on a capturing stream. Matrices whose analysis keeps the
 * storage order run the plain path without fingerprint from then on. Results: the plain call's bits.
 * k = 0 switches it off and frees the cached plans; gespmm_auto_plan_clear frees them and keeps the switch.
 */
typedef struct gespmm_auto_plan_stats {
    int64_t calls_planned;     /* products that ran through a cached plan */
    int64_t plans_created;
    int64_t invalidated;       /* plans dropped because the pattern's fingerprint changed */
    int64_t values_refreshed;  /* gespmm_plan_set_values calls after the values' fingerprint changed */
    int64_t fingerprints;      /* fingerprint passes (= stream synchronisations the switch added) */
    int64_t cached_plans;      /* plans alive now */
    int64_t calls_async;       /* of calls_planned: launched behind a device-side guard, without any synchronisation (see above) */
} gespmm_auto_plan_stats;
int gespmm_set_auto_plan(int32_t kth_call);
void gespmm_auto_plan_clear(void);
int gespmm_auto_plan_get_stats(gespmm_auto_plan_stats* out);

/*
 * The clustering by itself, HOST pointers (what gespmm_plan_create runs on its host copy): multi-level label
 * propagation on the bipartite row/column graph; perm_out[i] = row at position i. Deterministic, independent of
 * `threads`. levels_out / clusters_out[16] (row clusters after each level) may be NULL.
 */
int gespmm_cluster_rows(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int32_t threads,
                        int32_t* perm_out, int32_t* levels_out, int32_t* clusters_out);
/* Study hook (scripts/cluster_chain_study.py; HOST pointers): the same clustering with its depth / sweeps given, plus the coarsest
 * cluster of every row (top_label_out[M], may be NULL). Returns the number of levels built or a negative error. */
int gespmm_cluster_rows_study(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int32_t max_levels, int32_t sweeps,
                              int32_t* perm_out, int32_t* top_label_out, int32_t* clusters_out);
/*
 * Model of the per-XCD L2 used to judge an order (HOST pointers): rows processed in `perm` order (NULL = storage
 * order) in `slices` contiguous parts of equal non-zero count, each with an LRU of `window_rows` B rows; returns
 * the share of non-zeros whose B row is resident when gathered.
 */
double gespmm_simulate_l2_hits(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, const int32_t* perm,
                               int32_t slices, int64_t window_rows);

/*
 * The device analysis passes by themselves (what gespmm_plan_create runs; DEVICE rowptr / colind, HOST outputs):
 * gespmm_device_cluster_rows gives the SAME order as gespmm_cluster_rows (the rules are identical and every sum is an
 * integer sum); gespmm_device_l2_model estimates what gespmm_simulate_l2_hits simulates (exact LRU stack distances of
 * `samples_per_slice` stratified accesses per slice; max_entries_per_slice > 0 models only the head of every slice).
 */
int gespmm_device_cluster_rows(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                               int32_t* perm_out_host, int32_t* levels_out, int32_t* clusters_out, void* stream);
double gespmm_device_l2_model(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t nnz,
                              const int32_t* perm_host, int32_t slices, int64_t window_rows, int64_t max_entries_per_slice,
                              int32_t samples_per_slice, void* stream);
/* Test hook: task table of a clustered plan (which = 0 wavefront tasks, 1 lane-group tasks), int4 records, HOST memory. */
int gespmm_plan_debug_tasks(const gespmm_plan* plan, int32_t which, int32_t* out_host, int64_t capacity);

/*
 * The plan's POLICY by itself (host only, no device, no matrix): what gespmm_plan_create / gespmm_plan_spmm_f32 decide for a
 * matrix of this shape once the analysis has produced the numbers in the query (csrc/plan_policy.cpp holds the rules, each
 * with the log it was measured in). Lets callers and tests ask "what would a plan do" without building one.
 */
typedef struct gespmm_plan_policy_query {
    int64_t M, K, nnz, N;
    int64_t N_launch;        /* width of the launch asked about (0 = N) */
    int32_t variant;         /* GESPMM_VARIANT_* */
    int32_t max_degree;      /* longest row */
    int32_t reorder, kernel, analysis, flags, task_entries, row_floor; /* as in gespmm_plan_options */
    double hits_before, hits_after; /* modelled L2 hit rates in storage / clustered order */
    double staged_fraction;  /* share of the entries whose B row a staged-rows block holds in LDS */
    /* since 0.3 (read by gespmm_plan_policy_v2 when q_bytes covers them) */
    int32_t expected_launches; /* as in gespmm_plan_options (0 = 200) */
    int32_t cold_start;        /* since 0.3: 1 = the process has built no plan and gespmm_init has not run — the first analysis also loads
                                  its kernels and makes its arena (~29 ms): the cost rule is asked with that on top */
    double wedge_probe;      /* share of sampled (row r; c1, c2 in r) wedges with c2 in row c1 — the plan's cheap structure probe on square
                                matrices; negative = unknown (rectangular matrix, host analysis) */
    double record_slot_fill; /* padded-record kernel (round 6, read when q_bytes covers it): share of the batches' entry slots that carry an
                                entry once its tables are built; negative = not known yet (keep_records then repeats build_records) */
} gespmm_plan_policy_query;
typedef struct gespmm_plan_policy_answer {
    int32_t launch_flags;      /* user flags + GESPMM_FLAG_SPLIT_LONG_ROWS or _STRICT_ORDER */
    int32_t analyse;           /* clustering + L2 model run at all */
    int32_t dense_try;         /* dense graph: the clustered order needs >= 0.65 modelled hits */
    int32_t keep_clustered;    /* ... and is kept, given hits_before / hits_after */
    int32_t task_entries, group_task_entries, row_floor;
    int32_t build_staged, keep_staged; /* staged-rows tables are built / kept, given staged_fraction */
    int32_t shallow_unroll;
    int32_t segmented;         /* the streaming launch at N_launch takes the segmented-stream kernel */
    int32_t sddmm_route;       /* 0 CSR call, 1 COO on expanded row ids, 2 clustered edge order + scatter */
    int32_t narrow_vec4;       /* the launch at N_launch <= 64 takes 4 floats per lane (variant 3) instead of 1 */
    int64_t model_window, model_sample;
    /* since 0.3 (written by gespmm_plan_policy_v2 when a_bytes covers them) */
    int32_t cost_skipped;      /* reorder = AUTO would have analysed, but gain x launches < cost: storage order, no analysis */
    int32_t cluster_levels;    /* clustering depth of this plan (0 = the clustering's own default: six levels) */
    double est_gain_us;        /* estimated saving per launch if the clustered order is kept */
    double est_cost_us;        /* estimated time of the analysis */
    int32_t cluster_sweeps;    /* label-propagation sweeps per level (0 = the clustering's own default: five) */
    int32_t staged_rows;       /* rows per block of the staged-rows tables at this width (0: the width is not served) */
    /* round 6 (written when a_bytes covers them): the padded-record kernel (GESPMM_PLAN_KERNEL_RECORDS) */
    int32_t build_records, keep_records; /* its tables are built (clustered order kept, narrow width, short rows, no staged tables kept) /
                                            kept, given record_slot_fill */
    int32_t records_batches;   /* batches a wavefront task of those tables is cut at */
    int32_t slab_ranges;       /* column-slab tables (GESPMM_PLAN_KERNEL_STAGED_SLABS): ascending column ranges the plan cuts its clustered matrix
                                  into at this width — 0: no such tables (kept once >= 50 % of the entries find their B row staged) */
} gespmm_plan_policy_answer;
int gespmm_plan_policy(const gespmm_plan_policy_query* q, gespmm_plan_policy_answer* a);  /* the 0.2 layouts (up to staged_fraction / model_sample) */
int gespmm_plan_policy_v2(const gespmm_plan_policy_query* q, int64_t q_bytes, gespmm_plan_policy_answer* a, int64_t a_bytes);

/*
 * Comparison column, not a product path: the Gunrock app's edge map
 * (gunrock-test/app/spmm/spmm_enactor.cuh:92-105) — for every edge (src -> dest) of the CSR pattern
 * and every feature j, atomicAdd(out + dest*N + j, in[src*N + j]): out[K x N] = A^T * in with A == 1,
 * `out` zeroed by the call. `in` is M x N. The order of the additions is not fixed (atomics), so
 * results are tolerance-checked. spmm_test --atomic-baseline times it next to the row-product kernels.
 */
int gespmm_baseline_atomic_scatter_f32(const int32_t* rowptr, const int32_t* colind, const float* in, float* out,
                                       int64_t M, int64_t K, int64_t N, int64_t nnz, void* stream);

/*
 * Yardstick, not a product path: dst[i] = src[i] for n floats as one streaming kernel (dwordx4 per lane where both
 * pointers are 16-byte aligned). bench.py times it in the same process as the product, so that `roofline.ceiling_frac`
 * is priced with the read + write rate of the box the product ran on (no reference counterpart: the reference reports
 * GFLOP/s only, spmm_test.cu:728-738).
 */
int gespmm_baseline_copy_f32(const float* src, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ host side */

/*
 * MatrixMarket coordinate loader with the reference's readMtx semantics
 * (1-based -> 0-based, `symmetric` expanded with self-loops and duplicates
 * dropped, result sorted by (row, col); real -> value, integer -> value,
 * pattern -> 1.0). Output arrays are HOST memory allocated by the library;
 * release them with gespmm_mtx_free(). Unlike the reference this never exits:
 * a missing file is GESPMM_EIO, a bad banner/size line GESPMM_EFORMAT.
 */
typedef struct gespmm_coo {
    int32_t  nrows;
    int32_t  ncols;
    int64_t  nnz;
    int32_t* row;   /* [nnz] */
    int32_t* col;   /* [nnz] */
    float*   val;   /* [nnz] */
} gespmm_coo;

int  gespmm_mtx_read(const char* path, gespmm_coo* out);
void gespmm_mtx_free(gespmm_coo* coo);

/*
 * Same result through a binary cache: the parsed, expanded and sorted COO of `path`
 * is stored as `<cache_dir>/<basename>.<bytes>.<mtime>.gespmm-coo` on the first read
 * and read back (one fread per array) afterwards; a changed file gets a new cache
 * name. cache_dir == NULL behaves like gespmm_mtx_read. (The reference re-parses the
 * text with fscanf on every run, util.hpp:104-216 — seconds for 10^7-entry files.)
 */
int  gespmm_mtx_read_cached(const char* path, const char* cache_dir, gespmm_coo* out);

/*
 * COO (any order) -> CSR by counting sort on the row, keeping the input order
 * inside each row. val_in == NULL writes 1.0f for every entry, which is what
 * spmm_test.cu:574 does with the file's values. All HOST pointers. An index
 * outside [0,nrows) x [0,ncols) is GESPMM_EINVAL (the reference only prints
 * "out of bound row/column", spmm_test.cu:563,571, and corrupts memory).
 */
int gespmm_coo_to_csr(int32_t nrows, int32_t ncols, int64_t nnz,
                      const int32_t* row, const int32_t* col, const float* val_in,
                      int32_t* rowptr, int32_t* colind, float* val_out);

/*
 * 1-D row partition of a CSR matrix into `parts` contiguous row ranges with
 * (nearly) equal non-zero counts: cut[p] = first row of part p, cut[parts] = M.
 * rowptr is HOST memory (M+1 entries); cut has parts+1 entries.
 */
int gespmm_row_partition(const int32_t* rowptr, int64_t M, int32_t parts, int64_t* cut);

#ifdef __cplusplus
}
#endif
#endif /* GESPMM_H_ */
