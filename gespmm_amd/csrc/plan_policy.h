// plan_policy.h — every DECISION the plan object takes, as pure functions of numbers (no HIP calls, no device state).
//
// plan.cpp is the mechanism (device passes, tables, launches); this file is the policy. Each rule cites the log it was
// measured in; tests/test_plan_policy.py pins the answers for the BASELINE shapes and the hold-out graphs
// (profiles/r04/holdout_audit.log) through the C entry point gespmm_plan_policy().
#pragma once
#include <stdint.h>

#include "spmm_kernels.h"

namespace gespmm {

// What is known before / after the analysis passes.
struct PlanFacts {
    int64_t M = 0, K = 0, nnz = 0, N = 0;  // N: the width the plan is made for
    int variant = -1;                      // the caller's GESPMM_VARIANT_*
    int sel_variant = 0;                   // what a plain call resolves to (select.cpp)
    bool slab_blocked = false;             // ... and whether it takes the cache-blocked path (dense graphs)
    int64_t tile_cols = 0;                 // columns one workgroup tile covers (group x vec x strips)
    int32_t max_degree = 0;                // longest row (device_validate_csr)
    int reorder_mode = 0;                  // GESPMM_PLAN_REORDER_*
    int kernel_choice = 0;                 // GESPMM_PLAN_KERNEL_*
    bool host_analysis = false;            // GESPMM_PLAN_ANALYSIS_HOST
    int user_flags = 0;                    // gespmm_plan_options.flags
    int opt_task_entries = 0, opt_row_floor = 0;
    int expected_launches = 0;             // gespmm_plan_options.expected_launches (0 = kDefaultExpectedLaunches)
    bool cold_start = false;               // the process has built no plan and gespmm_init has not run: the analysis also loads its kernels
    double wedge_probe = -1.0;             // share of sampled wedges (c1, c2 in one row) that close (c2 in row c1); < 0: unknown

    int64_t mean_ceil() const { return M > 0 ? (nnz + M - 1) / M : 0; }
    int64_t mean_floor() const { return M > 0 ? nnz / M : 0; }
    int64_t b_bytes() const { return K * 4 * (N < tile_cols ? N : tile_cols); }  // B as one column tile sees it
};

// user flags + the long-row decision from the longest row (SPLIT_LONG_ROWS or STRICT_ORDER is always set on return)
int long_row_flags(int64_t M, int64_t nnz, int32_t max_degree, int user_flags);
// rounded-up mean degree from which AUTO considers the staged-rows kernel at width N (round 5: 4 at 128 columns, 5 elsewhere — with the
// record-stream walk the kernel is ahead on short rows at 128 columns too: com-Amazon-shaped communities 92.7 vs 107 us, planted
// communities of mean degree 4 / 5 / 6 / 8 / 12: x1.05 / x1.06 / x1.09 / x1.14 / x1.19, profiles/r05/staged_degree_sweep{,_retuned}.log;
// until round 4 short rows were level at best and 12 was asked)
inline int staged_min_mean_degree(int64_t N) { return staged_tile_class(N) <= 128 ? 4 : 5; }  // (rounded-up mean degree; staged_degree_sweep_retuned.log)

// ---- before the analysis: launch flags, whether to cluster at all, how to model the L2s
constexpr int kDefaultExpectedLaunches = 200;  // the reference's protocols: ITER = 200 (spmm_test.cu:714), 200 epochs (gcn_custom.py:134)

// What clustering is expected to buy per launch and what the analysis is expected to cost (both in us): the numbers decide_analysis
// weighs under reorder = AUTO. Pure functions of the facts (shape, width, probe).
struct CostEstimate {
    double hits_gain = 0.0;  // expected gain in modelled L2 hit rate
    double gain_us = 0.0;    // per launch
    double cost_us = 0.0;    // once
};
CostEstimate estimate_analysis_cost(const PlanFacts& f);

struct AnalysisDecision {
    int launch_flags = 0;      // user flags + the long-row decision (exact: the plan has seen the longest row)
    bool cost_skipped = false; // AUTO would analyse, but the expected launches do not amortise the analysis
    CostEstimate cost;
    bool analyse = false;      // run clustering + L2 model
    bool dense_try = false;    // a dense graph: the clustered order is kept only on strong community structure
    int64_t model_window = 0;  // B rows one XCD's L2 is modelled to hold
    int64_t model_sample = 0;  // entries per slice the model looks at (0 = all)
};
AnalysisDecision decide_analysis(const PlanFacts& f);
// Could an analysis of a matrix of this shape pay for itself inside `expected_launches` (0 = 200) once the library is warm — with the most
// structure the probe could report? If not, there is nothing to warm up for: gespmm_init (~60 ms) would be pure cost (pubmed-sized graphs:
// the reference's GCN run would pay 0.3-0.6 ms per epoch for it).
bool analysis_could_pay(int64_t M, int64_t K, int64_t nnz, int64_t N, int expected_launches);

// Clustering depth (0 = the library's default of 6 levels). Levels 4-6 merge little and cost ~2 ms of launch latency on a
// com-Amazon-sized graph; they are worth 4.5 % per launch on the structureless stand-in and nothing on graphs with communities
// (profiles/r04/like_regression.log, cluster_levels.log) — a plan that expects fewer than 2000 launches stops at three.
int cluster_levels_for(const PlanFacts& f);
int cluster_sweeps_for(const PlanFacts& f);  // label-propagation sweeps per level (0 = the clustering's own default)
int model_points_for(const PlanFacts& f);    // sampled accesses per slice in the L2 model
int staged_rows_for(const PlanFacts& f, int shape_rows, int shape_waves);  // rows per block of the staged-rows kernel

// ---- after the model: is the clustered order worth its per-launch indirection?
bool keep_clustered_order(const PlanFacts& f, const AnalysisDecision& a, double hits_before, double hits_after);

// ---- a clustered plan: task sizes, unroll depth, which tables to build
struct PlanKernelDecision {
    int task_entries = 0;        // non-zeros per wavefront task (batch-stream kernel)
    int group_task_entries = 0;  // non-zeros per lane-group task (segmented-stream kernel)
    int64_t row_floor = 0;       // a row counts as at least this many entries when tasks are cut
    bool build_staged = false;   // build the staged-rows tables for width N
    bool shallow_unroll = false; // 4 instead of 8 B rows in flight per lane group
};
PlanKernelDecision choose_plan_kernel(const PlanFacts& f, double hits_after);

// ---- the staged tables exist: does enough of the matrix find its B row staged?
// Column-slab tables (round 6): how many ascending column ranges the clustered matrix is cut into for the staged-rows kernel (0: no such
// tables), and whether they stay once the share of entries that find their B row staged is known.
int slab_count_for(const PlanFacts& f);
bool keep_slab_tables(const PlanFacts& f, double staged_fraction);
bool keep_staged_tables(const PlanFacts& f, double staged_fraction);
// The clustering was judged no better than the storage order (keep_clustered_order == false): does the plan still make its own copy
// of the matrix, in the storage order, because the staged-rows kernel — which walks the plan's tables — would be built for it?
bool storage_order_wants_plan_copy(const PlanFacts& f, double hits_before);
// Padded-record kernel (spmm_records.hip): are its tables built for this plan (order_hits: modelled L2 hits of the order the plan
// processes the rows in, < 0 unknown), and the batches a wavefront task is cut at.
bool want_record_tables(const PlanFacts& f, double order_hits);
int records_batches_per_task(const PlanFacts& f);
bool keep_record_tables(const PlanFacts& f, double slot_fill);  // slot_fill: share of the batches' entry slots that carry an entry

// ---- per launch (width N_launch may differ from the plan's): segmented-stream instead of batch-stream?
bool prefer_segmented(const PlanFacts& f, double hits_after, int64_t N_launch);

// ---- per launch at N_launch <= 64: four floats per lane (8 lanes per 32 columns = 8 rows per gather instruction) instead of the
//      one-lane-per-column geometry AUTO takes at narrow widths
bool narrow_vec4(const PlanFacts& f, double hits_after, int64_t N_launch);

// ---- SDDMM through the plan: 0 = CSR form on the caller's arrays, 1 = COO form on expanded row ids (storage order),
//      2 = the plan's clustered edge order + scatter
int sddmm_route(const PlanFacts& f, bool reordered, double hits_after, int64_t N_launch);

}  // namespace gespmm
