// plan_policy.cpp — the rules behind gespmm_plan_create / gespmm_plan_spmm_f32 (see plan_policy.h).
//
// Every threshold below was measured on this chip; the log is named next to it. The hold-out audit (graphs from generators
// this repository did not write: profiles/r04/holdout_audit.log, scripts/holdout_audit.py) is the check that they are not
// fitted to the stand-ins of gespmm_amd/graphs.py.

#include <cstdlib>
#include "plan_policy.h"

#include <cstddef>
#include <cstring>

#include "../../include/gespmm.h"
#include "select.h"
#include "spmm_kernels.h"

namespace gespmm {

// Long-row pass: the plain entry points run it on matrices of >= 2^23 non-zeros, or >= 2^20 with mean degree >= 8
// (select.cpp: they cannot see the degrees, and the pass costs ~6.5 us when it finds nothing). A plan runs it on the SAME
// matrices — so its result has the plain call's bits — but only when the longest row it has SEEN is beyond the threshold.
int long_row_flags(int64_t M, int64_t nnz, int32_t max_degree, int user_flags) {
    const int64_t mean = M > 0 ? (nnz + M - 1) / M : 0;
    int64_t threshold = 32 * mean;
    if (threshold < kLongRowThreshold) threshold = kLongRowThreshold;
    const bool plain_would = nnz >= kLongRowMinNnz || (nnz >= (1 << 20) && M > 0 && nnz / M >= 8);
    if (!(user_flags & (GESPMM_FLAG_STRICT_ORDER | GESPMM_FLAG_SPLIT_LONG_ROWS)))
        user_flags |= (plain_would && max_degree > threshold) ? GESPMM_FLAG_SPLIT_LONG_ROWS : GESPMM_FLAG_STRICT_ORDER;
    return user_flags;
}

static bool crc_family(int v) { return v >= GESPMM_VARIANT_CRC && v <= GESPMM_VARIANT_CRC_CWM8; }
static bool deep_analysis(const PlanFacts& f);  // (below: how much analysis a plan's expected life buys)

// Cost awareness (round 5). Until round 4 AUTO clustered every matrix with >= 16 384 rows and a B beyond 8 MB, whatever the analysis
// cost and however few launches would use it: the reference's own GCN configuration (pubmed, hidden 128, cached=True) got 52 % SLOWER
// per epoch, the structureless com-Amazon stand-in lost 16 % inside the reference's 200-launch protocol (profiles/r04/gcn_epochs.log,
// bench_round4.log). Two estimates, both from numbers known BEFORE the clustering:
//   gain per launch = gathered bytes (4 nnz N) x expected gain in L2 hit rate x 0.085 us per MB — measured savings per MB of gathers that
//     turn from misses into hits: 0.071 (com-Amazon-shaped communities: 150 -> 107 us), 0.097 (structureless), 0.10 (LFR, geometric),
//     0.12 (products-shaped): profiles/r05/probe_calibration.log;
//   expected gain in hit rate from the WEDGE PROBE (plan_device.hip: k_wedge_probe, one small kernel riding on the validation pass):
//     the share of sampled (row r; two of its columns c1, c2) wedges with c2 in row c1. Structureless graphs close none (1e-4: gains
//     0.02-0.15), every graph with communities, triangles or geometry closes 6-58 % (gains 0.29-0.90): 0.1 + 0.9 sqrt(probe), capped
//     at 0.85; unknown (rectangular matrix, host analysis): 0.6 — the benefit of the doubt;
//   cost of the analysis = a fixed part (launch and synchronisation latency at any size: 3.3 / 5 ms by the plan's expected life, below)
//     + 0.55-0.6 ns per entry — device analysis; the host form is ~30x that.
CostEstimate estimate_analysis_cost(const PlanFacts& f) {
    CostEstimate c;
    if (f.wedge_probe < 0.0) c.hits_gain = 0.60;  // unknown: the benefit of the doubt — only matrices too small to ever pay are skipped
    else {
        double r = 0.0, x = f.wedge_probe;  // sqrt by Newton (no <cmath> dependency on the device toolchain's host pass)
        if (x > 0.0) {
            r = x < 1.0 ? 1.0 : x;
            for (int i = 0; i < 40; ++i) r = 0.5 * (r + x / r);
        }
        c.hits_gain = 0.10 + 0.90 * r;
        if (c.hits_gain > 0.85) c.hits_gain = 0.85;
    }
    const double gathered_mb = 4.0 * (double)f.nnz * (double)f.N / 1e6;
    c.gain_us = gathered_mb * c.hits_gain * 0.085;
    const double e = (double)f.nnz;
    // (round 5, after the analysis kernels were reworked — profiles/r05/plan_ms_after_kernel_work.log: 3.4 / 5.0 / 5.4 / 7.9 / 73.6 ms at
    //  1.85 / 4.7 / 7.2 / 11.0 / 124 M entries for plans with a short life (three levels, 1 024 model samples per slice; +0.5 ms with five sweeps),
    //  6.4 / 7.9 / 78.9 ms at 1.85 / 7.2 / 124 M for the others; the fixed parts carry ~0.6 ms of margin: the gain side of the rule is an
    //  estimate from a probe — Barabasi-Albert: 30 us estimated, 16 measured)
    c.cost_us = !deep_analysis(f) ? 3300.0 + 0.55e-3 * e : 5000.0 + 0.6e-3 * e;
    if (f.host_analysis) c.cost_us *= 30.0;
    // The first analysis of a process also loads the analysis kernels and makes the arena: 33.0 ms against 3.7 on the headline graph
    // (BENCH_r05 plan_ms_first_creation; profiles/r06/plan_cold.log). Under the reference's protocol — one process per matrix, 200
    // launches — that is the cost that counts: 200 x 89 us + 33 ms is slower than 200 plain launches of 150 us. gespmm_init removes it.
    if (f.cold_start && !f.host_analysis) c.cost_us += 29000.0;
    return c;
}

bool analysis_could_pay(int64_t M, int64_t K, int64_t nnz, int64_t N, int expected_launches) {
    if (M <= 0 || nnz <= 0 || N <= 0) return false;
    PlanFacts f;
    f.M = M;
    f.K = K;
    f.nnz = nnz;
    f.N = N;
    f.expected_launches = expected_launches;
    f.wedge_probe = 1.0;  // the most structure a probe can report (hits gain capped at 0.85)
    f.cold_start = false;
    const CostEstimate c = estimate_analysis_cost(f);
    const int launches = expected_launches > 0 ? expected_launches : kDefaultExpectedLaunches;
    return c.gain_us * (double)launches >= c.cost_us;
}

AnalysisDecision decide_analysis(const PlanFacts& f) {
    AnalysisDecision a;
    a.launch_flags = long_row_flags(f.M, f.nnz, f.max_degree, f.user_flags);
    const int64_t mean = f.mean_ceil();
    const bool stream_family = crc_family(f.sel_variant) && !f.slab_blocked;
    const bool big_enough = f.M >= (1 << 14) && f.nnz >= f.M && f.nnz <= (1ll << 28) && f.b_bytes() > (8ll << 20);
    if (f.reorder_mode == GESPMM_PLAN_REORDER) {
        a.analyse = stream_family && f.M > 1 && f.nnz > 0;
        // (asked for by name, the column-slab tables are made for dense graphs — the plain call's cache-blocked family — too)
        if (!a.analyse && f.kernel_choice == GESPMM_PLAN_KERNEL_STAGED_SLABS && crc_family(f.sel_variant) && f.M > 1 && f.nnz > 0 && !f.host_analysis) {
            a.analyse = true;
            a.dense_try = f.slab_blocked;
        }
    } else if (f.reorder_mode == GESPMM_PLAN_REORDER_AUTO) {
        // B beyond the L2s (below that every order hits), enough rows to cluster
        a.analyse = stream_family && big_enough && mean <= 96;
        // Dense graphs (the plain call's cache-blocked path): worth clustering only when they have STRONG community
        // structure — a reddit-sized graph with planted communities modelled at 0.71-0.77 hits runs 3.0 vs 4.0 ms at N = 128
        // through a clustered plan; modelled at 0.37-0.50 the cache-blocked path wins (4.1 vs 4.9 ms), and on the
        // structureless stand-in by 2x (profiles/r03/dense_community_audit.log). The analysis runs and keep_clustered_order()
        // asks for 0.65.
        // Hold-out check (networkx LFR, mean degree 324-330, profiles/r04/holdout_audit_dense.log): modelled 0.75 / 0.68 -> clustered wins
        // (N = 32 / 64: 414 vs 464, 531 vs 646 us), 0.62 -> 4 % either way, 0.23-0.45 against the cache-blocked path -> clustered loses
        // 1.3-1.9x: the bar holds. It is a bar against the CACHE-BLOCKED path only: at widths that path does not serve (N = 32) a dense
        // graph is judged like any other (mu = 0.5 at N = 32, 0.31 -> 0.45: 436 vs 469 us).
        if (!a.analyse && !f.host_analysis && (f.slab_blocked || mean > 96) && crc_family(f.sel_variant) && big_enough) {
            a.analyse = true;
            a.dense_try = f.slab_blocked;
        }
    }
    a.cost = estimate_analysis_cost(f);
    if (a.analyse && f.reorder_mode == GESPMM_PLAN_REORDER_AUTO) {
        const int launches = f.expected_launches > 0 ? f.expected_launches : kDefaultExpectedLaunches;
        if (a.cost.gain_us * (double)launches < a.cost.cost_us) {
            a.analyse = false;  // the storage order serves these launches faster than an analysis they would have to pay for
            a.dense_try = false;
            a.cost_skipped = true;
        }
    }
    // the model of the XCD L2s: window = B rows that 3 MiB hold; matrices beyond 2^25 non-zeros: the first 2^22
    // non-zeros of each of the 8 slices are the sample
    a.model_sample = f.nnz <= (1ll << 25) ? 0 : (1ll << 22);
    const int64_t rowb = 4 * (f.N < f.tile_cols ? f.N : f.tile_cols);
    a.model_window = (3ll << 20) / (rowb > 0 ? rowb : 4);
    return a;
}

// How much analysis a plan's expected life buys. Deep = the clustering's defaults (six levels, five sweeps per level), shallow = three
// of each: 3.4 ms less on a com-Amazon-sized graph, and a launch that is 2-4 % slower (profiles/r05/plan_life_compare.log: com-Amazon-
// shaped 95.3 / 179 / 381 us against 93.2 / 174 / 366 at N = 128 / 256 / 512, geometric 205 / 352 / 692 against 197 / 341 / 661) — a
// share of a time that grows with N, so the launches it takes to pay for the depth shrink with N: ~1 600 / 220 at N = 128 / 512 on the
// first graph, ~340 / 90 on the second. launches x N >= 100 000 (780 launches at N = 128, 200 at N = 512) sits between them.
static bool deep_analysis(const PlanFacts& f) {
    const long long launches = f.expected_launches > 0 ? f.expected_launches : kDefaultExpectedLaunches;
    return launches * (long long)(f.N > 0 ? f.N : 1) >= 100000ll;
}

int cluster_levels_for(const PlanFacts& f) { return deep_analysis(f) ? 0 : 3; }

int cluster_sweeps_for(const PlanFacts& f) {
    // label-propagation sweeps per level: five (the default of cluster_rows) or three. With one lane group per degree class the two extra
    // sweeps cost 0.5-1.5 ms (profiles/r05/levels_vs_sweeps.log, N = 128: com-Amazon-shaped 3.85 -> 4.32 ms, geometric 6.08 -> 6.63,
    // Holme-Kim 7.13 -> 8.63) and buy 2-5 % of the launch (93.9 -> 89.4 us, 200.9 -> 196.1, 409 -> 391): paid back within ~100 launches
    // at N = 128 — more than the three extra LEVELS buy on the same graphs. Three sweeps only for plans that will not live that long.
    const long long launches = f.expected_launches > 0 ? f.expected_launches : kDefaultExpectedLaunches;
    return launches * (long long)(f.N > 0 ? f.N : 1) >= 12800ll ? 0 : 3;
}

int model_points_for(const PlanFacts& f) {
    // sampled accesses per XCD slice in the L2 model. A sample walks back through its slice until it has seen a window of distinct
    // columns: 32 768 samples are ~0.3 GB of reads per model, 0.39 ms on the headline graph, twice per plan. A quarter of them puts the
    // standard error of a modelled hit rate at 0.55 % (0.28 % before) — the rules that read it have margins of 3 % and more.
    return deep_analysis(f) ? 4096 : 1024;
}

bool keep_clustered_order(const PlanFacts& f, const AnalysisDecision& a, double hits_before, double hits_after) {
    if (f.reorder_mode != GESPMM_PLAN_REORDER_AUTO) return true;
    // the storage order is as good (already local, or nothing to find): keep it and pay nothing per launch. (0.05 until round 4: a
    // Barabasi-Albert graph at N = 32 modelled 0.112 -> 0.159 kept its storage order and was 5 % behind its clustered plan, the same graph
    // at N = 128, 0.038 -> 0.099, took it and gained 5 %: profiles/r04/holdout_audit.log)
    if (hits_after < hits_before + 0.03) return false;
    if (a.dense_try && hits_after < 0.65) return false;
    return true;
}

// Non-zeros per wavefront task. Storage-order launches take ~12 KB of gathered B per task (select.cpp);
// clustered plans take ~20 KB (40 entries at N = 128, 32 at N >= 256, 80 at N = 64: profiles/r02/plan_task_size_final.log — at
// N = 128 anything from 40 to 80 entries runs within 1 %, and the smaller task keeps fewer rows in flight per XCD: fabric bytes
// 1.48x algorithmic at 40 entries, 1.54x at 48, 1.62x at 56, plan_task_size_traffic.log). Round 4 re-swept the sizes on the stand-ins
// and the hold-out graphs (profiles/r04/constants_resweep.log): N = 128 / 512 confirmed; at N = 32 the cap of 96 entries cost 3-11 %
// on EVERY graph (128-192 entries: com-Amazon-shaped 34.9 vs 36.7 us, structureless 45.5 vs 49.2, LFR 70-71 vs 78-80, geometric 100 vs
// 106) — the cap is 192 now (N = 32: 160 entries).
static int default_task_entries(int64_t N) {
    const int64_t row_bytes = 4 * (N < 256 ? N : 256);
    int64_t t = (20 << 10) / (row_bytes > 0 ? row_bytes : 4);
    if (t < 32) t = 32;
    if (t > 192) t = 192;
    return (int)t;
}

PlanKernelDecision choose_plan_kernel(const PlanFacts& f, double hits_after) {
    PlanKernelDecision d;
    const int64_t mean = f.mean_ceil();
    const bool te_given = f.opt_task_entries > 0;
    int budget = te_given ? f.opt_task_entries : default_task_entries(f.N);
    // ... never fewer than ~5 rows of mean length per task (products-shaped graphs, degree 50: 256-entry tasks at N = 32 run
    // 1.48 ms, 96-entry tasks 2.33 ms) — up to 256 columns; beyond (two or more column tiles per row) 2 rows: 5 cost 4 % at N = 512
    // on the LFR and products-shaped graphs (constants_resweep.log: 832 vs 869 us, 3743 vs 3916)
    const int64_t min_rows = f.N > 256 ? 2 : 5;
    if (!te_given && budget < min_rows * mean) budget = (int)(min_rows * mean < 512 ? min_rows * mean : 512);
    d.task_entries = budget;
    d.row_floor = f.opt_row_floor < 0 ? 0 : (f.opt_row_floor > 0 ? f.opt_row_floor : 8);
    // segmented-stream kernel: a task per lane GROUP, cut by non-zeros alone, half the budget (short rows: 16 entries per lane
    // group — profiles/r02/plan_seg_task_size.log: 139 us at 16, 146 at 24, 150 at 32 on the com-Amazon stand-in)
    int gb = budget / 2 > 16 ? budget / 2 : 16;
    if (te_given) gb = f.opt_task_entries / 2 > 4 ? f.opt_task_entries / 2 : 4;
    else if (mean < 16) gb = 16;
    d.group_task_entries = gb;
    // Staged-rows kernel: worth its tables where a block of clustered rows uses the same B rows again and again
    // (profiles/r03/staged_rows.log, staged_degree_sweep.log; products-shaped communities: 3.0 vs 3.9 ms at N = 128, 5.8 vs
    // 7.8 ms at N = 256). Until round 4 short rows were level at best at N = 128 (com-Amazon-shaped communities: 108 vs 106 us; mean
    // degree 8: 239 vs 236 us) and a mean degree of 12 was asked there; with round 5's walk (row ends in the stream, packed
    // multiply-adds) they win at both widths: com-Amazon-shaped 92.7 vs 107 us (N = 128), 175 vs 216 us (N = 256); mean degree 6 / 8:
    // x1.06 / x1.14 at N = 128 (profiles/r05/staged_degree_sweep.log, kernel_ab_record_stream.log). Device analysis only.
    const bool v4 = f.variant == GESPMM_VARIANT_AUTO || f.variant == GESPMM_VARIANT_CRC_CWM4 || f.variant == GESPMM_VARIANT_CRC_CWM8;
    const int sclass = staged_kernel_class(f.M, f.K, f.N);
    const bool fits = f.nnz > 0 && sclass != kStagedNone && staged_stream_fits(f.M, f.nnz);
    const bool narrow = sclass == kStagedNarrow;  // N = 16 / 32 / 64: the lane-group form of the kernel (spmm_staged_narrow.hip)
    // the wide kernels AUTO considers: the tuned widths, and (round 6, spmm_staged_gen.hip) every even width beyond 64 columns — two or four
    // floats per lane, i.e. 128- or 256-column tiles, judged by the thresholds of that tile class; odd widths (one float per lane, two
    // gather instructions per 128 columns) and widths up to 64 other than 32 / 64 are served on request only
    const bool wide_auto = sclass == kStagedTuned || (sclass == kStagedGeneral && f.N > 64 && f.N % 2 == 0);
    const bool want = f.kernel_choice == GESPMM_PLAN_KERNEL_STAGED ||
                      (f.kernel_choice == GESPMM_PLAN_KERNEL_AUTO && wide_auto && mean >= staged_min_mean_degree(f.N) && hits_after >= 0.40 &&
                       f.nnz >= (1 << 20) && v4) ||
                      // the lane-group form at N = 32 / 64 (spmm_staged_narrow.hip): worth its tables where most of the entries will be
                      // staged — rows of 10+ entries in an order modelled at >= 0.65 hits; keep_staged_tables decides on the share, which is
                      // what the kernel's time follows. (0.75 until the plans with a short life clustered less: the small-world graph
                      // modelled just below it through such a plan, kept the streaming kernel at N = 32 and ran 150 instead of 124 us —
                      // profiles/r05/plan_life_compare.log. Tables that are then not kept cost 0.3-0.7 ms.)
                      (f.kernel_choice == GESPMM_PLAN_KERNEL_AUTO && narrow && (f.N == 32 || f.N == 64) && mean >= 10 && hits_after >= 0.65 &&
                       f.nnz >= (1 << 20) && f.variant == GESPMM_VARIANT_AUTO);
    d.build_staged = fits && want && !f.host_analysis;
    // Where the clustered order is modelled to hit L2 (>= 40 % of the gathers) four B rows in flight per lane group beat eight
    // at up to 128 columns (com-Amazon-shaped communities, N = 128: 105 vs 114 us, N = 64: 48 vs 60 us; at 256+ columns and on
    // the structureless graph eight stay ahead) — profiles/r02/plan_unroll_geometry.log (short rows only: degree-50 rows want
    // the depth — products-shaped communities, N = 32: 525 vs 365 us)
    d.shallow_unroll = hits_after >= 0.40 && f.N <= 128 && mean <= 8 && f.nnz >= (1 << 20) && !(f.user_flags & 0x20000);
    return d;
}

// A matrix that arrives in an order as local as the clustering's (the headline graph relabelled in its planted order models 0.70 in storage
// order, 0.66 clustered): until round 6 it kept its storage order AND the streaming kernels — N = 128 122 us where the same graph shuffled
// runs 82 through the staged-rows kernel (profiles/r06/records_preordered.log). Same rule as for a clustered order: choose_plan_kernel on the
// storage order's modelled hits.
bool storage_order_wants_plan_copy(const PlanFacts& f, double hits_before) {
    if (f.reorder_mode != GESPMM_PLAN_REORDER_AUTO || f.kernel_choice != GESPMM_PLAN_KERNEL_AUTO || f.host_analysis || hits_before < 0.0) return false;
    return choose_plan_kernel(f, hits_before).build_staged;
}

// Column-slab tables (plan.cpp: build_slab_tables). A block of the staged-rows kernel stages 160 B rows; on a dense clustered matrix a
// block of 96 rows refers to thousands of distinct columns and a seventh of its entries find their row in LDS (reddit-shaped communities,
// mean degree 492: 0.15 staged, 3.98 ms — behind the segmented-stream kernel's 2.97). Cut into P ascending column ranges with one
// staging list per (block, range), the same kernel stages 0.29 / 0.42 / 0.58 / 0.66 / 0.71 of the entries at P = 2 / 3 / 5 / 8 / 12
// (profiles/r06/reddit_slab_staged.log: products of the P ranges 3.38 / 3.00 / 2.40 / 2.23 / 2.25 ms) at the price of one pass over C
// per extra range. P ~ mean degree / 64. Explicit choice: always (GESPMM_SLABS overrides the count); AUTO: mean degree >= 96 at N = 128 —
// planted-community graphs of the reddit stand-in's size at mean degree 64 / 96 / 128 / 160 / 192 / 256 / 350 / 492, AUTO without slab
// tables against the tables at round(mean / 64) ranges (profiles/r06/slabs/slab_density.log): 315 vs 337 us (the one-launch staged kernel
// keeps mean 64, as it keeps the products-shaped graph) · 590 vs 464 · 799 vs 613 · 977 vs 787 · 1258 vs 946 · 1604 vs 1249 · 2179 vs 1631 ·
// 3024 vs 2212: x0.73-0.81 from 96 on, where the plan otherwise falls to the segmented-stream kernel.
int slab_count_for(const PlanFacts& f) {
    static const int env = getenv("GESPMM_SLABS") ? atoi(getenv("GESPMM_SLABS")) : 0;
    if (f.N != 128 || f.host_analysis || env < 0) return 0;  // (GESPMM_SLABS=-1: never — the A/B of profiles/r06/slabs/slab_density.log)
    const bool asked = f.kernel_choice == GESPMM_PLAN_KERNEL_STAGED_SLABS;
    const bool auto_ok = f.kernel_choice == GESPMM_PLAN_KERNEL_AUTO && f.variant == GESPMM_VARIANT_AUTO && f.mean_floor() >= 96;
    if (!asked && !auto_ok) return 0;
    int P = env > 0 ? env : (int)((f.mean_floor() + 32) / 64);
    if (P < 2) P = asked ? 2 : 0;
    const int cap = env > 0 ? 64 : 16;  // (the view takes up to 64 ranges; the rule stops at 16)
    return P > cap ? cap : P;
}

bool keep_slab_tables(const PlanFacts& f, double staged_fraction) {
    // (structureless dense graphs stage little whatever the cut: the streaming kernels / the cache-blocked path stay)
    return f.kernel_choice == GESPMM_PLAN_KERNEL_STAGED_SLABS || staged_fraction >= 0.50;
}

bool keep_staged_tables(const PlanFacts& f, double staged_fraction) {
    // Not enough reuse inside the blocks: the streaming kernels stay. The share of entries that find their B row staged is what
    // separates the graphs where the kernel wins from those where it loses — on the repository's stand-ins AND on the hold-out
    // graphs (profiles/r04/holdout_audit.log, time staged / best streaming kernel of the same plan):
    //   128-column tiles   share 0.94 geometric x0.90 · 0.89 small-world x0.89 · 0.63-0.67 planted communities x0.77-0.92 ·
    //                      0.57 LFR mu=0.1 x1.21 · 0.42 LFR mu=0.3 x1.41 · 0.34 Holme-Kim x1.7
    //   256-column tiles   0.80 geometric x0.69 · 0.77 small-world x0.73 · 0.72 com-Amazon-shaped x0.84 · 0.50-0.55 planted
    //                      communities x0.70-0.84 · 0.44 LFR mu=0.1 x0.99 (N = 256) / x0.93 (512) · 0.33 LFR mu=0.3 x1.13 / x1.03
    //                      (without the `nt` marks of round 3, which cost this tile width 4-9 %: holdout_audit.log after far_marks_by_graph.log)
    // (round 3 asked for 0.40 at both widths: fitted on the planted-community generator alone, 20-41 % behind on the LFR graphs)
    if (f.kernel_choice != GESPMM_PLAN_KERNEL_AUTO) return true;
    const bool tile128 = staged_tile_class(f.N) <= 128;  // (two floats per lane: N = 128 and the even widths the general kernel walks that way)
    if (f.N <= 64) {
        // Narrow widths (round 5, profiles/r05/kernel_ab_narrow.log; time of the best streaming kernel / lane-group staged kernel):
        //   share 0.93-0.98 geometric x1.35 (N = 32) / x1.35 (64) · 0.85-0.90 small-world x1.23 / x1.45 · 0.65-0.75 products-shaped
        //   communities (mean degree 50) x1.10 / x1.12 · 0.75-0.80 com-Amazon-shaped (mean degree 5.5) x0.91 / x0.84 ·
        //   0.63-0.75 LFR mu = 0.1 (mean degree 16) x0.69 / x0.74 · 0.22 structureless x0.57
        // — most entries staged, or long rows with two thirds staged. (0.85 until the plans with a short life clustered less: the
        //   small-world graph through such a plan stages 0.844 / 0.831 of its entries at N = 32 / 64 instead of 0.876 / 0.856, lost its
        //   tables and ran 153 / 285 us instead of ~125 / ~200 — profiles/r05/plan_life_compare.log. Between LFR's x0.69 at 0.71 and the
        //   small-world graph's x1.23 at 0.87 the break-even interpolates to ~0.79; tables are only BUILT for rows of 10+ entries.)
        return staged_fraction >= 0.80 || (f.mean_ceil() >= 32 && staged_fraction >= 0.62);
    }
    // (128-column tiles: 0.60 until the blocks of graphs with short rows grew to 112 rows — LFR mu = 0.1 then stages 0.552 of its entries and
    //  runs 167 us against 172-178 through the streaming kernels, LFR mu = 0.3 at 0.409 loses 17 %: profiles/r05/kernel_ab_rows_rule.log)
    // Short rows gain from the record stream itself (no row pointers, no per-row round trips), not only from the rows in LDS: planted
    // communities of mean degree 4 / 5 / 6 run x1.05 / x1.06 / x1.09 ahead of the streaming kernels at shares of 0.45 / 0.47 / 0.53
    // (profiles/r05/staged_degree_sweep_retuned.log; mean degree 3: level) where LFR's 16-entry rows lose 17 % at 0.41.
    if (tile128 && f.mean_ceil() <= 8 && staged_fraction >= 0.42) return true;
    return staged_fraction >= (tile128 ? 0.55 : 0.42);  // (128-column tiles : 256-column tiles)
}

// Rows per block of the staged-rows kernel (the kernel does not need the number: blocks are tasks + a staging list; the plan cuts them).
// The shape's default — as many rows as LDS holds staged rows, 96 / 64 at 128- / 256-column tiles — dates from round 3's walk. With the
// record stream (profiles/r05/staged_rows_per_block.log, us at 80 / 96 / 112 / 128 rows, N = 128): com-Amazon-shaped (mean degree 5.5)
// 91.8 / 91.3 / 88.0 / 88.6, LFR mu = 0.1 (16) 177.7 / 176.5 / 170.1 / 168.9, geometric and small-world (11-12) flat, products-shaped
// (50) 2792 / 2789 / 2880 / 2901; 256-column tiles at 48 / 64 / 80 rows: com-Amazon-shaped 180.8 / 175.3 / 176.5, geometric 341.9 /
// 335.6 / 341.8, products-shaped 5271 / 5502 / 5600. Short rows want more of them per block, long rows fewer.
// Padded-record kernel (spmm_records.hip): asked for by name, or (AUTO) where it was measured ahead of the streaming kernels and the
// lane-group staged kernel — profiles/r06/records_audit_before_rule.log, time AUTO-before / records on the clustered order:
//   com-Amazon-shaped communities (mean degree 5.5, modelled hits 0.67-0.69): N = 16 x1.23, N = 32 x1.25, N = 64 x1.08
//   N = 16 at any mean degree: products-shaped communities (mean 50) x1.20, geometric (12) x1.87, small-world (11) x1.67
//   N = 32 / 64 with rows of 10+ entries: level with or behind the lane-group staged kernel (x0.99 / x0.83 products-shaped, x0.99 / x0.76
//     geometric, x0.92 / x0.71 small-world) — its tables are kept there, and these are not built
//   LFR (mean degree 16, longest row 306: rows of very different lengths share a task, 28-46 % of the slots filled): x0.86 / x0.96 / x0.95
//     at mu = 0.1, x0.78 / x0.94 / x0.86 at mu = 0.3 — keep_record_tables drops them by their slot fill
//   an order that does not hit L2 (structureless graph, modelled 0.19-0.28): x0.66-0.84; the storage order of a scrambled graph: level
//     with the plain call. Hence: clustered order kept and modelled at >= 0.60 hits.
//   Widths that are not multiples of 4 (4-byte-aligned vectors; the streaming kernels fall to one float per lane there) under the same
//     rule, profiles/r06/records_anywidth.log: com-Amazon-shaped N = 7 / 30 / 41 / 47 / 62 x1.15 / x1.19 / x1.32 / x1.35 / x1.20;
//     products-shaped N = 7 / 10 / 15 x1.50 / x1.55 / x1.42 (N = 20 ... 62 level: mean degree 50, not taken).
//   `order_hits`: the modelled L2 hits of the order the plan PROCESSES the rows in — the clustered order if it was kept, the storage order
//   if the model judged it as good (a matrix that arrives clustered: profiles/r06/records_preordered.log); negative = never modelled.
bool want_record_tables(const PlanFacts& f, double order_hits) {
    if (f.kernel_choice == GESPMM_PLAN_KERNEL_RECORDS) return true;
    if (f.kernel_choice != GESPMM_PLAN_KERNEL_AUTO || f.variant != GESPMM_VARIANT_AUTO) return false;
    if (order_hits < 0.60 || f.nnz < (1 << 20) || f.M <= 0) return false;
    return f.N <= 16 || (double)f.nnz / (double)f.M <= 8.0;
}

// ... and kept when enough of their slots carry an entry (tasks cut by work, profiles/r06/records_audit.log): N <= 16 com-Amazon-shaped 0.53,
// geometric 0.57, products-shaped 0.62, small-world 0.69 win, LFR 0.42 loses; N = 32 / 64 com-Amazon-shaped 0.59 / 0.62 win.
bool keep_record_tables(const PlanFacts& f, double slot_fill) {
    if (f.kernel_choice != GESPMM_PLAN_KERNEL_AUTO) return true;
    return slot_fill >= (f.N <= 16 ? 0.48 : 0.50);
}

// Batches a task is cut at: a multiple of the mean number of 8-entry pieces per row — short tasks keep the grid several generations deep,
// longer ones pack rows of different lengths better (a chain ends within one row of the task's longest). Measured, best T by graph
// (profiles/r06/records_sweep2.log, records_sweep3.log; mean pieces per row in brackets):
//   N = 32 / 64   com-Amazon-shaped [1.2] 3-4 / 3 · small-world, geometric [1.9-2.0] 8 / 5-12 · LFR [2.5] 8-24 / 8-16 · products-shaped [6.8] 16-24 / 16
//   N = 16        com-Amazon-shaped 4 · small-world 3 (5: +12 %) · geometric 3-5 · products-shaped 16 (12: +4 %) · LFR flat
int records_batches_per_task(const PlanFacts& f) {
    double mp = f.M > 0 ? (double)f.nnz / (8.0 * (double)f.M) + 0.5 : 1.0;
    if (mp < 1.0) mp = 1.0;
    int t = (int)((f.N <= 16 ? 1.6 : 3.0) * mp + 0.5);
    return t < 3 ? 3 : (t > 24 ? 24 : t);
}

int staged_rows_for(const PlanFacts& f, int shape_rows, int shape_waves) {
    const int sclass = staged_kernel_class(f.M, f.K, f.N);
    if (shape_waves != kStagedMaxWaves || (sclass != kStagedTuned && sclass != kStagedGeneral)) return shape_rows;  // (narrow widths, experiment shapes)
    const int64_t mean = f.mean_ceil();
    const int tc = staged_tile_class(f.N);
    if (tc == 128) return mean <= 24 ? 112 : shape_rows;
    if (tc == 256) return mean > 32 ? 48 : shape_rows;
    return shape_rows;  // (64-column tiles: the general kernel's default)
}

// Which streaming kernel a clustered plan launches (AUTO rule + the caller's choice).
//   segmented-stream (one continuous gather stream per lane group): ahead of the batch kernel on clustered matrices with
//     longer rows at one column tile (products-shaped communities, N = 128: 3.95 vs 4.37 ms; N = 16: 1.06 vs 1.17 ms; N = 32:
//     1.35 vs 1.38), behind at N = 64 (2.25 vs 2.03);
//   batch-stream otherwise — on short rows the two are within 2 % of each other at N >= 128 (com-Amazon stand-ins: 138.5 vs
//     140.1 us at 128, 267.7 vs 261.8 at 256, 574.6 vs 570.4 at 512) and the batch kernel is far ahead below (N = 64: 61 vs 91 us)
//     and on small graphs (pubmed N = 128: 9.6 vs 13.2 us) — and whenever long rows are split (run_spmm decides that).
//   (profiles/r02/plan_seg_widths.log; dense clustered graphs, mean degree in the hundreds: segmented also at N = 256 and 512 —
//   6.36 vs 7.03 ms and 15.5 vs 16.8 ms on the reddit-sized community graph, profiles/r03/dense_community_audit.log)
//   Hold-out audit (round 4, profiles/r04/holdout_audit.log): with mean degree 51 but only 0.52-0.57 modelled hits (LFR, mu = 0.3) the batch
//   kernel is 6-8 % ahead at N = 32 / 128 where the planted-community graph (0.84 hits) has the segmented one 7-11 % ahead — the
//   continuous stream pays off when most gathers are L2 hits: 0.70 asked for, not 0.40. At 129-512 columns the segmented kernel is
//   2-8 % ahead on mid-range hit rates (LFR mu = 0.3 / 0.5, dense LFR: 0.27-0.52) and 12-13 % behind on high ones with short rows
//   (geometric, small-world: 0.8-0.9).
bool prefer_segmented(const PlanFacts& f, double hits_after, int64_t N) {
    if (f.kernel_choice == GESPMM_PLAN_KERNEL_SEG_STREAM) return true;
    if (f.kernel_choice != GESPMM_PLAN_KERNEL_AUTO) return false;
    const int64_t mean_deg = f.mean_floor();
    if (!(f.nnz >= (1 << 20) && N % 4 == 0)) return false;
    // (mid-range hit rates beyond 128 columns: LFR mu = 0.3 / 0.5 and the dense LFR graph, 3-8 % at N = 256 and 2-7 % at 512;
    //  their mean degree is 15.7-15.9: the ceiling is asked for 16 here)
    if (N > 128 && N <= 512 && f.mean_ceil() >= 16 && hits_after >= 0.25 && hits_after < 0.60) return true;
    if (mean_deg < 16) return false;
    if ((N <= 32 || (N > 64 && N <= 128)) && hits_after >= 0.70) return true;
    if (mean_deg >= 128 && N > 64 && N <= 512 && hits_after >= 0.40) return true;
    return false;
}

// N <= 64: select.cpp gives every column its own lane (V = 1: 32 lanes per row at N = 32, two rows per gather instruction) — right
// for every graph measured with rows of 10+ entries (LFR N = 32: 78 vs 110 us with V = 4; products-shaped: 327 vs 365) and for
// miss-bound short rows (structureless com-Amazon stand-in: 48.6 vs 53.1). Short rows that HIT L2 are bound by the number of
// gather instructions instead: V = 4 carries 8 rows per instruction — com-Amazon-shaped communities, mean degree 5.5: 37.1 vs
// 44.6 us at N = 32, 49.6 vs 62.5 at N = 64 (profiles/r04/narrow_vec_rule.log). Same condition as the shallow unroll above.
bool narrow_vec4(const PlanFacts& f, double hits_after, int64_t N) {
    return f.variant == GESPMM_VARIANT_AUTO && N <= 64 && N >= 16 && N % 4 == 0 && hits_after >= 0.40 && f.mean_ceil() <= 8 &&
           f.nnz >= (1 << 20);
}

// The clustered edge walk pays a scatter pass at the end: worth it where the order is modelled to hit L2 for >= 40 % of the
// gathers and the rows are >= 256 bytes (com-Amazon-shaped communities, N = 128: 114 vs 151 us COO / 167 us CSR; on the
// structureless graph or at N = 41 it is equal or slower — profiles/r02/sddmm_plan.log). Otherwise short rows take the COO
// form on row ids expanded once (the CSR form spends a row search per wavefront: 4-18 %, profiles/r03/sddmm_audit.log) and
// long rows the CSR call, whose row-walking / cache-blocked forms need no row ids at all.
int sddmm_route(const PlanFacts& f, bool reordered, double hits_after, int64_t N) {
    if (!reordered || hits_after < 0.40 || N < 64) return (f.M > 0 && f.nnz / f.M < 32) ? 1 : 0;
    return 2;
}

}  // namespace gespmm

namespace {
constexpr int64_t kQueryBytesV1 = (int64_t)offsetof(gespmm_plan_policy_query, expected_launches);
constexpr int64_t kAnswerBytesV1 = (int64_t)offsetof(gespmm_plan_policy_answer, cost_skipped);
}  // namespace

extern "C" int gespmm_plan_wants_warmup(int64_t M, int64_t K, int64_t nnz, int64_t N, int32_t expected_launches) {
    if (M < 0 || K < 0 || nnz < 0 || N < 0 || expected_launches < 0) return GESPMM_EINVAL;
    return gespmm::analysis_could_pay(M, K, nnz, N, expected_launches) ? 1 : 0;
}

int gespmm_plan_policy(const gespmm_plan_policy_query* q, gespmm_plan_policy_answer* a) {
    return gespmm_plan_policy_v2(q, kQueryBytesV1, a, kAnswerBytesV1);
}

extern "C" int gespmm_plan_policy_v2(const gespmm_plan_policy_query* q_in, int64_t q_bytes, gespmm_plan_policy_answer* a_out, int64_t a_bytes) {
    if (!q_in || !a_out || q_bytes < kQueryBytesV1 || a_bytes < kAnswerBytesV1 || q_bytes % 4 != 0 || a_bytes % 4 != 0) return GESPMM_EINVAL;
    gespmm_plan_policy_query qq;
    std::memset(&qq, 0, sizeof qq);
    qq.wedge_probe = -1.0;
    if (q_bytes < (int64_t)offsetof(gespmm_plan_policy_query, cold_start) + 4) qq.cold_start = 0;
    if (q_bytes < (int64_t)offsetof(gespmm_plan_policy_query, record_slot_fill) + 8) qq.record_slot_fill = -1.0;  // (a 0.2 caller: unknown)
    std::memcpy(&qq, q_in, (size_t)(q_bytes < (int64_t)sizeof qq ? q_bytes : (int64_t)sizeof qq));
    if (q_bytes < (int64_t)offsetof(gespmm_plan_policy_query, wedge_probe) + 8) qq.wedge_probe = -1.0;
    if (q_bytes < (int64_t)offsetof(gespmm_plan_policy_query, cold_start) + 4) qq.cold_start = 0;
    gespmm_plan_policy_answer aa;
    std::memset(&aa, 0, sizeof aa);
    const gespmm_plan_policy_query* q = &qq;
    gespmm_plan_policy_answer* a = &aa;

    if (q->M < 0 || q->K < 0 || q->nnz < 0 || q->N < 0 || q->expected_launches < 0) return GESPMM_EINVAL;
    if (q->variant < GESPMM_VARIANT_AUTO || q->variant >= GESPMM_NUM_VARIANTS || q->reorder < 0 || q->reorder > 2) return GESPMM_EINVAL;
    // the kernel values gespmm_plan_create accepts (2 and 4 belonged to the removed opt-in kernels)
    if (q->kernel != GESPMM_PLAN_KERNEL_AUTO && q->kernel != GESPMM_PLAN_KERNEL_STREAM && q->kernel != GESPMM_PLAN_KERNEL_SEG_STREAM &&
        q->kernel != GESPMM_PLAN_KERNEL_STAGED && q->kernel != GESPMM_PLAN_KERNEL_RECORDS && q->kernel != GESPMM_PLAN_KERNEL_STAGED_SLABS)
        return GESPMM_EINVAL;
    if (q->analysis != GESPMM_PLAN_ANALYSIS_DEVICE && q->analysis != GESPMM_PLAN_ANALYSIS_HOST) return GESPMM_EINVAL;
    gespmm::PlanFacts f;
    f.M = q->M;
    f.K = q->K;
    f.nnz = q->nnz;
    f.N = q->N;
    f.variant = q->variant;
    f.max_degree = q->max_degree;
    f.reorder_mode = q->reorder;
    f.kernel_choice = q->kernel;
    f.host_analysis = q->analysis == GESPMM_PLAN_ANALYSIS_HOST;
    f.user_flags = q->flags;
    f.opt_task_entries = q->task_entries;
    f.opt_row_floor = q->row_floor;
    f.expected_launches = q->expected_launches;
    f.wedge_probe = q->wedge_probe;
    f.cold_start = q->cold_start != 0;
    gespmm::Selection sel;
    int max_vec = 4;
    while (max_vec > 1 && (q->N % max_vec) != 0) max_vec >>= 1;
    const int lr = gespmm::long_row_flags(q->M, q->nnz, q->max_degree, q->flags);
    if (gespmm::resolve_geometry(q->M, q->K, q->N > 0 ? q->N : 1, q->nnz, q->variant, max_vec, 0, 0, 0, 0, 0, lr, &sel) != 0)
        return GESPMM_EINVAL;
    f.sel_variant = sel.variant;
    f.slab_blocked = sel.geo.slab_blocked;
    f.tile_cols = (int64_t)sel.geo.group * sel.geo.vec * sel.geo.strips;
    const gespmm::AnalysisDecision ad = gespmm::decide_analysis(f);
    const bool keep = ad.analyse && gespmm::keep_clustered_order(f, ad, q->hits_before, q->hits_after);
    const gespmm::PlanKernelDecision kd = gespmm::choose_plan_kernel(f, q->hits_after);
    const int64_t Nl = q->N_launch > 0 ? q->N_launch : q->N;
    a->launch_flags = ad.launch_flags;
    a->analyse = ad.analyse;
    a->dense_try = ad.dense_try;
    a->keep_clustered = keep;
    a->task_entries = kd.task_entries;
    a->group_task_entries = kd.group_task_entries;
    a->row_floor = (int32_t)kd.row_floor;
    a->build_staged = keep && kd.build_staged;
    a->keep_staged = a->build_staged && gespmm::keep_staged_tables(f, q->staged_fraction);
    a->shallow_unroll = keep && kd.shallow_unroll;
    // what gespmm_plan_spmm_f32 launches at N_launch: the staged-rows kernel only at the plan's own width (its tables are made
    // for one width); at every other width the streaming rules apply — and the segmented-stream kernel is dropped wherever the
    // launch splits long rows (capi.cpp: run_spmm; the long-row pass belongs to the batch-stream kernel)
    const bool staged_here = a->keep_staged && Nl == q->N;
    bool seg = keep && !staged_here && gespmm::prefer_segmented(f, q->hits_after, Nl);
    if (seg) {
        gespmm::Selection sl;
        int mv = 4;
        while (mv > 1 && (Nl % mv) != 0) mv >>= 1;
        if (gespmm::resolve_geometry(q->M, q->K, Nl > 0 ? Nl : 1, q->nnz, q->variant, mv, 0, 0, 0, 0, 0,
                                     ad.launch_flags | GESPMM_FLAG_SEG_STREAM | GESPMM_FLAG_NO_SLAB_BLOCKED, &sl) == 0 &&
            sl.geo.split_long_rows)
            seg = false;
    }
    a->segmented = seg;
    a->narrow_vec4 = keep && !staged_here && gespmm::narrow_vec4(f, q->hits_after, Nl);
    a->sddmm_route = gespmm::sddmm_route(f, keep, q->hits_after, Nl);
    a->model_window = ad.model_window;
    a->model_sample = ad.model_sample;
    a->cost_skipped = ad.cost_skipped;
    a->est_gain_us = ad.cost.gain_us;
    a->est_cost_us = ad.cost.cost_us;
    a->cluster_levels = ad.analyse ? gespmm::cluster_levels_for(f) : 0;
    a->cluster_sweeps = ad.analyse ? gespmm::cluster_sweeps_for(f) : 0;
    {
        const gespmm::StagedShape sh = gespmm::staged_shape_any(q->N);
        a->staged_rows = sh.waves ? gespmm::staged_rows_for(f, sh.rows, sh.waves) : 0;
    }
    // the padded-record kernel (plan.cpp builds its tables after the staged ones: not beside tables that were kept)
    a->build_records = gespmm::records_serves(q->M, q->K, q->N, q->max_degree) && q->nnz > 0 && gespmm::want_record_tables(f, !ad.analyse ? -1.0 : (keep ? q->hits_after : q->hits_before)) &&
                       !a->keep_staged;
    a->keep_records = a->build_records && (q->record_slot_fill < 0.0 || gespmm::keep_record_tables(f, q->record_slot_fill));
    a->records_batches = gespmm::records_batches_per_task(f);
    a->slab_ranges = (a->keep_clustered || f.kernel_choice == GESPMM_PLAN_KERNEL_STAGED_SLABS) ? gespmm::slab_count_for(f) : 0;
    std::memcpy(a_out, &aa, (size_t)(a_bytes < (int64_t)sizeof aa ? a_bytes : (int64_t)sizeof aa));
    return 0;
}
