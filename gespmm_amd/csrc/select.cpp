// select.cpp — variant and geometry selection.
//
// The reference dispatches on N alone (spmm_kernel.cu:186-206, 437-457):
//   N < 32 -> naive, 32 <= N < 64 -> CRC, N >= 64 -> CRC + CWM(2).
// On a 64-lane wavefront that table wastes lanes for every N < 64, so the
// geometry here is derived instead: the widest contiguous vector the row stride
// allows (V), then just enough lanes per row (W) to cover N, and the remaining
// 64/W lane groups take further rows.

#include "select.h"

#include "../../include/gespmm.h"

namespace gespmm {

static int pow2_ceil(int64_t x) {
    int p = 1;
    while (p < x && p < 64) p <<= 1;
    return p;
}

int auto_variant(int64_t M, int64_t nnz, int64_t N) {
    (void)nnz;
    // Up to 64 columns one lane per column already spans a whole wavefront-wide group;
    // coarsening would only shrink the group and put more rows on one wavefront, and a
    // 64-entry CSR tile of a long row feeds ONE group at a time. Measured (profiles/r01/
    // narrow_n_kernel_choice.log): V=1 is equal or faster for every graph at N <= 64,
    // by 30-58 % on the denser ones (products-like N=32, reddit-like N<=32).
    if (N <= 64) return GESPMM_VARIANT_CRC;
    // Just above one 256-column tile the second tile would be mostly empty yet walk every row again:
    // two strips per lane (one 512-column tile) instead — 24 % faster at N = 260, 3 % at 384 on a
    // 335 k-row graph; small graphs prefer the extra workgroups of two tiles (profiles/r01/width_audit.log).
    if (N % 4 == 0 && N > 256 && N <= 384 && M >= (1 << 17)) return GESPMM_VARIANT_CRC_CWM8;
    if (N % 4 == 0) return GESPMM_VARIANT_CRC_CWM4;
    if (N % 2 == 0) return GESPMM_VARIANT_CRC_CWM2;
    return GESPMM_VARIANT_CRC;
}

int resolve_geometry(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, int max_vec,
                     int cfg_vec, int cfg_strips, int cfg_group, int cfg_rows_per_wave, int cfg_slab_rows, int flags,
                     Selection* out) {
    if (variant == GESPMM_VARIANT_AUTO) {
        variant = auto_variant(M, nnz, N);
        // Opt-in (GESPMM_FLAG_ALLOW_REASSOCIATION): narrow N on dense rows is 1.3-2x faster with
        // lanes spread over the non-zeros (variant 5) — a re-ordered sum, within 1e-4, not bit-exact.
        if ((flags & kFlagAllowReassoc) && N <= 16 && nnz > 0 && M > 0 && nnz / M >= 32)
            variant = GESPMM_VARIANT_PARREDUCE;
    }
    Geometry g;
    g.reduce = kReduceSum;
    g.idx64 = ((flags & kFlagForceIdx64) != 0) || ((uint64_t)K * (uint64_t)N * 4ull >= (1ull << 32));
    g.strips = 1;
    switch (variant) {
        case GESPMM_VARIANT_NAIVE:
        case GESPMM_VARIANT_CRC: g.vec = 1; break;
        case GESPMM_VARIANT_CRC_CWM2: g.vec = 2; break;
        case GESPMM_VARIANT_CRC_CWM4: g.vec = 4; break;
        case GESPMM_VARIANT_CRC_CWM8: g.vec = 4; g.strips = 2; break;
        case GESPMM_VARIANT_PARREDUCE: g.vec = 1; break;
        default: return GESPMM_EINVAL;
    }
    if (cfg_vec) {
        if (cfg_vec != 1 && cfg_vec != 2 && cfg_vec != 4) return GESPMM_EINVAL;
        g.vec = cfg_vec;
    }
    if (cfg_strips) {
        if (cfg_strips != 1 && cfg_strips != 2) return GESPMM_EINVAL;
        g.strips = cfg_strips;
    }
    // Degrade to what alignment and N allow; results do not depend on V/S/W.
    if (g.vec > max_vec) g.vec = max_vec;
    if (g.strips == 2 && g.vec != 4) g.strips = 1;
    // Widths with no dwordx4 access (odd N, N = 2 mod 4, or unaligned operands) beyond one 64-lane tile: two
    // strips per lane halve the number of column tiles — each tile walks every entry of every row again
    // (reddit-like N = 65: 5.3 -> 3.9 ms; bench graph N = 513: 719 -> 647 us; no gain where scalar gathers are issue-bound, N = 127; profiles/r01/width_audit.log). Batch-stream, blocked and
    // long-row kernels only; W = 64.
    const bool wide_narrow_vec = g.vec < 4 && cfg_strips == 0 && cfg_group == 0 && N > 64 * g.vec &&
                                 variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE &&
                                 (flags & kFlagSegStream) == 0;
    if (wide_narrow_vec) g.strips = 2;

    if (variant == GESPMM_VARIANT_PARREDUCE) {
        const int64_t avg = (nnz > 0 && M > 0) ? (nnz + M - 1) / M : 16;
        g.group = pow2_ceil(avg);
    } else {
        const int64_t per_lane = (int64_t)g.vec * g.strips;
        g.group = pow2_ceil((N + per_lane - 1) / per_lane);
    }
    if (g.group < 4) g.group = 4;
    if (cfg_group) {
        if (cfg_group < 4 || cfg_group > 64 || (cfg_group & (cfg_group - 1))) return GESPMM_EINVAL;
        g.group = cfg_group;
    }
    // Rows per wavefront of the batch-stream kernel. Measured over the graph families at
    // N = 32..512 (profiles/r01/rows_per_wave_sweep.log): the best task size is ~12 KB of B
    // gathered per wavefront task — 96 CSR entries at N = 32, 24 at N = 128 — never fewer
    // than 16 entries; larger tasks lose 5-17 % (coarser dynamic balance over the CUs).
    // (grid depth: see below)
    const int rows_in_flight = 64 / g.group;
    const bool b_resident = (uint64_t)K * (uint64_t)N * 4ull <= (8ull << 20);
    {
        const int64_t avg = (nnz > 0 && M > 0) ? (nnz + M - 1) / M : 8;
        const int64_t tile_cols = (int64_t)g.group * g.vec * g.strips;
        const int64_t entry_bytes = 4 * (N < tile_cols ? N : tile_cols);
        int64_t target = (12 << 10) / (entry_bytes > 0 ? entry_bytes : 4);
        if (target < 16) target = 16;
        if (target > 96) target = 96;
        // B small enough to live in the L2s: the kernel is issue-bound, not fabric-bound, and the
        // per-task prologue counts — larger tasks (73-82 vs 97 us at K = 2048, cache_regime_sweep*.log)
        if (b_resident) target = 128;
        int rpw = kMaxRowsPerWave;
        while (rpw > rows_in_flight && (int64_t)rpw * avg > target) rpw >>= 1;
        // ... but never fewer wavefronts than the chip has slots (256 CUs x 32): one full round
        while (rpw > rows_in_flight && M / rpw < 8192) rpw >>= 1;
        if (rpw < rows_in_flight) rpw = rows_in_flight;
        g.rows_per_wave = rpw;
    }
    // Rows per lane group of the segmented-stream kernel: ~64 CSR entries per group task
    // (two 32-entry tiles), again keeping the grid several waves deep.
    {
        const int64_t avg = (nnz > 0 && M > 0) ? (nnz + M - 1) / M : 8;
        int rpg = kMaxRowsPerWave;
        while (rpg > 1 && (int64_t)rpg * avg > 64) rpg >>= 1;
        while (rpg > 1 && M / ((int64_t)rpg * rows_in_flight) < 4 * 8192) rpg >>= 1;
        if (b_resident && rpg > 4) rpg = 4;  // kernel_generations_cache_regimes.log: g4 58 us, g8 65
        g.rows_per_group = rpg;
    }
    if (cfg_rows_per_wave) {
        g.rows_per_wave = cfg_rows_per_wave;
        g.rows_per_group = cfg_rows_per_wave;
    }
    // Kernel generation. Wide groups (W >= 32: one or two rows per wavefront) run the
    // segmented-stream kernel; with many narrow groups per wavefront its per-entry row
    // flushes diverge between groups and the batch kernel (rows in lock-step, one store
    // phase per batch) is faster — measured in profiles/r01/kernel_generations.log.
    const int64_t avg_deg = (nnz > 0 && M > 0) ? (nnz + M - 1) / M : 8;
    // Cache blocking for dense graphs (one launch per ~6 MB column slab of B, see
    // spmm_kernels.hip): worth it when B per column tile is much larger than an L2 and a
    // row still carries enough work in every slab.
    {
        auto plan = [&](int64_t row_bytes, int64_t* slab_rows_out, int64_t* nslab_out) {
            int64_t slab_rows = cfg_slab_rows > 0 ? cfg_slab_rows : (6 << 20) / row_bytes;
            if (slab_rows < 64 && cfg_slab_rows <= 0) slab_rows = 64;
            if (slab_rows > 0x3fffffff) slab_rows = 0x3fffffff;
            const int64_t nslab = (K + slab_rows - 1) / slab_rows;
            *slab_rows_out = slab_rows;
            *nslab_out = nslab;
            // Measured (profiles/r01/slab_blocking.log, heuristic_audit.log, dense_width_audit.log): blocking wins
            // when the part of a row that falls into one slab still gathers >= ~10 KB (reddit-like: 13 KB at every
            // width; a 200 k-row degree-150 matrix: 4.5 KB, 27 % slower blocked) — with 6 MB slabs that is a matrix
            // density of ~0.16 %. Below it the per-slab read-modify-write of C outweighs the L2 hits. Also: B at
            // least 4 slabs, mean degree >= 64, row slices >= 256 B (a loss at N = 32).
            return nnz > 0 && M > 0 && row_bytes >= 256 && nslab >= 4 && nslab <= 4096 && avg_deg >= 64 &&
                   avg_deg * row_bytes >= 10240 * nslab;
        };
        int64_t slab_rows = 0, nslab = 0;
        bool dense;
        const bool forced = (flags & kFlagSlabBlocked) != 0;
        if (g.vec == 4 && N > 128 && cfg_strips == 0 && cfg_group == 0) {
            // The blocked path picks its own column tiling (a tile with few live columns costs as much as a full
            // one: every tile walks every entry). Column tiles are dealt to workgroups round-robin (tile = id mod
            // ntile) and workgroup ids to XCDs round-robin (id mod 8), so with 2, 4 or 8 FULL 128-column tiles
            // an XCD only ever sees one tile and its L2 holds slab_rows x 512 B: N = 256 / 512 / 1024 use
            // 512-byte tiles (N = 256: 9.96 -> 8.79 ms on reddit-like, slab_size_sweep_v2.log). Every other
            // width takes ceil(N / 256) tiles of 1 KB (N = 132: 8.9 -> 6.6 ms, N = 384: 23.8 -> 17.1,
            // N = 516: 43 -> 30, dense_width_audit.log).
            const int64_t ntile128 = (N + 127) / 128;
            const bool xcd_tiles = (N % 128 == 0) && (ntile128 == 2 || ntile128 == 4 || ntile128 == 8);
            dense = plan(xcd_tiles ? 512 : 1024, &slab_rows, &nslab);
            if (dense || forced) {
                g.group = xcd_tiles ? 32 : 64;
                g.strips = 1;
                if (variant == GESPMM_VARIANT_CRC_CWM8) variant = GESPMM_VARIANT_CRC_CWM4;  // report what runs
            }
        } else {
            dense = plan((int64_t)g.group * g.vec * g.strips * 4, &slab_rows, &nslab);
        }
        g.slab_rows = (int)slab_rows;
        g.K = K;
        g.slab_blocked = (forced || dense) && (flags & kFlagNoSlabBlocked) == 0 && nslab <= 65536 &&
                         variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE;
    }
    // (needs nnz to size its workspace: callers that pass nnz = -1 keep the strict chain)
    // The pass costs ~6.5 us of extra launches when there is no long row (profiles/r01/longrow_threshold_audit.log),
    // and the host cannot see the degrees. Always on from 2^23 entries; from 2^20 when the mean degree is >= 8
    // (RMAT-16..18: 1.8-5x faster with it; com-Amazon-shaped graphs, mean degree 5.5 and no hubs, keep their 150 us).
    // Smaller or sparser matrices with hub rows: GESPMM_FLAG_SPLIT_LONG_ROWS.
    const bool auto_split = nnz >= kLongRowMinNnz || (nnz >= (1 << 20) && M > 0 && nnz / M >= 8);
    g.split_long_rows = nnz > 0 && ((flags & kFlagSplitLongRows) != 0 || auto_split) && (flags & kFlagStrictOrder) == 0 &&
                        variant != GESPMM_VARIANT_NAIVE && variant != GESPMM_VARIANT_PARREDUCE;
    // A row is "long" when it dwarfs the average wavefront's work: 32x the mean degree,
    // at least kLongRowThreshold entries (reddit-like graphs, mean degree ~500, keep
    // their many 2k..15k-entry rows on the main kernel, which is faster on them).
    {
        int64_t thr = 32 * avg_deg;
        if (thr < kLongRowThreshold) thr = kLongRowThreshold;
        if (thr > 0x3fffffff) thr = 0x3fffffff;
        g.long_row_threshold = (int)thr;
    }
    // With task sizes tuned per kernel the batch kernel is equal or faster on every graph family
    // whose B exceeds the L2s (profiles/r01/rows_per_wave_sweep.log), and it implements the long-row
    // split. The segmented kernel keeps the short-row, L2-resident-B corner (58 vs 68 us,
    // kernel_generations_cache_regimes.log) and is otherwise opt-in (GESPMM_FLAG_SEG_STREAM).
    // ... and very short rows (mean degree <= 3: road-network-like), where the batch kernel spends a dependent round trip per
    // row pair and the continuous stream runs at copy speed (M = 335 k, N = 128, every row 1 / 2 / 3 entries: 64-69 / 76 / 98 us vs
    // 79-82 / 91 / 101 us; degrees 1..8 mixed, mean 2.75: 92 vs 102 us; profiles/r02/low_degree_floor.log)
    g.segmented = ((flags & kFlagSegStream) != 0 || (b_resident && g.group >= 32 && avg_deg <= 12) ||
                   (g.group >= 32 && avg_deg <= 3 && M >= (1 << 16))) &&
                  !g.split_long_rows && !(g.strips == 2 && g.vec < 4);
    // System-scope C stores (GESPMM_FLAG_SC1_STORE) stay opt-in: with B L2-resident and C not, they help when the reuse of B is
    // skewed (the BENCH graph's rows folded onto a 1 MB B: 81 -> 65 us; 4 MB: 86 -> 82) and cost when it is uniform and store-heavy
    // (uniform degree 6 over the same B: 54.5 -> 61.9 us; 73.7 -> 75.6) — profiles/r02/l2_resident_store_scope.log. The host
    // cannot tell the two apart without reading the matrix.
    g.sc1_store = false;
    out->variant = variant;
    out->geo = g;
    return 0;
}

}  // namespace gespmm
