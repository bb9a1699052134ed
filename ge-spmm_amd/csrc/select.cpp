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
    (void)M;
    (void)nnz;
    if (N % 4 == 0) return GESPMM_VARIANT_CRC_CWM4;
    if (N % 2 == 0) return GESPMM_VARIANT_CRC_CWM2;
    return GESPMM_VARIANT_CRC;
}

int resolve_geometry(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, int max_vec,
                     int cfg_vec, int cfg_strips, int cfg_group, int cfg_rows_per_wave, int flags, Selection* out) {
    if (variant == GESPMM_VARIANT_AUTO) variant = auto_variant(M, nnz, N);
    Geometry g;
    g.reduce = kReduceSum;
    g.crc = variant != GESPMM_VARIANT_NAIVE;
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
    // Rows per wavefront of the streaming kernel: aim at ~2 LDS tiles (128 CSR entries)
    // per wavefront so row pointers and tiles are fetched in full coalesced loads, but
    // keep >= ~4 wavefronts per wave slot of the chip (256 CUs x 32 slots) in the grid.
    const int rows_in_flight = 64 / g.group;
    {
        const int64_t avg = (nnz > 0 && M > 0) ? (nnz + M - 1) / M : 8;
        int rpw = kMaxRowsPerWave;
        while (rpw > rows_in_flight && (int64_t)rpw * avg > 128) rpw >>= 1;
        while (rpw > rows_in_flight && M / rpw < 4 * 8192) rpw >>= 1;
        if (rpw < rows_in_flight) rpw = rows_in_flight;
        g.rows_per_wave = rpw;
    }
    if (cfg_rows_per_wave) g.rows_per_wave = cfg_rows_per_wave;
    out->variant = variant;
    out->geo = g;
    return 0;
}

}  // namespace gespmm
