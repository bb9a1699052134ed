// select.h — host-only variant / launch-geometry selection (no HIP calls).
#pragma once
#include <stdint.h>

#include "spmm_kernels.h"

namespace gespmm {

struct Selection {
    int variant;   // resolved GESPMM_VARIANT_* (never AUTO)
    Geometry geo;
};

// What GESPMM_VARIANT_AUTO resolves to. Only bit-exact variants (0-4) are ever
// chosen automatically; the parallel-reduction variant is opt-in.
int auto_variant(int64_t M, int64_t nnz, int64_t N);

// Fill `out` for (shape, variant, optional overrides). max_vec is the widest
// vector the pointers/N allow (1, 2 or 4). Returns 0 or a GESPMM_E* code.
int resolve_geometry(int64_t M, int64_t K, int64_t N, int64_t nnz, int variant, int max_vec,
                     int cfg_vec, int cfg_strips, int cfg_group, int cfg_rows_per_wave, int cfg_slab_rows, int flags,
                     Selection* out);

}  // namespace gespmm
