// spmm_stream_plan.hip — plan-mode instantiations of the two streaming kernels (task tables + row permutation, plan.cpp).
// A translation unit of its own: the plain kernels in spmm_kernels.hip stay exactly what they are without plans, and the
// two files compile side by side.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spmm_kernels.h"
#include "spmm_stream.h"

namespace gespmm {

hipError_t launch_spmm_stream_planned(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    if (!a.tasks || !a.perm) return hipErrorInvalidValue;
    return launch_spmm_stream_impl<true>(a, geo, st);
}

hipError_t launch_spmm_segstream_planned(const SpmmArgs& a, const Geometry& geo, hipStream_t st) {
    if (!a.gtasks || !a.perm) return hipErrorInvalidValue;
    return launch_spmm_segstream_impl<true>(a, geo, st);
}

}  // namespace gespmm
