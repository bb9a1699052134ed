// plan.h — internal interface between the C ABI (capi.cpp) and the plan object (plan.cpp).
#pragma once
#include <stdint.h>

#include "../../include/gespmm.h"
#include "spmm_kernels.h"

namespace gespmm {

struct PlanLaunch {
    const int32_t* tasks;  // int4 per task (device)
    int32_t ntasks;
    const int32_t* perm;   // permuted row -> original row (device)
    const int32_t* gtasks; // int4 per lane-group task of the segmented-stream kernel (device), may be NULL
    int32_t ngtasks;
    bool prefer_segmented; // launch the segmented-stream kernel when the geometry allows it
};

int run_spmm(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C, int64_t M,
             int64_t K, int64_t N, int64_t nnz, int variant, const gespmm_launch_cfg* cfg, int reduce, float empty,
             void* stream, void* ws, int64_t ws_bytes, const PlanLaunch* pl, const LaunchGuard* guard = nullptr);
// A plan's product behind a launch guard (auto_plan.cpp): kNotGuardable — and nothing launched — when the plan's launch is more than
// one kernel (hub rows handed to the long-row pass, the cache-blocked path).
int plan_spmm_guarded(gespmm_plan* plan, const float* B, float* C, int64_t N, int reduce, float empty, void* stream, const LaunchGuard* guard);

}  // namespace gespmm
