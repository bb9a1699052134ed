// auto_plan.h — internal interface of the opt-in plan cache behind the stateless entry points (auto_plan.cpp).
#pragma once
#include <stdint.h>

struct gespmm_plan;

namespace gespmm {

bool auto_plan_enabled();  // one relaxed atomic load: the switch is off by default
// true: the call was served through a cached plan (*rc = its result); false: run the plain path
bool auto_plan_try(const int32_t* rowptr, const int32_t* colind, const float* val, const float* B, float* C, int64_t M, int64_t K,
                   int64_t N, int64_t nnz, int variant, int reduce, float empty, void* stream, int* rc);
bool plan_is_clustered(const gespmm_plan* p);  // plan.cpp: the plan kept a clustered order (else its launch is the plain call's)

// plan_device.hip: the analysis arena a matrix of this size will ask for, allocated now and kept for the first plan (within the cache cap)
int reserve_analysis_arena(int64_t M, int64_t K, int64_t nnz, void* stream);
// the process has built a plan (or run gespmm_init): the analysis kernels are loaded, the arena exists
bool analysis_is_warm();
void mark_analysis_warm();

}  // namespace gespmm
