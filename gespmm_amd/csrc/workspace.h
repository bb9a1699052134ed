// workspace.h — stream-ordered temporaries of the library (slab split points, long-row partials).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gespmm {

// Allocation ordered on `st`, from a memory pool the library owns (one per device, created on
// first use): the application's default pool keeps its own release policy, and the blocks are
// retained across synchronisations instead of going back to the driver after every epoch.
hipError_t workspace_alloc(void** ptr, size_t bytes, hipStream_t st);
hipError_t workspace_free(void* ptr, hipStream_t st);

}  // namespace gespmm
