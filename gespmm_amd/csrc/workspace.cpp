// workspace.cpp — see workspace.h. The only mutable global state of the library besides
// cached device properties: one hipMemPool_t per device, created lazily under a mutex.
#include "workspace.h"

#include <stdint.h>

#include <cstdlib>
#include <mutex>

namespace gespmm {

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_lock;
hipMemPool_t g_pool[kMaxDevices] = {};

hipError_t pool_for_current_device(hipMemPool_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_lock);
    if (!g_pool[dev]) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        e = hipMemPoolCreate(&pool, &props);
        if (e != hipSuccess) return e;
        // keep up to 4 GiB of freed blocks for reuse (RMAT-26 x N=256 needs ~1 GiB of partial rows)
        uint64_t keep = 4ull << 30;
        e = hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        if (e != hipSuccess) {
            (void)hipMemPoolDestroy(pool);
            return e;
        }
        g_pool[dev] = pool;
    }
    *out = g_pool[dev];
    return hipSuccess;
}
}  // namespace

static const bool g_no_pool = getenv("GESPMM_NO_POOL") != nullptr;  // debugging aid: plain hipMalloc / synchronised hipFree

hipError_t workspace_alloc(void** ptr, size_t bytes, hipStream_t st) {
    if (g_no_pool) return hipMalloc(ptr, bytes ? bytes : 1);
    hipMemPool_t pool = nullptr;
    if (pool_for_current_device(&pool) != hipSuccess) {
        (void)hipGetLastError();  // no explicit pools on this runtime: the device's default pool will do
        return hipMallocAsync(ptr, bytes ? bytes : 1, st);
    }
    return hipMallocFromPoolAsync(ptr, bytes ? bytes : 1, pool, st);
}

hipError_t workspace_free(void* ptr, hipStream_t st) {
    if (g_no_pool) {
        (void)hipStreamSynchronize(st);
        return hipFree(ptr);
    }
    return hipFreeAsync(ptr, st);
}

}  // namespace gespmm
