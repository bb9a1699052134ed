// Can the SCALAR memory path pull the B rows that will miss L2 into L2 ahead of the vector gathers, so that the per-CU vector
// request slots only ever hold (short) L2 hits? Mixed gather stream: of the U = 8 rows a half-wavefront gathers per step, HOT come
// from a 2 MB table (L2 hits) and 8 - HOT from a 268 MB table (misses). PREFETCH = 1: before the gathers of step s, one
// s_load_dword per 128-byte line of the cold rows of step s + 1 (wave-uniform addresses, results discarded).
//   hipcc -O3 --offload-arch=gfx950 scalar_prefetch.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int HOT, int PREFETCH>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ hot, const float4* __restrict__ cold, int steps, float* __restrict__ out) {
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, l = lane & 31;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned s = gw * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    // rows of the NEXT step are generated one step ahead so that they can be prefetched
    unsigned nxt[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { s = s * 1664525u + 1013904223u; nxt[j] = s >> 8; }
    for (int it = 0; it < steps; ++it) {
        unsigned cur[U];
#pragma unroll
        for (int j = 0; j < U; ++j) cur[j] = nxt[j];
#pragma unroll
        for (int j = 0; j < U; ++j) { s = s * 1664525u + 1013904223u; nxt[j] = s >> 8; }
        unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (PREFETCH) {
#pragma unroll
            for (int j = HOT; j < U; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // both half-wavefronts' rows, all four lines of each
                    const unsigned row = (nxt[j] + h * 7919u) & ((1u << 19) - 1);
                    const unsigned long long pa = (unsigned long long)(reinterpret_cast<const char*>(cold) + (size_t)row * 512);
                    const unsigned long long p = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
                                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(pa & 0xffffffffu));
                    // issue + wait inside ONE asm block: the compiler must not touch the address SGPRs while the loads are
                    // outstanding (it does not track inline-asm SMEM), so this wavefront waits for its prefetch — the point is the
                    // vector request slots, which then only see L2 hits
                    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x80\n\ts_load_dword %2, %4, 0x100\n\ts_load_dword %3, %4, 0x180\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(p) : "memory");
                }
            }
        }
        float4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (j < HOT) {
                const unsigned row = (cur[j] + half * 7919u) & 4095u;
                v[j] = hot[(size_t)row * 32 + l];
            } else {
                const unsigned row = (cur[j] + half * 7919u) & ((1u << 19) - 1);
                v[j] = cold[(size_t)row * 32 + l];
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int HOT, int PREFETCH>
static void run(const float4* hot, const float4* cold, float* out) {
    const int wgs = 256 * 8, steps = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<HOT, PREFETCH>), dim3(wgs), dim3(256), 0, 0, hot, cold, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)wgs * 4 * steps * 8 * 1024.0;
    printf("hot %d of 8 rows from L2, %d from a 268 MB table, scalar prefetch %d: %7.3f ms  %6.2f TB/s\n", HOT, 8 - HOT, PREFETCH, ms, bytes / ms / 1e9);
}

int main() {
    float4 *hot, *cold; float* out;
    hipMalloc(&hot, (size_t)4096 * 512); hipMemset(hot, 0, (size_t)4096 * 512);
    hipMalloc(&cold, (size_t)(1 << 19) * 512); hipMemset(cold, 0, (size_t)(1 << 19) * 512);
    hipMalloc(&out, 64);
    run<8, 0>(hot, cold, out);
    run<6, 0>(hot, cold, out); run<6, 1>(hot, cold, out);
    run<5, 0>(hot, cold, out); run<5, 1>(hot, cold, out);
    run<4, 0>(hot, cold, out); run<4, 1>(hot, cold, out);
    run<0, 0>(hot, cold, out); run<0, 1>(hot, cold, out);
    return 0;
}
