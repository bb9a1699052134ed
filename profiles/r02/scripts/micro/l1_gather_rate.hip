// How fast can a CU pull 16-byte-per-lane gathers out of its L1 / the L2? (the ceiling of every SpMM kernel here in the
// cache-hit regime). Each wavefront issues U independent global_load_dwordx4 per step; lanes 0-31 read one 512-byte
// "row", lanes 32-63 another (MODE 0, like the SpMM kernels at N = 128), or all 64 lanes one 1024-byte row (MODE 1),
// rows picked pseudo-randomly from `rows` rows of the buffer. 4 adds per load keep the data live.
//   hipcc -O3 --offload-arch=gfx950 l1_gather_rate.hip -o /tmp/l1_gather_rate && /tmp/l1_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int U, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ buf, int rows_mask, int steps,
                                                       float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int half = lane >> 5;
    unsigned s = gw * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < steps; ++it) {
        float4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            s = s * 1664525u + 1013904223u;  // wave-uniform stream
            unsigned r = (s >> 8);
            size_t idx;
            if (MODE == 0) {
                const unsigned row = (r + half * 7919u) & rows_mask;  // 512-byte rows: 32 float4
                idx = (size_t)row * 32 + (lane & 31);
            } else {
                const unsigned row = r & (rows_mask >> 1);            // 1024-byte rows: 64 float4
                idx = (size_t)row * 64 + lane;
            }
            v[j] = buf[idx];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U, int MODE>
static void run(const float4* buf, int rows, float* out, int wgs, const char* name) {
    const int steps = 2000 / U * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gather_kernel<U, MODE>), dim3(wgs), dim3(256), 0, 0, buf, rows - 1, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * 4 * steps * U * 1024.0;
    printf("%-34s rows=%-7d (%8.3f MB) wgs=%-5d U=%d: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz\n", name, rows,
           rows * 512.0 / 1e6, wgs, U, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const int maxrows = 1 << 19;  // 256 MB
    float4* buf; float* out;
    hipMalloc(&buf, (size_t)maxrows * 512); hipMalloc(&out, 64);
    hipMemset(buf, 0, (size_t)maxrows * 512);
    for (int rows : {16, 64, 1024, 4096, 8192, 65536, 1 << 19}) {
        for (int wgs : {256 * 8, 256 * 4, 256 * 2}) {
            run<8, 0>(buf, rows, out, wgs, "two 512-B rows per load");
            run<4, 0>(buf, rows, out, wgs, "two 512-B rows per load");
            run<8, 1>(buf, rows, out, wgs, "one 1024-B row per load");
        }
    }
    return 0;
}
