// From the pure-gather ceiling towards the SpMM inner loop, one ingredient at a time (table resident in L2: 4096 rows x 512 B):
//   MODE 0  addresses computed in registers, 4 adds per gather (= l1_gather_rate.hip, MODE 0)
//   MODE 1  row offsets read from LDS (ds_read -> v_add -> global_load), as the kernels stage colind
//   MODE 2  + a value per gather read from LDS, fused multiply-add instead of add
//   MODE 3  + every 4 steps the wavefront loads 64 new (offset, value) pairs from global memory (coalesced), publishes them to LDS
//   MODE 4  + every 8 steps a 512-byte row per half-wavefront is stored (the C row)
//   MODE 5  as 4, but finished rows are parked in LDS and stored 8 at a time (one burst per 64 steps)
//   hipcc -O3 --offload-arch=gfx950 l2_gather_steps.hip -o /tmp/l2s && /tmp/l2s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int OUTROWS_LOG2 = 21, int STORE = 0>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ buf, const unsigned* __restrict__ offs,
                                         const float* __restrict__ vals, int rows_mask, int steps, float4* __restrict__ out) {
    constexpr int U = 8;
    __shared__ unsigned s_off[4][2][32];
    __shared__ float s_val[4][2][32];
    __shared__ float4 s_rows[MODE == 5 ? 4 : 1][MODE == 5 ? 2 : 1][MODE == 5 ? 8 : 1][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, l = lane & 31;
    const int gw = blockIdx.x * 4 + wave;
    unsigned s = gw * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    const char* base = reinterpret_cast<const char*>(buf);
    size_t meta = ((size_t)gw * 64) % (1 << 20);
    unsigned po = offs[meta + lane];
    float pv = vals[meta + lane];
    s_off[wave][g][l] = po; s_val[wave][g][l] = pv;
    __builtin_amdgcn_wave_barrier();
    for (int it = 0; it < steps; ++it) {
        float4 v[U];
        float w[U];
        const int t = (it & 3) * U;
        if (MODE >= 3 && (it & 3) == 0) {
            s_off[wave][g][l] = po; s_val[wave][g][l] = pv;
            meta = (meta + 64) % (1 << 20);
            po = __builtin_nontemporal_load(offs + meta + lane);
            pv = __builtin_nontemporal_load(vals + meta + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            unsigned off;
            if (MODE == 0) {
                s = s * 1664525u + 1013904223u;
                off = (((s >> 8) + g * 7919u) & rows_mask) * 512u;
            } else {
                off = s_off[wave][g][t + j];
            }
            w[j] = (MODE >= 2) ? s_val[wave][g][t + j] : 1.0f;
            v[j] = *reinterpret_cast<const float4*>(base + off + l * 16);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (MODE >= 2) {
                acc.x = __builtin_fmaf(w[j], v[j].x, acc.x); acc.y = __builtin_fmaf(w[j], v[j].y, acc.y);
                acc.z = __builtin_fmaf(w[j], v[j].z, acc.z); acc.w = __builtin_fmaf(w[j], v[j].w, acc.w);
            } else {
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
        if (MODE == 4 && (it & 7) == 7) {
            float4* dst = out + ((size_t)(gw * 2 + g) * (steps / 8) + it / 8) % (1 << OUTROWS_LOG2) * 32 + l;
            f4 val = {acc.x, acc.y, acc.z, acc.w};
            if (STORE == 0) *dst = acc;
            else if (STORE == 1) __builtin_nontemporal_store(val, reinterpret_cast<f4*>(dst));
            else if (STORE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
            else if (STORE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(val) : "memory");
            acc = make_float4(0, 0, 0, 0);
        }
        if (MODE == 5 && (it & 7) == 7) {
            s_rows[wave][g][(it >> 3) & 7][l] = acc;
            acc = make_float4(0, 0, 0, 0);
            if (((it >> 3) & 7) == 7) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    out[((size_t)(gw * 2 + g) * (steps / 8) + (it / 8 - 7 + r)) % (1 << OUTROWS_LOG2) * 32 + l] = s_rows[wave][g][r][l];
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc;
}

// Short-lived wavefronts, as the library launches them: every wavefront does only `steps` steps (two 64-entry rows = 16 steps),
// behind a dependent prologue (a "row pointer" load whose value is needed to address the first tile), and stores its rows.
template <int STORE, int PERIOD = 8>
__global__ __launch_bounds__(256) void kshort(const float4* __restrict__ buf, const unsigned* __restrict__ offs, const float* __restrict__ vals,
                                              const int* __restrict__ ptrs, int steps, float4* __restrict__ out) {
    constexpr int U = 8;
    __shared__ unsigned s_off[4][2][32];
    __shared__ float s_val[4][2][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, l = lane & 31;
    const int gw = blockIdx.x * 4 + wave;
    const char* base = reinterpret_cast<const char*>(buf);
    size_t meta = (size_t)ptrs[gw & 0xffff] + ((size_t)gw * 64) % (1 << 20);  // dependent: pointer first, then the tile
    meta %= (1 << 20);
    unsigned po = __builtin_nontemporal_load(offs + meta + lane);
    float pv = __builtin_nontemporal_load(vals + meta + lane);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < steps; ++it) {
        float4 v[U];
        float w[U];
        const int t = (it & 3) * U;
        if ((it & 3) == 0) {
            s_off[wave][g][l] = po; s_val[wave][g][l] = pv;
            meta = (meta + 64) % (1 << 20);
            po = __builtin_nontemporal_load(offs + meta + lane);
            pv = __builtin_nontemporal_load(vals + meta + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            w[j] = s_val[wave][g][t + j];
            v[j] = *reinterpret_cast<const float4*>(base + s_off[wave][g][t + j] + l * 16);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            acc.x = __builtin_fmaf(w[j], v[j].x, acc.x); acc.y = __builtin_fmaf(w[j], v[j].y, acc.y);
            acc.z = __builtin_fmaf(w[j], v[j].z, acc.z); acc.w = __builtin_fmaf(w[j], v[j].w, acc.w);
        }
        if ((it % PERIOD) == PERIOD - 1) {
            float4* dst = out + ((size_t)(gw * 2 + g) * (steps / PERIOD) + it / PERIOD) % (1 << 21) * 32 + l;
            f4 val = {acc.x, acc.y, acc.z, acc.w};
            if (STORE == 0) *dst = acc;
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(val) : "memory");
            acc = make_float4(0, 0, 0, 0);
        }
    }
}

template <int STORE, int PERIOD = 8>
static void run_short(const float4* buf, const unsigned* offs, const float* vals, const int* ptrs, float4* out, int steps, const char* name) {
    const long total_steps = 2048L * 8 * 2048;  // same gather volume as run<>
    const int wgs = (int)(total_steps / (4L * steps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kshort<STORE, PERIOD>), dim3(wgs), dim3(256), 0, 0, buf, offs, vals, ptrs, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)wgs * 4 * steps * 8 * 1024.0;
    printf("short  %-64s %7.3f ms  %6.2f TB/s  (%d workgroups x %d steps)\n", name, ms, bytes / ms / 1e9, wgs, steps);
}

template <int MODE, int OUTROWS_LOG2 = 21, int STORE = 0>
static void run(const float4* buf, const unsigned* offs, const float* vals, int rows, float4* out, const char* name) {
    const int wgs = 256 * 8, steps = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, OUTROWS_LOG2, STORE>), dim3(wgs), dim3(256), 0, 0, buf, offs, vals, rows - 1, steps, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)wgs * 4 * steps * 8 * 1024.0;
    printf("MODE %d %-64s %7.3f ms  %6.2f TB/s\n", MODE, name, ms, bytes / ms / 1e9);
}

int main() {
    const int rows = 4096;
    float4* buf; float4* out; unsigned* offs; float* vals;
    hipMalloc(&buf, (size_t)rows * 512); hipMemset(buf, 0, (size_t)rows * 512);
    hipMalloc(&out, (size_t)(1 << 21) * 512);
    std::vector<unsigned> ho(1 << 20); std::vector<float> hv(1 << 20);
    unsigned s = 7;
    for (size_t i = 0; i < ho.size(); ++i) { s = s * 1664525u + 1013904223u; ho[i] = ((s >> 8) & (rows - 1)) * 512u; hv[i] = 0.5f; }
    hipMalloc(&offs, ho.size() * 4); hipMalloc(&vals, hv.size() * 4);
    hipMemcpy(offs, ho.data(), ho.size() * 4, hipMemcpyHostToDevice); hipMemcpy(vals, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    run<0>(buf, offs, vals, rows, out, "addresses in registers, adds");
    run<1>(buf, offs, vals, rows, out, "+ row offsets read from LDS");
    run<2>(buf, offs, vals, rows, out, "+ values from LDS, fused multiply-add");
    run<3>(buf, offs, vals, rows, out, "+ a new 64-entry (offset, value) tile from global memory every 4 steps");
    run<4>(buf, offs, vals, rows, out, "+ a 512-byte row stored per half-wavefront every 8 steps");
    run<5>(buf, offs, vals, rows, out, "  same rows parked in LDS, stored 8 at a time");
    run<4, 16>(buf, offs, vals, rows, out, "  MODE 4 with the stored rows wrapping inside 32 MB");
    run<4, 12>(buf, offs, vals, rows, out, "  MODE 4 with the stored rows wrapping inside 2 MB");
    run<4, 21, 1>(buf, offs, vals, rows, out, "  MODE 4 (1 GB of output), non-temporal store (nt)");
    run<4, 21, 2>(buf, offs, vals, rows, out, "  MODE 4 (1 GB of output), sc1 store");
    run<4, 21, 3>(buf, offs, vals, rows, out, "  MODE 4 (1 GB of output), sc0 sc1 store");
    run<4, 21, 4>(buf, offs, vals, rows, out, "  MODE 4 (1 GB of output), sc0 sc1 nt store");
    int* ptrs; hipMalloc(&ptrs, 65536 * 4); hipMemset(ptrs, 0, 65536 * 4);
    for (int steps : {16, 32, 64, 256, 2048}) run_short<1>(buf, offs, vals, ptrs, out, steps, "short-lived wavefronts, sc1 stores");
    run_short<0>(buf, offs, vals, ptrs, out, 16, "short-lived wavefronts, plain stores");
    run_short<0, 1>(buf, offs, vals, ptrs, out, 16, "short-lived, a row stored EVERY step (8-entry rows), plain stores");
    run_short<1, 1>(buf, offs, vals, ptrs, out, 16, "short-lived, a row stored EVERY step (8-entry rows), sc1 stores");
    run_short<0, 1>(buf, offs, vals, ptrs, out, 2048, "persistent, a row stored EVERY step, plain stores");
    run_short<1, 1>(buf, offs, vals, ptrs, out, 2048, "persistent, a row stored EVERY step, sc1 stores");
    return 0;
}
