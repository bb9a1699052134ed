// spmm_hotrows.hip — EXPERIMENT (round 3): the B rows a block of clustered rows uses more than once, staged in LDS.
//
// A workgroup of WAVES wavefronts owns a BLOCK of consecutive rows of the plan's row-permuted matrix. The analysis (here:
// hotrows_time.py, torch on the device) lists per block the <= H columns used most often inside it (>= 2 uses) and rewrites
// the block's column indices: entry code >= 0 = column id (B row from memory), code < 0 = slot of the staged row. The
// workgroup copies the listed B rows into LDS once (coalesced, 512 B per row at N = 128), then every wavefront walks its task
// with the batch-stream loop of spmm_stream.h; each gather takes its row from LDS or from memory. One fp32 chain per output
// element in CSR order, one FMA per entry: same bits as every other variant.
//
// MODE 0: one `flat_load_dwordx4` per gather (the tile holds 64-bit generic addresses: LDS aperture or global);
// MODE 1: `ds_read_b128` or `global_load_dwordx4` under a per-lane predicate; MODE 2: nothing staged (all entries cold) —
// the same kernel structure as a yardstick.
// N = 128 only (W = 32 lanes x dwordx4, two rows per wavefront at a time).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kTile = 64;
using f4 = float __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4* gf4_ptr;  // global address space: `global_load`, not `flat_load`

struct HotArgs {
    const int32_t* rowptr;    // permuted matrix
    const int32_t* code;      // per entry
    const float* val;         // per entry
    const int32_t* perm;      // C row of permuted row i
    const int4* tasks;        // nblocks * WAVES: {first row, #rows, CSR begin, CSR end}
    const int32_t* hot_cols;  // nblocks * H
    const int32_t* nhot;      // nblocks
    const float* B;
    float* C;
    int nblocks;
    int b_bytes;  // K * N * 4 (MODE 3: < 0xFFFFF000)
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int H, int WAVES, int MAXR, int U, int MODE>
__global__ __launch_bounds__(WAVES * 64) void hot_kernel(HotArgs a) {
    constexpr int W = 32, G = 2;
    __shared__ f4 s_hot[(MODE == 2 ? 1 : H + (MODE == 3 ? 1 : 0)) * 32];  // MODE 3: row H is all zeros
    __shared__ uint64_t s_off[WAVES][kTile];
    __shared__ float s_val[WAVES][kTile];
    __shared__ int s_ptr[WAVES][MAXR + 1];
    __shared__ int s_perm[WAVES][MAXR];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane / W, l = lane % W;
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);

    const int4 t = a.tasks[blk * WAVES + wave];
    const int row_first = __builtin_amdgcn_readfirstlane(t.x);
    const int nrows = __builtin_amdgcn_readfirstlane(t.y);
    const int wb = __builtin_amdgcn_readfirstlane(t.z);
    const int we = __builtin_amdgcn_readfirstlane(t.w);

    int pc = 0;
    float pv = 0.0f;
    auto fetch_tile_regs = [&](int base) {
        const int p = base + lane;
        if (p < we) {
            pc = __builtin_nontemporal_load(a.code + p);
            pv = __builtin_nontemporal_load(a.val + p);
        }
    };
    fetch_tile_regs(wb);
    for (int i = lane; i <= nrows; i += 64) s_ptr[wave][i] = a.rowptr[row_first + i];
    for (int i = lane; i < nrows; i += 64) s_perm[wave][i] = a.perm[row_first + i];

    if constexpr (MODE != 2) {
        const int total = a.nhot[blk] * 32;
        const int32_t* hc = a.hot_cols + (size_t)blk * H;
        const f4* B4 = reinterpret_cast<const f4*>(a.B);
        for (int i0 = 0; i0 < total; i0 += WAVES * 64 * 4) {
            f4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // (clamped, not predicated: the four loads stay in flight together)
                const int i = i0 + u * WAVES * 64 + tid;
                const int ic = i < total ? i : total - 1;
                r[u] = B4[(size_t)hc[ic >> 5] * 32 + (ic & 31)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * WAVES * 64 + tid;
                if (i < total) s_hot[i] = r[u];
            }
        }
        if constexpr (MODE == 3) {
            if (tid < 32) s_hot[H * 32 + tid] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        __syncthreads();
    }

    const uint64_t lds_base = (uint64_t)(uintptr_t)(const void*)&s_hot[0];  // generic address of the staged rows
    const uint64_t Bbase = (uint64_t)(uintptr_t)a.B;
    const uint32_t cb = (uint32_t)l * 16u;
    auto publish_tile = [&]() {
        uint64_t addr;
        if constexpr (MODE == 3) {
            // low word: byte offset into B for the buffer load (staged: out of range = returns 0, no memory access);
            // high word: byte address of the LDS row (not staged: the zero row)
            const uint32_t lo = (pc < 0) ? 0xFFFFF000u : (uint32_t)pc * 512u;
            const uint32_t hi = (pc < 0) ? (uint32_t)(pc & 0x7fffffff) * 512u : (uint32_t)H * 512u;
            addr = (uint64_t)lo | ((uint64_t)hi << 32);
        } else if constexpr (MODE == 0) {
            addr = (pc < 0) ? lds_base + (uint64_t)(uint32_t)(pc & 0x7fffffff) * 512u : Bbase + (uint64_t)(uint32_t)pc * 512u;
        } else {
            // MODE 1/2: low 32 bits = byte offset, bit 63 = staged
            addr = (uint64_t)(uint32_t)(pc & 0x7fffffff) * 512u | ((pc < 0) ? (1ull << 63) : 0ull);
        }
        s_off[wave][lane] = addr;
        s_val[wave][lane] = pv;
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B), 0, a.b_bytes, 0x00020000);
    using i4 = int __attribute__((ext_vector_type(4)));
    auto gather = [&](uint64_t o) -> f4 {
        if constexpr (MODE == 3) {
            const i4 m = __builtin_bit_cast(i4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((uint32_t)o + cb), 0, 0));
            const i4 s = __builtin_bit_cast(i4, s_hot[((uint32_t)(o >> 32) >> 4) + l]);
            return __builtin_bit_cast(f4, m | s);
        } else if constexpr (MODE == 0) {
            return *reinterpret_cast<const f4*>((const char*)(uintptr_t)o + cb);  // flat load
        } else if constexpr (MODE == 2) {
            return *(gf4_ptr)(Bbase + (o & 0x7fffffffffffull) + cb);
        } else {
            const uint64_t off = o & 0x3fffffffffffffffull;
            if ((int64_t)o < 0) return s_hot[(uint32_t)(off >> 4) + l];
            return *(gf4_ptr)(Bbase + off + cb);
        }
    };

    int t0 = wb;
    publish_tile();
    fetch_tile_regs(t0 + kTile);
    wave_lds_sync();

    for (int b = 0; b < nrows; b += G) {
        const int r = b + g;
        const bool rowok = r < nrows;
        int lb = 0, hb = 0;
        if (rowok) {
            lb = s_ptr[wave][r];
            hb = s_ptr[wave][r + 1];
        }
        const int be = __builtin_amdgcn_readfirstlane(s_ptr[wave][(b + G < nrows) ? b + G : nrows]);
        f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (;;) {
            const int tend = t0 + kTile;
            int k = (lb > t0 ? lb : t0) - t0;
            const int ke = (hb < tend ? hb : tend) - t0;
            for (; k + U <= ke; k += U) {
                uint64_t off[U];
                float v[U];
                f4 bv[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    off[j] = s_off[wave][k + j];
                    v[j] = s_val[wave][k + j];
                }
                if constexpr (MODE == 1) {
                    // staged rows first (LDS latency), then the memory gathers into the other lanes of the same registers:
                    // all of a step's global loads are in flight together
#pragma unroll
                    for (int j = 0; j < U; ++j)
                        if ((int64_t)off[j] < 0) bv[j] = s_hot[(uint32_t)((off[j] & 0x3fffffffffffffffull) >> 4) + l];
#pragma unroll
                    for (int j = 0; j < U; ++j)
                        if ((int64_t)off[j] >= 0) bv[j] = *(gf4_ptr)(Bbase + off[j] + cb);
                } else {
#pragma unroll
                    for (int j = 0; j < U; ++j) bv[j] = gather(off[j]);
                }
#pragma unroll
                for (int j = 0; j < U; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(v[j], bv[j][i], acc[i]);
            }
            const int rem = ke - k;
            if (rem > 0) {
                uint64_t off[U - 1];
                float v[U - 1];
                f4 bv[U - 1];
#pragma unroll
                for (int j = 0; j < U - 1; ++j) {
                    const int kj = k + ((j < rem) ? j : rem - 1);
                    off[j] = s_off[wave][kj];
                    v[j] = s_val[wave][kj];
                }
                // (every slot gathers — slots past `rem` re-read the last entry, same lines — so the loads are straight-line code)
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < U - 1; ++j)
                        if ((int64_t)off[j] < 0) bv[j] = s_hot[(uint32_t)((off[j] & 0x3fffffffffffffffull) >> 4) + l];
#pragma unroll
                    for (int j = 0; j < U - 1; ++j)
                        if ((int64_t)off[j] >= 0) bv[j] = *(gf4_ptr)(Bbase + off[j] + cb);
                } else {
#pragma unroll
                    for (int j = 0; j < U - 1; ++j) bv[j] = gather(off[j]);
                }
#pragma unroll
                for (int j = 0; j < U - 1; ++j)
                    if (j < rem) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(v[j], bv[j][i], acc[i]);
                    }
            }
            if (be <= tend) break;
            wave_lds_sync();
            t0 = tend;
            publish_tile();
            fetch_tile_regs(t0 + kTile);
            wave_lds_sync();
        }
        if (rowok) {
            float* Crow = a.C + (size_t)s_perm[wave][r] * 128u + l * 4;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(Crow), "v"(acc) : "memory");
        }
    }
}


// ----------------------------------------------------------------------------- MODE 4: the scalar-stream kernel
// One wavefront = one task = one row at a time; a B row of N = 128 is 64 lanes x dwordx2. Everything that is the same for
// the 64 lanes — the CSR stream (column, value), row ends, C row ids — lives in SGPRs and comes through the scalar cache;
// a gather is `global_load_dwordx2 v, v_lane_offset, s[base]` with the base computed on the scalar unit; the multiply-add takes
// the value from an SGPR. No LDS, no per-lane address arithmetic, ~30 VGPRs.
typedef const __attribute__((address_space(4))) int32_t* cint_ptr;
typedef const __attribute__((address_space(4))) float* cflt_ptr;
using f2 = float __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f2* gf2_ptr;

template <int U>
__global__ __launch_bounds__(256) void scalar_kernel(HotArgs a, int ntasks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
    if (task >= ntasks) return;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr code = (cint_ptr)(uintptr_t)a.code;
    cflt_ptr val = (cflt_ptr)(uintptr_t)a.val;
    const uint64_t Bbase = (uint64_t)(uintptr_t)a.B;
    const uint32_t loff = (uint32_t)lane * 8u;

    int cur = 0;
    int rend = rowptr[1], rend_next = rowptr[nrows > 1 ? 2 : 1];
    int crow = perm[0], crow_next = perm[nrows > 1 ? 1 : 0];
    f2 acc = {0.0f, 0.0f};
    auto flush = [&]() {
        float* Crow = a.C + (size_t)crow * 128u;
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(acc), "s"(Crow) : "memory");
        acc = f2{0.0f, 0.0f};
        ++cur;
        rend = rend_next;
        crow = crow_next;
        const int nx = (cur + 1 < nrows) ? cur + 1 : nrows - 1;
        rend_next = rowptr[nx + 1];
        crow_next = perm[nx];
    };
    int k = wb;
    for (; k + U <= we; k += U) {  // whole chunks: no per-entry bounds
        int c[U];
        float v[U];
        f2 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) bv[j] = *(gf2_ptr)(Bbase + (uint64_t)(uint32_t)c[j] * 512u + loff);
        if (k + U <= rend) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        }
    }
    if (k < we) {  // last chunk (arrays are padded by U entries; slots past the end gather row 0 and are not summed)
        int c[U];
        float v[U];
        f2 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int cj = (k + j < we) ? c[j] : 0;
            bv[j] = *(gf2_ptr)(Bbase + (uint64_t)(uint32_t)cj * 512u + loff);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (k + j < we) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        }
    }
    while (cur < nrows) flush();
}


// MODE 8/9: the same with 32 ACTIVE lanes x dwordx4 per wavefront (does the address path charge per lane or per instruction?)
template <int U>
__global__ __launch_bounds__(256) void scalar_half_kernel(HotArgs a, int ntasks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + wave;
    if (task >= ntasks || lane >= 32) return;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr code = (cint_ptr)(uintptr_t)a.code;
    cflt_ptr val = (cflt_ptr)(uintptr_t)a.val;
    const uint64_t Bbase = (uint64_t)(uintptr_t)a.B;
    const uint32_t loff = (uint32_t)lane * 16u;

    int cur = 0;
    int rend = rowptr[1], rend_next = rowptr[nrows > 1 ? 2 : 1];
    int crow = perm[0], crow_next = perm[nrows > 1 ? 1 : 0];
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto flush = [&]() {
        float* Crow = a.C + (size_t)crow * 128u;
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(loff), "v"(acc), "s"(Crow) : "memory");
        acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
        ++cur;
        rend = rend_next;
        crow = crow_next;
        const int nx = (cur + 1 < nrows) ? cur + 1 : nrows - 1;
        rend_next = rowptr[nx + 1];
        crow_next = perm[nx];
    };
    int k = wb;
    for (; k + U <= we; k += U) {  // whole chunks: no per-entry bounds
        int c[U];
        float v[U];
        f4 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) bv[j] = *(gf4_ptr)(Bbase + (uint64_t)(uint32_t)c[j] * 512u + loff);
        if (k + U <= rend) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
                acc[2] = __builtin_fmaf(v[j], bv[j][2], acc[2]);
                acc[3] = __builtin_fmaf(v[j], bv[j][3], acc[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
                acc[2] = __builtin_fmaf(v[j], bv[j][2], acc[2]);
                acc[3] = __builtin_fmaf(v[j], bv[j][3], acc[3]);
            }
        }
    }
    if (k < we) {  // last chunk (arrays are padded by U entries; slots past the end gather row 0 and are not summed)
        int c[U];
        float v[U];
        f4 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int cj = (k + j < we) ? c[j] : 0;
            bv[j] = *(gf4_ptr)(Bbase + (uint64_t)(uint32_t)cj * 512u + loff);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (k + j < we) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
                acc[2] = __builtin_fmaf(v[j], bv[j][2], acc[2]);
                acc[3] = __builtin_fmaf(v[j], bv[j][3], acc[3]);
            }
        }
    }
    while (cur < nrows) flush();
}



// ----------------------------------------------------------------------------- MODE 6: scalar-stream + staged rows
// The scalar-stream kernel inside a workgroup that has staged its block's hot rows: the entry's code sits in an SGPR, so
// "staged or not" is a SCALAR branch — a staged entry is one `ds_read_b64` and issues no vector memory instruction at all.
template <int H, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void scalar_hot_kernel(HotArgs a) {
    __shared__ f2 s_hot[H * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);
    const int task = blk * WAVES + wave;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr code = (cint_ptr)(uintptr_t)a.code;
    cflt_ptr val = (cflt_ptr)(uintptr_t)a.val;
    const uint64_t Bbase = (uint64_t)(uintptr_t)a.B;
    const uint32_t loff = (uint32_t)lane * 8u;
    {
        const int total = ((cint_ptr)(uintptr_t)a.nhot)[blk] * 32;  // f4 elements
        const int32_t* hc = a.hot_cols + (size_t)blk * H;
        const f4* B4 = reinterpret_cast<const f4*>(a.B);
        f4* s_hot4 = reinterpret_cast<f4*>(s_hot);
        for (int i0 = 0; i0 < total; i0 += WAVES * 64 * 4) {
            f4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * WAVES * 64 + tid;
                const int ic = i < total ? i : total - 1;
                r[u] = B4[(size_t)hc[ic >> 5] * 32 + (ic & 31)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * WAVES * 64 + tid;
                if (i < total) s_hot4[i] = r[u];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0);  // (the compiler's scoreboard is clean when the assembly gathers start)
    }
    int cur = 0;
    int rend = rowptr[1], rend_next = rowptr[nrows > 1 ? 2 : 1];
    int crow = perm[0], crow_next = perm[nrows > 1 ? 1 : 0];
    if (nrows == 0) return;
    f2 acc = {0.0f, 0.0f};
    auto flush = [&]() {
        float* Crow = a.C + (size_t)crow * 128u;
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(acc), "s"(Crow) : "memory");
        acc = f2{0.0f, 0.0f};
        ++cur;
        rend = rend_next;
        crow = crow_next;
        const int nx = (cur + 1 < nrows) ? cur + 1 : nrows - 1;
        rend_next = rowptr[nx + 1];
        crow_next = perm[nx];
    };
    // One gather = a scalar branch around ONE of {ds_read_b64, global_load_dwordx2} in inline assembly: the compiler's own
    // placement put `s_waitcnt vmcnt(0)` in front of every LDS read (it cannot see that the two never write the same
    // register in the same pass). The results are awaited by one explicit s_waitcnt per chunk.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)s_hot + loff;
    auto gather = [&](int cj, f2& d) {
        const uint32_t rowb = (uint32_t)(cj & 0x7fffffff) * 512u;
        const uint64_t base = Bbase + (uint64_t)rowb;
        const uint32_t la = lds0 + rowb;
        asm volatile(
            "s_cmp_lt_i32 %3, 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "global_load_dwordx2 %0, %1, %2\n\t"
            "s_branch 2f\n"
            "1:\n\t"
            "ds_read_b64 %0, %4\n"
            "2:"
            : "=&v"(d)
            : "v"(loff), "s"(base), "s"(cj), "v"(la)
            : "memory", "scc");
    };
    auto wait_all = [&](f2 (&bv)[U]) {
        if constexpr (U == 8)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7])::"memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]),
                           "+v"(bv[8]), "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15])::"memory");
    };
    int k = wb;
    for (; k + U <= we; k += U) {
        int c[U];
        float v[U];
        f2 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) gather(c[j], bv[j]);
        wait_all(bv);
        if (k + U <= rend) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        }
    }
    if (k < we) {
        int c[U];
        float v[U];
        f2 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            c[j] = code[k + j];
            v[j] = val[k + j];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) gather((k + j < we) ? c[j] : 0, bv[j]);
        wait_all(bv);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (k + j < we) {
                while (k + j >= rend) flush();
                acc[0] = __builtin_fmaf(v[j], bv[j][0], acc[0]);
                acc[1] = __builtin_fmaf(v[j], bv[j][1], acc[1]);
            }
        }
    }
    while (cur < nrows) flush();
}


// ----------------------------------------------------------------------------- MODE 10/11: the same, lean on the scalar unit
// The all-staged floor of MODE 6/7 (2.0-2.4 ms on the products-shaped graph, no memory gather at all) is the SCALAR unit:
// ~9 scalar instructions per entry at one per clock per CU. Here: {code, value} interleaved (one s_load per chunk, a running
// pointer), ONE vector add forms the offset used by either path (`code << 9` drops the flag bit: LDS byte address of the staged
// row or byte offset of the B row, + lane * 8), the branch is cmp + cbranch, the multiply-adds take the value straight from
// its SGPR. ~4 scalar instructions per entry.
template <int H, int WAVES, int U, int DBG = 0>
__global__ __launch_bounds__(WAVES * 64) void scalar_hot2_kernel(HotArgs a) {
    __shared__ f2 s_hot[H * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);
    const int task = blk * WAVES + wave;
    cint_ptr tk = (cint_ptr)(uintptr_t)a.tasks + (size_t)task * 4;
    const int row_first = tk[0], nrows = tk[1], wb = tk[2], we = tk[3];
    cint_ptr rowptr = (cint_ptr)(uintptr_t)a.rowptr + row_first;
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm + row_first;
    cint_ptr ev = (cint_ptr)(uintptr_t)a.code + (size_t)wb * 2;  // {code, value bits} per entry
    const float* Bp = a.B;
    const uint32_t loff = (uint32_t)lane * 8u;
    {
        const int total = ((cint_ptr)(uintptr_t)a.nhot)[blk] * 32;  // f4 elements
        const int32_t* hc = a.hot_cols + (size_t)blk * H;
        const f4* B4 = reinterpret_cast<const f4*>(a.B);
        f4* s_hot4 = reinterpret_cast<f4*>(s_hot);
        for (int i0 = 0; i0 < total; i0 += WAVES * 64 * 4) {
            f4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * WAVES * 64 + tid;
                const int ic = i < total ? i : total - 1;
                r[u] = B4[(size_t)hc[ic >> 5] * 32 + (ic & 31)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * WAVES * 64 + tid;
                if (i < total) s_hot4[i] = r[u];
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (nrows == 0) return;
    int cur = 0;
    int rend = rowptr[1], rend_next = rowptr[nrows > 1 ? 2 : 1];
    int crow = perm[0], crow_next = perm[nrows > 1 ? 1 : 0];
    float acc0 = 0.0f, acc1 = 0.0f;
    auto flush = [&]() {
        float* Crow = a.C + (size_t)crow * 128u;
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(f2{acc0, acc1}), "s"(Crow) : "memory");
        acc0 = acc1 = 0.0f;
        ++cur;
        rend = rend_next;
        crow = crow_next;
        const int nx = (cur + 1 < nrows) ? cur + 1 : nrows - 1;
        rend_next = rowptr[nx + 1];
        crow_next = perm[nx];
    };
    auto gather = [&](int code, f2& d) {
        uint32_t voff = ((uint32_t)code << 9) + loff;
        // (the reference to s_hot is what keeps the staging stores alive: the reads below are invisible to the compiler)
        voff += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)s_hot;
        if constexpr (DBG & 4) {
            // no branches: the path not taken runs with EXEC = 0 (is a vector memory instruction without active lanes free?)
            asm volatile(
                "s_cmp_lt_i32 %2, 0\n\t"
                "s_cselect_b64 exec, 0, -1\n\t"
                "global_load_dwordx2 %0, %1, %3\n\t"
                "s_not_b64 exec, exec\n\t"
                "ds_read_b64 %0, %1\n\t"
                "s_mov_b64 exec, -1"
                : "=&v"(d)
                : "v"(voff), "s"(code), "s"(Bp)
                : "memory", "scc");
        } else
        asm volatile(
            "s_cmp_lt_i32 %2, 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "global_load_dwordx2 %0, %1, %3\n\t"
            "s_branch 2f\n"
            "1:\n\t"
            "ds_read_b64 %0, %1\n"
            "2:"
            : "=&v"(d)
            : "v"(voff), "s"(code), "s"(Bp)
            : "memory", "scc");
    };
    auto fma2 = [&](int vbits, const f2& b) {
        if constexpr (DBG & 1) {
            acc0 = __builtin_fmaf(__builtin_bit_cast(float, vbits), b[0], acc0);
            acc1 = __builtin_fmaf(__builtin_bit_cast(float, vbits), b[1], acc1);
        } else {
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "s"(vbits), "v"(b[0]));
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "s"(vbits), "v"(b[1]));
        }
    };
    auto wait_all = [&](f2 (&bv)[U]) {
        if constexpr (U == 8)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7])::"memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]),
                           "+v"(bv[8]), "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15])::"memory");
    };
    // Two chunks per trip, the {code, value} words of the NEXT chunk requested before this one's gathers: the CSR stream is
    // read once, so every scalar load is a miss all the way to memory — it now overlaps the gathers instead of preceding them.
    // (The array is padded; slots past `we` belong to the next task or the padding: harmless gathers, not summed.)
    auto process = [&](const int (&e)[2 * U], int k) {
        f2 bv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) gather(e[2 * j], bv[j]);
        wait_all(bv);
        if (k + U <= rend) {
#pragma unroll
            for (int j = 0; j < U; ++j) fma2(e[2 * j + 1], bv[j]);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (k + j < we) {
                    while (k + j >= rend) flush();
                    fma2(e[2 * j + 1], bv[j]);
                }
            }
        }
    };
    int eA[2 * U], eB[2 * U];
#pragma unroll
    for (int i = 0; i < 2 * U; ++i) eA[i] = ev[i];
    for (int k = wb; k < we;) {
#pragma unroll
        for (int i = 0; i < 2 * U; ++i) eB[i] = ev[2 * U + i];
        process(eA, k);
        k += U;
        if (k >= we) break;
#pragma unroll
        for (int i = 0; i < 2 * U; ++i) eA[i] = ev[4 * U + i];
        process(eB, k);
        k += U;
        ev += 4 * U;
    }
    while (cur < nrows) flush();
}

template <int H, int WAVES, int MAXR, int MODE>
int launch(const HotArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((hot_kernel<H, WAVES, MAXR, 8, MODE>), dim3((unsigned)a.nblocks), dim3(WAVES * 64), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int hotrows_spmm(int H, int waves, int mode, const int32_t* rowptr, const int32_t* code, const float* val,
                            const int32_t* perm, const void* tasks, const int32_t* hot_cols, const int32_t* nhot,
                            const float* B, float* C, int nblocks, long long b_bytes, void* stream) {
    HotArgs a = {rowptr, code, val, perm, reinterpret_cast<const int4*>(tasks), hot_cols, nhot, B, C, nblocks, (int)(unsigned)b_bytes};
    if (mode == 3 && b_bytes >= 0xFFFFF000ll) return -2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mode == 4 || mode == 5 || mode == 8 || mode == 9) {  // scalar-stream kernel: `nblocks * waves` tasks, four per workgroup
        const int ntasks = nblocks * waves;
        // H = workgroups per CU allowed (through an unused dynamic LDS allocation); 0 = no limit
        const unsigned lds = (H > 0 && H <= 8) ? (unsigned)(160 * 1024 / H) & ~255u : 0u;
        static bool once = false;
        if (!once) {
            hipFuncSetAttribute((const void*)scalar_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void*)scalar_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            once = true;
        }
        if (mode == 8) hipLaunchKernelGGL((scalar_half_kernel<8>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), lds, st, a, ntasks);
        else if (mode == 9) hipLaunchKernelGGL((scalar_half_kernel<16>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), lds, st, a, ntasks);
        else if (mode == 4) hipLaunchKernelGGL((scalar_kernel<8>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), lds, st, a, ntasks);
        else hipLaunchKernelGGL((scalar_kernel<16>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), lds, st, a, ntasks);
        return (int)hipGetLastError();
    }
#define SCASE(h, w)                                                                                              \
    if (mode == 6 && H == h && waves == w) {                                                                         \
        hipLaunchKernelGGL((scalar_hot_kernel<h, w, 8>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a);           \
        return (int)hipGetLastError();                                                                               \
    }                                                                                                                \
    if (mode == 10 && H == h && waves == w) {                                                                        \
        hipLaunchKernelGGL((scalar_hot2_kernel<h, w, 8>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a);          \
        return (int)hipGetLastError();                                                                               \
    }                                                                                                                \
    if (mode == 11 && H == h && waves == w) {                                                                        \
        hipLaunchKernelGGL((scalar_hot2_kernel<h, w, 16>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a);         \
        return (int)hipGetLastError();                                                                               \
    }                                                                                                                \
    if (mode >= 12 && mode <= 14 && H == h && waves == w) {                                                          \
        if (mode == 12) hipLaunchKernelGGL((scalar_hot2_kernel<h, w, 8, 1>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a); \
        if (mode == 13) hipLaunchKernelGGL((scalar_hot2_kernel<h, w, 8, 2>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a); \
        if (mode == 14) hipLaunchKernelGGL((scalar_hot2_kernel<h, w, 8, 4>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a); \
        return (int)hipGetLastError();                                                                               \
    }                                                                                                                \
    if (mode == 7 && H == h && waves == w) {                                                                         \
        hipLaunchKernelGGL((scalar_hot_kernel<h, w, 16>), dim3((unsigned)nblocks), dim3(w * 64), 0, st, a);          \
        return (int)hipGetLastError();                                                                               \
    }
    SCASE(64, 4)
    SCASE(64, 8)
    SCASE(128, 8)
    SCASE(128, 16)
    SCASE(96, 8)
    SCASE(64, 16)
    SCASE(32, 4)
    SCASE(32, 8)
    SCASE(48, 8)
    SCASE(96, 16)
    SCASE(160, 16)
    SCASE(144, 16)
    SCASE(256, 16)
#define CASE(h, w)                                                     \
    if (H == h && waves == w) {                                        \
        if (mode == 0) return launch<h, w, 128, 0>(a, st);             \
        if (mode == 1) return launch<h, w, 128, 1>(a, st);             \
        if (mode == 3) return launch<h, w, 128, 3>(a, st);             \
        return launch<h, w, 128, 2>(a, st);                            \
    }
    CASE(64, 4)
    CASE(64, 8)
    CASE(128, 4)
    CASE(128, 8)
    CASE(128, 16)
    CASE(192, 8)
    CASE(256, 16)
#undef CASE
    return -1;
}

// ----------------------------------------------------------------------------- MODE 30: cluster-blocked LDS walk (dense clustered graphs)
// A block of R clustered rows uses few distinct B rows when the graph is dense AND clustered (reddit-sized graph with 290 planted communities: 96 rows x
// 492 entries touch ~1500 distinct columns). The block's distinct columns, ascending, are cut into SLABS of H = 128 rows; the workgroup stages slab after
// slab in LDS and every wavefront adds, for each of its RPW rows, the row's entries that fall into the staged slab — ascending column order is CSR order
// for sorted rows, so each output element is still one chain in CSR order. Every gather is an LDS read; accumulators of the RPW rows stay in registers
// across the slabs.
struct SlabArgs {
    const int32_t* split;     // M x (NS + 1): CSR position where row r enters slab s
    const int32_t* ev;        // {slot, value} per entry
    const int32_t* perm;
    const int32_t* ucols;     // distinct columns of every block, ascending
    const int32_t* uoff;      // nblocks + 1
    const float* B;
    float* C;
    int nblocks, R, NS, M;
};

template <int H, int WAVES, int RPW, int U, bool MIXED>
__global__ __launch_bounds__(WAVES * 64) void slab_walk_kernel(SlabArgs a) {
    __shared__ f2 s_hot[H * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = xcd_contiguous(blockIdx.x, a.nblocks);
    cint_ptr uoff = (cint_ptr)(uintptr_t)a.uoff;
    const int u0 = uoff[blk], u1 = uoff[blk + 1];
    const int nslab = (u1 - u0 + H - 1) / H;
    const int row0 = blk * a.R + wave * RPW;
    cint_ptr split = (cint_ptr)(uintptr_t)a.split;
    cint_ptr ev = (cint_ptr)(uintptr_t)a.ev;
    const uint32_t loff = (uint32_t)lane * 8u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) f2*)s_hot + loff;
    float acc[RPW][2];
#pragma unroll
    for (int j = 0; j < RPW; ++j) acc[j][0] = acc[j][1] = 0.0f;
    const f4* B4 = reinterpret_cast<const f4*>(a.B);
    f4* s_hot4 = reinterpret_cast<f4*>(s_hot);
    for (int s = 0; s < nslab; ++s) {
        __syncthreads();
        {
            const int c0 = u0 + s * H;
            const int total = ((u1 - c0 < H) ? (u1 - c0) : H) * 32;
            const int32_t* hc = a.ucols + c0;
            for (int i0 = 0; i0 < total; i0 += WAVES * 64 * 4) {
                f4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * WAVES * 64 + tid;
                    const int ic = i < total ? i : total - 1;
                    r[u] = B4[(size_t)hc[ic >> 5] * 32 + (ic & 31)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * WAVES * 64 + tid;
                    if (i < total) s_hot4[i] = r[u];
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int r = row0 + j;
            if (r >= a.M) break;
            const size_t sp = (size_t)r * (size_t)(a.NS + 1) + (size_t)s;
            const int kb = split[sp], ke = split[sp + 1];
            for (int k = kb; k < ke; k += U) {
                int e[2 * U];
                f2 bv[U];
#pragma unroll
                for (int i = 0; i < 2 * U; ++i) e[i] = ev[(size_t)k * 2 + i];
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const uint32_t la = ((uint32_t)e[2 * i] << 9) + lds0;
                    if constexpr (MIXED) {  // code < 0: slot of the staged row; else column (gathered from memory)
                        asm volatile(
                            "s_cmp_lt_i32 %2, 0\n\t"
                            "s_cbranch_scc1 1f\n\t"
                            "global_load_dwordx2 %0, %1, %3\n\t"
                            "s_branch 2f\n"
                            "1:\n\t"
                            "ds_read_b64 %0, %1\n"
                            "2:"
                            : "=&v"(bv[i])
                            : "v"(la), "s"(e[2 * i]), "s"(a.B)
                            : "memory", "scc");
                    } else {
                        asm volatile("ds_read_b64 %0, %1" : "=&v"(bv[i]) : "v"(la) : "memory");
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                             : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7])::"memory");
                if (k + U <= ke) {
#pragma unroll
                    for (int i = 0; i < U; ++i) {
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][0]) : "s"(e[2 * i + 1]), "v"(bv[i][0]));
                        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][1]) : "s"(e[2 * i + 1]), "v"(bv[i][1]));
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < U; ++i) {
                        if (k + i < ke) {
                            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][0]) : "s"(e[2 * i + 1]), "v"(bv[i][0]));
                            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][1]) : "s"(e[2 * i + 1]), "v"(bv[i][1]));
                        }
                    }
                }
            }
        }
    }
    cint_ptr perm = (cint_ptr)(uintptr_t)a.perm;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = row0 + j;
        if (r < a.M && r < (blk + 1) * a.R) {
            float* Crow = a.C + (size_t)perm[r] * 128u;
            f2 out = {acc[j][0], acc[j][1]};
            asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(loff), "v"(out), "s"(Crow) : "memory");
        }
    }
}

extern "C" int slabwalk_spmm(int R, const int32_t* split, const int32_t* ev, const int32_t* perm, const int32_t* ucols,
                             const int32_t* uoff, const float* B, float* C, int nblocks, int NS, int M, int mixed, void* stream) {
    SlabArgs a = {split, ev, perm, ucols, uoff, B, C, nblocks, R, NS, M};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define SW(r, rpw)                                                                                                            \
    if (R == r) {                                                                                                             \
        if (mixed) hipLaunchKernelGGL((slab_walk_kernel<128, 16, rpw, 8, true>), dim3((unsigned)nblocks), dim3(16 * 64), 0, st, a);  \
        else hipLaunchKernelGGL((slab_walk_kernel<128, 16, rpw, 8, false>), dim3((unsigned)nblocks), dim3(16 * 64), 0, st, a);       \
        return (int)hipGetLastError();                                                                                        \
    }
    SW(96, 6) SW(128, 8) SW(64, 4) SW(32, 2) SW(160, 10) SW(192, 12) SW(256, 16)
#undef SW
    return -1;
}
