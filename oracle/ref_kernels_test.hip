// ref_kernels_test.hip — TEST INFRASTRUCTURE. Frame around the reference's benchmark kernels:
// spmm_test_kernels.inc = /root/reference/spmm_test.cu:62-492 (warmup, spmm_test0..4<T>,
// spmmWrapper), cut at build time by oracle/make_ref.sh and compiled by hipcc for gfx950
// exactly as written (the CUDA dialect it uses — <<<>>>, __syncwarp, extern __shared__ — is
// accepted by hipcc; no header of the image is replaced). On a 64-lane wavefront a (32, tile_row)
// block puts two of the reference's 32-thread "warps" in one wavefront; __syncwarp is then a
// wavefront-scope fence, which orders the LDS traffic of both halves, so the kernels compute what
// they compute on the reference's GPUs. Used (a) as an independent statement of the reference's
// DEVICE arithmetic for the -m gpu parity tests and (b) as the "reference kernels on this MI355X"
// timing column of bench.py. Never linked into the product.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "spmm_test_kernels.inc"

extern "C" {

// spmmWrapper(method, tile_row, ...) — spmm_test.cu:456. Launches on the null stream, as the
// reference does. sync != 0: wait and return the device status.
int ref_spmm_wrapper(int method, int tile_row, int A_nrows, int B_ncols, int* A_rowPtr, int* A_colInd, float* A_val,
                     float* B, float* C, int sync) {
    if (method < 0 || method > 4 || tile_row < 1) return (int)hipErrorInvalidValue;
    spmmWrapper(method, tile_row, A_nrows, B_ncols, A_rowPtr, A_colInd, A_val, B, C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return sync ? (int)hipDeviceSynchronize() : 0;
}

// the reference's 200 empty launches before timing (spmm_test.cu:718-720)
int ref_warmup(int n) {
    for (int i = 0; i < n; ++i) warmup<<<1, 1>>>();
    return (int)hipDeviceSynchronize();
}

}  // extern "C"
