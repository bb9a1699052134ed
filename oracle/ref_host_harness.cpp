// ref_host_harness.cpp — TEST INFRASTRUCTURE. A frame around lines of the REFERENCE that
// oracle/make_ref.sh cuts out of /root/reference at build time (nothing of the reference is
// stored here; the .inc files live in a temporary directory during the build only).
//
// It gives the reference's own host code C entry points so tests can pin
// oracle/gespmm_oracle.c and the product loader against it, bit for bit:
//   util_body.inc    = util/util.hpp:57-333   compare / customSort / readTuples / makeSymmetric / readMtx
//   coo_to_csr.inc   = spmm_test.cu:558-581   COO -> CSR (A_data = 1)
//   fill_b.inc       = spmm_test.cu:592-594   B[i] = float(rand()%100-50)/100
//   golden.inc       = spmm_test.cu:596-604   the CPU golden loop (i -> k -> ptr, fp32 accumulator)
// The frame only declares the variable names those lines use (spmm_test.cu:497-516) and the
// standard headers util.hpp itself includes at :4-12 (the boost include at :14 and the
// `namespace po` alias at :21 are outside the cut — no stand-in header exists).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <tuple>
#include <typeinfo>
#include <vector>

#include "util/mmio.hpp"  // the reference's file, through -I/root/reference

#include "util_body.inc"

static std::vector<int> g_rows, g_cols;
static std::vector<float> g_vals;

extern "C" {

// readMtx<float>(fname, ...) — util.hpp:286. exit(1)s on a missing file / bad banner, as the
// reference does: call it through the ref_readmtx process for those cases.
int ref_read_mtx(const char* fname, int* nrows, int* ncols, int* nvals) {
    g_rows.clear();
    g_cols.clear();
    g_vals.clear();
    int rc = readMtx<float>(fname, g_rows, g_cols, g_vals, *nrows, *ncols, *nvals);
    return rc;
}

// sizes of the three vectors after readMtx (nvals can disagree with them on short files)
int ref_tuple_count(void) { return (int)g_rows.size(); }

void ref_copy_tuples(int* rows, int* cols, float* vals) {
    if (rows) std::memcpy(rows, g_rows.data(), g_rows.size() * sizeof(int));
    if (cols) std::memcpy(cols, g_cols.data(), g_cols.size() * sizeof(int));
    if (vals) std::memcpy(vals, g_vals.data(), g_vals.size() * sizeof(float));
}

// spmm_test.cu:558-581 with the names of :497-516
void ref_coo_to_csr(int A_nrows, int A_ncols, int nnz, const int* rows, const int* cols, int* A_indptr,
                    int* A_indices, float* A_data) {
    std::vector<int> row_indices(rows, rows + nnz);
    std::vector<int> col_indices(cols, cols + nnz);
#include "coo_to_csr.inc"
}

// spmm_test.cu:592-594; the reference seeds with time(0) (:587-588), the frame takes the seed
void ref_fill_B(unsigned seed, int max_ncols, int A_ncols, float* B) {
    srand(seed);
#include "fill_b.inc"
}

// spmm_test.cu:596-604
void ref_golden(int A_nrows, int max_ncols, const int* A_indptr, const int* A_indices, const float* A_data,
                const float* B, float* golden) {
#include "golden.inc"
}

}  // extern "C"

#ifdef REF_MAIN
// ref_readmtx file.mtx [dump.bin]  ->  "<M> <N> <nvals> <tuples>" on the last stdout line;
// dump.bin = rows[], cols[], vals[] (int32, int32, float32). Exit status is the reference's.
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    int M = 0, N = 0, nv = 0;
    ref_read_mtx(argv[1], &M, &N, &nv);
    printf("\n%d %d %d %d\n", M, N, nv, ref_tuple_count());
    if (argc > 2) {
        FILE* f = fopen(argv[2], "wb");
        if (!f) return 3;
        fwrite(g_rows.data(), sizeof(int), g_rows.size(), f);
        fwrite(g_cols.data(), sizeof(int), g_cols.size(), f);
        fwrite(g_vals.data(), sizeof(float), g_vals.size(), f);
        fclose(f);
    }
    return 0;
}
#endif
