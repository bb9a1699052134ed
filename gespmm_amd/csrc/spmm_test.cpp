// spmm_test.cpp — the benchmark driver (drop-in boundary #1).
//
// Same command line, stdout lines, CSV side file and exit behaviour as the
// reference's driver (spmm_test.cu:495-826, built by compile.sh:1, looped by
// run_test.sh:5-29):
//
//     ./spmm_test <file.mtx> [device_id=0]
//
//   stdout:  "reading file ...", "read file ok. N=%d nnz=%d", "max_ncols = %d",
//            "running tests..."
//   appends to ./spmm_test_out.out, for N in {128,256,512} (<= max_ncols), the pair
//            "<vendor GFLOP/s>,<GE-SpMM GFLOP/s>," with no newline (run_test.sh adds
//            the matrix name and the newline). The vendor column is rocSPARSE's generic
//            SpMM (rocsparse_spmm, CSR, row-major B and C) where the reference times
//            cusparseScsrmm2 (spmm_test.cu:660,730-738); --no-vendor prints 0.000000.
//   exit 1 on a missing file / bad banner (util.hpp:300-313), EXIT_FAILURE on a
//   device error, 0 otherwise. Device allocation failure halves max_ncols and
//   retries, like spmm_test.cu:619-634.
//
// Everything device-side goes through the C ABI (include/gespmm.h); this file only
// owns host buffers, device buffers, events and the CSV. Options the reference
// hard-codes are flags here, with the reference's values as defaults:
//   --ncols a,b,c   feature widths to time           (default 128,256,512; spmm_test.cu:726)
//   --method m      kernel variant, -1 = library pick (default 2 = CRC+CWM2; spmm_test.cu:756)
//   --iters n       timed launches per width          (default 200; spmm_test.cu:714)
//   --seed s        srand seed for B                  (default time(0); spmm_test.cu:587)
//   --use-values    keep the file's values            (default: all ones; spmm_test.cu:574)
//   --validate      run every variant (and the library's own pick when --method -1 is given) against an
//                   in-driver CPU loop and print "kernel<m> WA: ..." on |diff| > 1e-2 (the reference's
//                   `#define VALIDATE` block, spmm_test.cu:595-605, 671-698) — at every width of --ncols,
//                   or at N = max_ncols like the reference when no widths are given. The CPU loop only
//                   CHECKS device output; it never produces results.
//   --cpu-baseline  time that CPU loop (1 thread) at the same widths and print its GFLOP/s
//   --atomic-baseline  also time the Gunrock app's edge-parallel atomicAdd scatter
//                   (gunrock-test/app/spmm/spmm_enactor.cuh:92-105) per width, printed on stdout (the CSV keeps
//                   the reference's two columns); with --validate it is checked against a CPU scatter loop
//   --out path      CSV side file                     (default spmm_test_out.out)
//   --no-vendor     skip the rocSPARSE comparison column
//   --plan          time the launches through a gespmm plan (the analysis stage: built once per width before the timed
//                   loop, its time printed; same results) — with --validate the planned product is checked as well
//   --tune          with --plan: the plan's kernel is chosen by gespmm_plan_tune (candidates timed on the driver's operands) before the
//                   timed loop; the tuning time is printed with the plan
//   --expected-launches n   with --plan: the launches the plan is told to expect (default: --iters, i.e. what the timed loop will run);
//                   an AUTO plan skips its analysis where that many launches would not pay for it
//   --describe      print what the library launches for each N (gespmm_describe_launch)
//   --cache dir     keep the parsed matrix as a binary file in `dir` and reuse it next time
//
// There is no CPU fallback: without a HIP device the driver fails with EXIT_FAILURE.

#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/gespmm.h"

__global__ void warmup() {}

namespace {

struct Buffers {
    gespmm_coo coo{};
    int32_t* indptr = nullptr;
    int32_t* indices = nullptr;
    float* data = nullptr;
    float* B = nullptr;
    float* C = nullptr;
    float* golden = nullptr;
    int32_t* indptr_dev = nullptr;
    int32_t* indices_dev = nullptr;
    float* data_dev = nullptr;
    float* B_dev = nullptr;
    float* C_dev = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
    FILE* fpo = nullptr;
    void release() {
        gespmm_mtx_free(&coo);
        free(indptr);
        free(indices);
        free(data);
        free(B);
        free(C);
        free(golden);
        if (indptr_dev) (void)hipFree(indptr_dev);
        if (indices_dev) (void)hipFree(indices_dev);
        if (data_dev) (void)hipFree(data_dev);
        if (B_dev) (void)hipFree(B_dev);
        if (C_dev) (void)hipFree(C_dev);
        if (start) (void)hipEventDestroy(start);
        if (stop) (void)hipEventDestroy(stop);
        if (fpo) fclose(fpo);
        fflush(stdout);
    }
};

Buffers g;

[[noreturn]] void die_hip(hipError_t e, int line) {
    fprintf(stderr, "HIP runtime error in line %d of file %s : %s \n", line, __FILE__, hipGetErrorString(e));
    printf("Exit.");
    g.release();
    exit(EXIT_FAILURE);
}
#define CHECK_HIP(x)                              \
    do {                                          \
        hipError_t e_ = (x);                      \
        if (e_ != hipSuccess) die_hip(e_, __LINE__); \
    } while (0)

[[noreturn]] void die_gespmm(int rc, int line) {
    fprintf(stderr, "gespmm error in line %d of file %s : %s \n", line, __FILE__, gespmm_error_string(rc));
    printf("Exit.");
    g.release();
    exit(EXIT_FAILURE);
}
#define CHECK_GE(x)                         \
    do {                                    \
        int rc_ = (x);                      \
        if (rc_ != 0) die_gespmm(rc_, __LINE__); \
    } while (0)

// The reference's validation loop (spmm_test.cu:596-604): rows, then columns, then
// the row's non-zeros in CSR order, fp32 accumulator. Checker only.
void cpu_check_loop(int M, int N, const int32_t* indptr, const int32_t* indices, const float* data, const float* B,
                    float* out) {
    for (int i = 0; i < M; i++)
        for (int k = 0; k < N; k++) {
            float acc = 0.0f;
            for (int p = indptr[i]; p < indptr[i + 1]; p++) acc += data[p] * B[(size_t)N * indices[p] + k];
            out[(size_t)N * i + k] = acc;
        }
}

// Vendor baseline: rocSPARSE generic SpMM on the same device buffers (comparison
// column only — the role cuSPARSE csrmm2 plays in the reference driver).
struct VendorSpmm {
    rocsparse_handle handle = nullptr;
    rocsparse_spmat_descr A = nullptr;
    rocsparse_dnmat_descr B = nullptr, C = nullptr;
    void* buffer = nullptr;
    size_t buffer_size = 0;
    float alpha = 1.0f, beta = 0.0f;
    bool ok = false;

    bool setup(int M, int K, int N, int nnz, int32_t* indptr, int32_t* indices, float* data, float* Bd, float* Cd) {
        teardown();
        if (rocsparse_create_handle(&handle) != rocsparse_status_success) return false;
        if (rocsparse_create_csr_descr(&A, M, K, nnz, indptr, indices, data, rocsparse_indextype_i32,
                                       rocsparse_indextype_i32, rocsparse_index_base_zero,
                                       rocsparse_datatype_f32_r) != rocsparse_status_success)
            return false;
        if (rocsparse_create_dnmat_descr(&B, K, N, N, Bd, rocsparse_datatype_f32_r, rocsparse_order_row) !=
            rocsparse_status_success)
            return false;
        if (rocsparse_create_dnmat_descr(&C, M, N, N, Cd, rocsparse_datatype_f32_r, rocsparse_order_row) !=
            rocsparse_status_success)
            return false;
        if (rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C,
                           rocsparse_datatype_f32_r, rocsparse_spmm_alg_default, rocsparse_spmm_stage_buffer_size,
                           &buffer_size, nullptr) != rocsparse_status_success)
            return false;
        if (hipMalloc(&buffer, buffer_size ? buffer_size : 4) != hipSuccess) return false;
        if (rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C,
                           rocsparse_datatype_f32_r, rocsparse_spmm_alg_default, rocsparse_spmm_stage_preprocess,
                           &buffer_size, buffer) != rocsparse_status_success)
            return false;
        ok = true;
        return true;
    }
    bool run() {
        return rocsparse_spmm(handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, C,
                              rocsparse_datatype_f32_r, rocsparse_spmm_alg_default, rocsparse_spmm_stage_compute,
                              &buffer_size, buffer) == rocsparse_status_success;
    }
    void teardown() {
        if (buffer) (void)hipFree(buffer);
        if (C) rocsparse_destroy_dnmat_descr(C);
        if (B) rocsparse_destroy_dnmat_descr(B);
        if (A) rocsparse_destroy_spmat_descr(A);
        if (handle) rocsparse_destroy_handle(handle);
        buffer = nullptr;
        A = nullptr;
        B = C = nullptr;
        handle = nullptr;
        ok = false;
    }
};

std::vector<int> parse_list(const char* s) {
    std::vector<int> v;
    while (*s) {
        char* e;
        long x = strtol(s, &e, 10);
        if (e == s) break;
        v.push_back((int)x);
        s = (*e == ',') ? e + 1 : e;
    }
    return v;
}

}  // namespace

int main(int argc, char** argv) {
    int max_ncols = 512;
    int dev_id = 0;
    int method = GESPMM_VARIANT_CRC_CWM2;
    int iters = 200;
    bool validate = false, cpu_baseline = false, use_values = false, seed_given = false, vendor = true, describe = false;
    bool atomic_baseline = false, use_plan = false, tune_plan = false;
    unsigned seed = 0;
    int expected_launches = 0;  // 0: what the timed loop runs (--iters)
    std::vector<int> ncols_list;
    const char* out_path = "spmm_test_out.out";
    const char* mtx_path = nullptr;
    const char* cache_dir = nullptr;
    int positional = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s needs a value\n", flag);
                exit(EXIT_FAILURE);
            }
            return argv[++i];
        };
        if (a == "--ncols") ncols_list = parse_list(next("--ncols"));
        else if (a == "--method") method = atoi(next("--method"));
        else if (a == "--iters") iters = atoi(next("--iters"));
        else if (a == "--seed") { seed = (unsigned)strtoul(next("--seed"), nullptr, 10); seed_given = true; }
        else if (a == "--out") out_path = next("--out");
        else if (a == "--validate") validate = true;
        else if (a == "--cpu-baseline") cpu_baseline = true;
        else if (a == "--use-values") use_values = true;
        else if (a == "--no-vendor") vendor = false;
        else if (a == "--describe") describe = true;
        else if (a == "--atomic-baseline") atomic_baseline = true;
        else if (a == "--plan") use_plan = true;
        else if (a == "--tune") use_plan = tune_plan = true;
        else if (a == "--expected-launches") expected_launches = atoi(next("--expected-launches"));
        else if (a == "--cache") cache_dir = next("--cache");
        else if (positional == 0) { mtx_path = argv[i]; positional++; }
        else if (positional == 1) { dev_id = atoi(argv[i]); positional++; }
    }
    if (!mtx_path) {
        fprintf(stderr, "usage: %s <file.mtx> [device_id] [--ncols a,b,c] [--method m] [--iters n] [--seed s] "
                        "[--use-values] [--validate] [--cpu-baseline] [--atomic-baseline] [--plan] [--tune] [--expected-launches n] [--no-vendor] [--describe] [--out path]\n", argv[0]);
        return EXIT_FAILURE;
    }
    if (iters < 1) iters = 1;

    g.fpo = fopen(out_path, "a");
    printf("reading file ...\n");
    int rc = gespmm_mtx_read_cached(mtx_path, cache_dir, &g.coo);
    if (rc == GESPMM_EIO) {
        printf("File %s not found", mtx_path);
        g.release();
        exit(1);
    }
    if (rc == GESPMM_EFORMAT) {
        printf("Could not process Matrix Market banner.\n");
        g.release();
        exit(1);
    }
    if (rc != 0) {
        printf("%s\n", gespmm_error_string(rc));
        g.release();
        exit(1);
    }
    const int M = g.coo.nrows, K = g.coo.ncols;
    const int nnz = (int)g.coo.nnz;

    if (!ncols_list.empty())
        for (int n : ncols_list)
            if (n > max_ncols) max_ncols = n;

    g.data = (float*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    g.indptr = (int32_t*)malloc(((size_t)M + 1) * sizeof(int32_t));
    g.indices = (int32_t*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
    g.B = (float*)malloc((size_t)max_ncols * (size_t)K * sizeof(float));
    if (validate || cpu_baseline) {
        g.C = (float*)malloc((size_t)M * (size_t)max_ncols * sizeof(float));
        g.golden = (float*)malloc((size_t)M * (size_t)max_ncols * sizeof(float));
        if (!g.C || !g.golden) {
            printf("Host malloc failed\n");
            g.release();
            return 1;
        }
    }
    if (!g.data || !g.indices || !g.indptr || !g.B) {
        printf("Host malloc failed\n");
        g.release();
        return 1;
    }

    rc = gespmm_coo_to_csr(M, K, nnz, g.coo.row, g.coo.col, use_values ? g.coo.val : nullptr, g.indptr, g.indices,
                           g.data);
    if (rc != 0) {
        fprintf(stderr, "out of bound row or column\n");
        g.release();
        return 1;
    }
    printf("read file ok. N=%d nnz=%d\n", M, nnz);

    if (!seed_given) seed = (unsigned)time(0);
    srand(seed);
    for (size_t i = 0; i < (size_t)max_ncols * (size_t)K; i++) g.B[i] = float(rand() % 100 - 50) / 100;

    CHECK_HIP(hipSetDevice(dev_id));
    for (;;) {
        hipError_t s1 = hipMalloc((void**)&g.indptr_dev, ((size_t)M + 1) * sizeof(int32_t));
        hipError_t s2 = hipMalloc((void**)&g.indices_dev, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
        hipError_t s3 = hipMalloc((void**)&g.data_dev, (size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
        hipError_t s4 = hipMalloc((void**)&g.B_dev, (size_t)max_ncols * (size_t)K * sizeof(float));
        hipError_t s5 = hipMalloc((void**)&g.C_dev, (size_t)M * (size_t)max_ncols * sizeof(float));
        if (s1 == hipSuccess && s2 == hipSuccess && s3 == hipSuccess && s4 == hipSuccess && s5 == hipSuccess) break;
        if (s1 == hipErrorNoDevice || s1 == hipErrorInvalidDevice) die_hip(s1, __LINE__);
        (void)hipGetLastError();
        if (g.indptr_dev) { (void)hipFree(g.indptr_dev); g.indptr_dev = nullptr; }
        if (g.indices_dev) { (void)hipFree(g.indices_dev); g.indices_dev = nullptr; }
        if (g.data_dev) { (void)hipFree(g.data_dev); g.data_dev = nullptr; }
        if (g.B_dev) { (void)hipFree(g.B_dev); g.B_dev = nullptr; }
        if (g.C_dev) { (void)hipFree(g.C_dev); g.C_dev = nullptr; }
        max_ncols /= 2;
        if (max_ncols < 1) die_hip(hipErrorOutOfMemory, __LINE__);
    }
    printf("max_ncols = %d\n", max_ncols);

    CHECK_HIP(hipMemcpy(g.indptr_dev, g.indptr, ((size_t)M + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (nnz > 0) {
        CHECK_HIP(hipMemcpy(g.indices_dev, g.indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(g.data_dev, g.data, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice));
    }
    // B holds max_ncols columns at upload time; every timed width N re-reads it with
    // leading dimension N (spmm_test.cu:732,757 do exactly that with the same buffer).
    CHECK_HIP(hipMemcpy(g.B_dev, g.B, (size_t)max_ncols * (size_t)K * sizeof(float), hipMemcpyHostToDevice));

    hipLaunchKernelGGL(warmup, dim3(1), dim3(1), 0, 0);
    CHECK_HIP(hipDeviceSynchronize());

    // Validation / CPU timing happen AFTER the allocation loop: max_ncols is final here, so the CPU result and
    // the device result always share one leading dimension.
    if (validate || cpu_baseline) {
        std::vector<int> widths;
        for (int n : ncols_list)
            if (n >= 1 && n <= max_ncols) widths.push_back(n);
        if (widths.empty()) widths.push_back(max_ncols);  // the reference validates at max_ncols only
        for (int N : widths) {
            const auto t0 = std::chrono::steady_clock::now();
            cpu_check_loop(M, N, g.indptr, g.indices, g.data, g.B, g.golden);
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (cpu_baseline)
                printf("cpu golden loop: %f GFLOP/s (1 thread, N=%d)\n", 2.0 * nnz * N / secs / 1e9, N);
            if (!validate) continue;
            auto compare = [&](const char* who) {
                for (int i = 0; i < M; i++)
                    for (int j = 0; j < N; j++)
                        if (!(fabs(g.C[(size_t)i * N + j] - g.golden[(size_t)i * N + j]) <= 1e-2)) {
                            printf("%s WA: C[%d, %d] = %f, golden = %f\n", who, i, j, g.C[(size_t)i * N + j],
                                   g.golden[(size_t)i * N + j]);
                            return false;  // first mismatch per variant is enough
                        }
                return true;
            };
            int checked = 0;
            std::vector<int> methods;
            for (int m = 0; m < GESPMM_NUM_VARIANTS; m++) methods.push_back(m);
            if (method == GESPMM_VARIANT_AUTO) methods.push_back(GESPMM_VARIANT_AUTO);
            for (int m : methods) {
                CHECK_HIP(hipMemset(g.C_dev, 0xff, (size_t)M * N * sizeof(float)));  // NaNs: every element must be written
                CHECK_GE(gespmm_csr_spmm_f32(g.indptr_dev, g.indices_dev, g.data_dev, g.B_dev, g.C_dev, M, K, N, nnz, m,
                                             nullptr));
                CHECK_HIP(hipMemcpy(g.C, g.C_dev, (size_t)M * N * sizeof(float), hipMemcpyDeviceToHost));
                char who[32];
                snprintf(who, sizeof who, "kernel%d", m);
                compare(who);
                checked++;
            }
            if (use_plan) {  // the same product through the analysis stage (forced row clustering): must pass the same check
                gespmm_plan* plan = nullptr;
                gespmm_plan_options po = {GESPMM_PLAN_REORDER, 0, 0, 0, 0, 0};
                CHECK_GE(gespmm_plan_create(&plan, g.indptr_dev, g.indices_dev, g.data_dev, M, K, nnz, N,
                                            method == GESPMM_VARIANT_NAIVE || method == GESPMM_VARIANT_PARREDUCE ? GESPMM_VARIANT_AUTO : method,
                                            &po, nullptr));
                CHECK_HIP(hipMemset(g.C_dev, 0xff, (size_t)M * N * sizeof(float)));
                CHECK_GE(gespmm_plan_spmm_f32(plan, g.B_dev, g.C_dev, N, nullptr));
                CHECK_HIP(hipMemcpy(g.C, g.C_dev, (size_t)M * N * sizeof(float), hipMemcpyDeviceToHost));
                compare("plan");
                gespmm_plan_destroy(plan);
                checked++;
            }
            if (vendor) {  // reference: csrmm2 checked against golden too (spmm_test.cu:671-679)
                VendorSpmm vs;
                if (vs.setup(M, K, N, nnz, g.indptr_dev, g.indices_dev, g.data_dev, g.B_dev, g.C_dev) && vs.run()) {
                    CHECK_HIP(hipMemcpy(g.C, g.C_dev, (size_t)M * N * sizeof(float), hipMemcpyDeviceToHost));
                    compare("rocsparse");
                } else {
                    printf("rocsparse spmm unavailable\n");
                }
                vs.teardown();
            }
            if (atomic_baseline && M == K) {
                // scatter form out[dest] += in[src] on the pattern (gunrock CPU_Reference, spmm_test.cuh): CPU loop as checker
                float* out_dev = nullptr;
                CHECK_HIP(hipMalloc((void**)&out_dev, (size_t)K * N * sizeof(float)));
                CHECK_GE(gespmm_baseline_atomic_scatter_f32(g.indptr_dev, g.indices_dev, g.B_dev, out_dev, M, K, N, nnz,
                                                            nullptr));
                CHECK_HIP(hipMemcpy(g.C, out_dev, (size_t)K * N * sizeof(float), hipMemcpyDeviceToHost));
                CHECK_HIP(hipFree(out_dev));
                for (size_t i = 0; i < (size_t)K * N; i++) g.golden[i] = 0.0f;
                for (int i = 0; i < M; i++)
                    for (int p2 = g.indptr[i]; p2 < g.indptr[i + 1]; p2++)
                        for (int j = 0; j < N; j++) g.golden[(size_t)g.indices[p2] * N + j] += g.B[(size_t)i * N + j];
                compare("atomic-baseline");
            }
            printf("validate done (%d variants, N=%d)\n", checked, N);
        }
    }

    CHECK_HIP(hipEventCreate(&g.start));
    CHECK_HIP(hipEventCreate(&g.stop));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(warmup, dim3(1), dim3(1), 0, 0);
    if (use_plan) {
        // the library's own warm-up, next to the reference's 200 empty launches (spmm_test.cu:720-721) and like them outside every timed
        // region: analysis kernels loaded, the analysis arena of a matrix this size allocated (gespmm.h: gespmm_init)
        const auto t0 = std::chrono::steady_clock::now();
        CHECK_GE(gespmm_init(M > K ? M : K, nnz, nullptr));
        printf("gespmm_init: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("running tests...\n");

    if (ncols_list.empty())
        for (int n = 128; n <= max_ncols; n *= 2) ncols_list.push_back(n);
    for (int N : ncols_list) {
        if (N > max_ncols || N < 1) continue;
        if (describe) {
            char what[256];
            if (gespmm_describe_launch(M, K, N, nnz, method, nullptr, what, sizeof what) > 0) printf("N=%d launches: %s\n", N, what);
        }
        const double gflop = (double)nnz * 2 / 1000000 * N;
        float rt = 0.0f;
        // vendor column (reference: cusparseScsrmm2, spmm_test.cu:730-738)
        double vendor_gflops = 0.0;
        if (vendor) {
            VendorSpmm vs;
            if (vs.setup(M, K, N, nnz, g.indptr_dev, g.indices_dev, g.data_dev, g.B_dev, g.C_dev) && vs.run()) {
                CHECK_HIP(hipEventRecord(g.start, 0));
                for (int i = 0; i < iters; i++) vs.run();
                CHECK_HIP(hipEventRecord(g.stop, 0));
                CHECK_HIP(hipEventSynchronize(g.stop));
                CHECK_HIP(hipEventElapsedTime(&rt, g.start, g.stop));
                vendor_gflops = gflop / (rt / iters);
            }
            vs.teardown();
        }
        if (g.fpo) fprintf(g.fpo, "%f,", vendor_gflops);

        gespmm_plan* plan = nullptr;
        double plan_secs = 0.0;
        if (use_plan) {
            const auto t0 = std::chrono::steady_clock::now();
            gespmm_plan_options po;
            memset(&po, 0, sizeof po);
            po.expected_launches = expected_launches > 0 ? expected_launches : iters;
            CHECK_GE(gespmm_plan_create_v2(&plan, g.indptr_dev, g.indices_dev, g.data_dev, M, K, nnz, N,
                                           method == GESPMM_VARIANT_NAIVE || method == GESPMM_VARIANT_PARREDUCE ? GESPMM_VARIANT_AUTO : method,
                                           &po, (int64_t)sizeof po, nullptr));
            if (tune_plan) CHECK_GE(gespmm_plan_tune(plan, g.B_dev, g.C_dev, N, 3, nullptr));
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            plan_secs = secs;
            char what[768];
            if (gespmm_plan_describe(plan, what, sizeof what) > 0) printf("N=%d plan (%.3f s%s): %s\n", N, secs, tune_plan ? ", tuned" : "", what);
        }
        auto launch = [&]() {
            return plan ? gespmm_plan_spmm_f32(plan, g.B_dev, g.C_dev, N, nullptr)
                        : gespmm_csr_spmm_f32(g.indptr_dev, g.indices_dev, g.data_dev, g.B_dev, g.C_dev, M, K, N, nnz, method, nullptr);
        };
        CHECK_GE(launch());
        CHECK_HIP(hipEventRecord(g.start, 0));
        for (int i = 0; i < iters; i++) CHECK_GE(launch());
        CHECK_HIP(hipEventRecord(g.stop, 0));
        CHECK_HIP(hipEventSynchronize(g.stop));
        CHECK_HIP(hipEventElapsedTime(&rt, g.start, g.stop));
        if (g.fpo) fprintf(g.fpo, "%f,", gflop / (rt / iters));
        printf("N=%d method=%d%s: %f ms/iter, %f GFLOP/s (rocsparse %f GFLOP/s)\n", N, method, plan ? " plan" : "", rt / iters,
               gflop / (rt / iters), vendor_gflops);
        // the same launches with the analysis INSIDE the time: what one process per matrix (run_test.sh:5-11) pays for its plan
        if (plan)
            printf("N=%d plan incl. analysis: %f ms for %d launches, %f GFLOP/s\n", N, plan_secs * 1e3 + rt, iters,
                   gflop * iters / (plan_secs * 1e3 + rt));
        if (plan) gespmm_plan_destroy(plan);
        if (atomic_baseline) {
            // out = A^T * B[0:M] by one atomicAdd per edge and feature; reuses C_dev when it is large enough (M == K)
            float* out_dev = g.C_dev;
            if (K != M) CHECK_HIP(hipMalloc((void**)&out_dev, (size_t)K * N * sizeof(float)));
            const int ait = iters < 20 ? iters : 20;
            CHECK_GE(gespmm_baseline_atomic_scatter_f32(g.indptr_dev, g.indices_dev, g.B_dev, out_dev, M, K, N, nnz, nullptr));
            CHECK_HIP(hipEventRecord(g.start, 0));
            for (int i = 0; i < ait; i++)
                CHECK_GE(gespmm_baseline_atomic_scatter_f32(g.indptr_dev, g.indices_dev, g.B_dev, out_dev, M, K, N, nnz,
                                                            nullptr));
            CHECK_HIP(hipEventRecord(g.stop, 0));
            CHECK_HIP(hipEventSynchronize(g.stop));
            CHECK_HIP(hipEventElapsedTime(&rt, g.start, g.stop));
            printf("N=%d atomic-baseline: %f ms/iter, %f GFLOP/s\n", N, rt / ait, gflop / (rt / ait));
            if (K != M) CHECK_HIP(hipFree(out_dev));
        }
    }

    g.release();
    return 0;
}
