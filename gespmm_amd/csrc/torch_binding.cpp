// torch_binding.cpp — the PyTorch-ROCm extension of the op (pybind11 module `_gespmm_torch`).
//
// Counterpart of the reference's pybind modules `spmm` (pytorch-custom/spmm.cpp:96-101:
// csr_spmm, csr_spmm_no_edge_value, csr2csc) and `sddmm` (sddmm.cpp:62-67: coo_sddmm,
// csr_sddmm): same functions, same argument order. It owns no kernels — every function
// validates its tensors (the reference only `assert`s, spmm.cpp:30-41), allocates the
// result with torch::empty on the input's device (spmm_kernel.cu:183,434) and calls the C
// ABI of include/gespmm.h on the CURRENT torch HIP stream (the reference launches on the
// legacy default stream). Built with plain g++ against the torch headers (no device code
// in this file); gespmm_amd/spmm.py uses it when present and falls back to the ctypes
// binding of the same C ABI otherwise.

#include <torch/extension.h>

// ROCm builds of PyTorch keep the device type "cuda": the guard / stream types are the
// "masquerading" ones (what torch's own sources use after hipification).
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "../../include/gespmm.h"

namespace {

void need(const torch::Tensor& t, const char* name, c10::ScalarType dtype, int64_t dim) {
    TORCH_CHECK(t.is_cuda(), name, " must be a HIP (cuda) device tensor; gespmm_amd has no CPU path");
    TORCH_CHECK_TYPE(t.scalar_type() == dtype, name, " must have dtype ", dtype, ", got ", t.scalar_type());
    TORCH_CHECK_VALUE(t.dim() == dim, name, " must be ", dim, "-dimensional");
    TORCH_CHECK_VALUE(t.is_contiguous(), name, " must be contiguous");
}

void same_device(const torch::Tensor& a, const torch::Tensor& b) {
    TORCH_CHECK(a.device() == b.device(), "all tensors must live on the same device");
}

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed: ", gespmm_error_string(rc), " (code ", rc, ")");
}

void* current_stream(const torch::Tensor& t) {
    return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream());
}

torch::Tensor spmm_impl(const torch::Tensor& rowptr, const torch::Tensor& colind,
                        const c10::optional<torch::Tensor>& values, const torch::Tensor& dense, int64_t variant,
                        const c10::optional<torch::Tensor>& workspace = c10::nullopt, int64_t flags = 0) {
    need(rowptr, "rowptr", torch::kInt32, 1);
    need(colind, "colind", torch::kInt32, 1);
    need(dense, "dense", torch::kFloat32, 2);
    same_device(dense, rowptr);
    same_device(dense, colind);
    const float* val = nullptr;
    if (values.has_value()) {
        need(*values, "values", torch::kFloat32, 1);
        same_device(dense, *values);
        TORCH_CHECK_VALUE(values->numel() == colind.numel(), "values and colind must have the same length");
        val = values->data_ptr<float>();
    }
    TORCH_CHECK_VALUE(rowptr.numel() >= 1, "rowptr must have M+1 >= 1 entries");
    const int64_t M = rowptr.numel() - 1, K = dense.size(0), N = dense.size(1), nnz = colind.numel();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dense.device());
    auto out = torch::empty({M, N}, dense.options());
    // Scratch for the two paths that need it (dense-graph cache blocking, long-row pass) comes from
    // torch's caching allocator: no driver allocation per call, and legal under torch.cuda.graph.
    // A caller-kept workspace (gespmm_amd/spmm.py: SpmmPlan) is used as is.
    gespmm_launch_cfg cfg = {0, 0, 0, 0, 0, (int32_t)flags};
    const int64_t ws_bytes = gespmm_csr_spmm_workspace_bytes(M, K, N, nnz, (int)variant, &cfg);
    TORCH_CHECK(ws_bytes >= 0, "gespmm_csr_spmm_workspace_bytes failed: ", gespmm_error_string((int)ws_bytes));
    torch::Tensor ws;
    void* ws_ptr = nullptr;
    if (workspace.has_value() && workspace->defined() && workspace->numel() > 0) {
        TORCH_CHECK(workspace->is_cuda() && workspace->device() == dense.device() && workspace->is_contiguous() &&
                        workspace->scalar_type() == torch::kUInt8,
                    "workspace must be a contiguous uint8 tensor on the device of `dense`");
        TORCH_CHECK_VALUE(workspace->numel() >= ws_bytes, "workspace holds ", workspace->numel(), " bytes, need ", ws_bytes);
        ws = *workspace;
        ws_ptr = ws.data_ptr();
    } else if (ws_bytes > 0) {
        ws = torch::empty({ws_bytes}, dense.options().dtype(torch::kUInt8));
        ws_ptr = ws.data_ptr();
        cfg.flags &= ~GESPMM_FLAG_REUSE_SPLIT;  // a fresh block holds nothing to reuse
    }
    check_rc(gespmm_csr_spmm_f32_ws(rowptr.data_ptr<int32_t>(), colind.data_ptr<int32_t>(), val,
                                    dense.data_ptr<float>(), out.data_ptr<float>(), M, K, N, nnz, (int)variant, &cfg,
                                    ws_ptr, ws_ptr ? ws.numel() : 0, current_stream(dense)),
             "gespmm_csr_spmm_f32");
    return out;
}

torch::Tensor csr_spmm(const torch::Tensor& rowptr, const torch::Tensor& colind, const torch::Tensor& values,
                       const torch::Tensor& dense, int64_t variant, const c10::optional<torch::Tensor>& workspace,
                       int64_t flags) {
    return spmm_impl(rowptr, colind, values, dense, variant, workspace, flags);
}

torch::Tensor csr_spmm_no_edge_value(const torch::Tensor& rowptr, const torch::Tensor& colind,
                                     const torch::Tensor& dense, int64_t variant,
                                     const c10::optional<torch::Tensor>& workspace, int64_t flags) {
    return spmm_impl(rowptr, colind, c10::nullopt, dense, variant, workspace, flags);
}

torch::Tensor csr_spmm_max(const torch::Tensor& rowptr, const torch::Tensor& colind, const torch::Tensor& dense,
                           double empty_value, int64_t variant) {
    need(rowptr, "rowptr", torch::kInt32, 1);
    need(colind, "colind", torch::kInt32, 1);
    need(dense, "dense", torch::kFloat32, 2);
    same_device(dense, rowptr);
    same_device(dense, colind);
    const int64_t M = rowptr.numel() - 1, K = dense.size(0), N = dense.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dense.device());
    auto out = torch::empty({M, N}, dense.options());
    check_rc(gespmm_csr_spmm_max_f32(rowptr.data_ptr<int32_t>(), colind.data_ptr<int32_t>(), dense.data_ptr<float>(),
                                     out.data_ptr<float>(), M, K, N, colind.numel(), (float)empty_value, (int)variant,
                                     current_stream(dense)),
             "gespmm_csr_spmm_max_f32");
    return out;
}

torch::Tensor csr2csc(const torch::Tensor& rowptr, const torch::Tensor& colind, torch::Tensor colptr,
                      torch::Tensor rowind, const torch::Tensor& csr_data) {
    need(rowptr, "rowptr", torch::kInt32, 1);
    need(colind, "colind", torch::kInt32, 1);
    need(colptr, "colptr", torch::kInt32, 1);
    need(rowind, "rowind", torch::kInt32, 1);
    need(csr_data, "csr_data", torch::kFloat32, 1);
    same_device(rowptr, colind);
    same_device(rowptr, colptr);
    same_device(rowptr, rowind);
    same_device(rowptr, csr_data);
    const int64_t M = rowptr.numel() - 1, K = colptr.numel() - 1, nnz = colind.numel();
    TORCH_CHECK_VALUE(rowind.numel() == nnz && csr_data.numel() == nnz, "rowind and csr_data must have nnz entries");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(rowptr.device());
    auto out = torch::empty({nnz}, csr_data.options());
    const int64_t ws_bytes = gespmm_csr2csc_workspace_bytes(M, K, nnz);
    TORCH_CHECK(ws_bytes >= 0, "gespmm_csr2csc_workspace_bytes failed");
    auto ws = torch::empty({ws_bytes > 0 ? ws_bytes : 1}, csr_data.options().dtype(torch::kUInt8));
    check_rc(gespmm_csr2csc_f32(rowptr.data_ptr<int32_t>(), colind.data_ptr<int32_t>(), csr_data.data_ptr<float>(),
                                colptr.data_ptr<int32_t>(), rowind.data_ptr<int32_t>(), out.data_ptr<float>(), M, K,
                                nnz, ws.data_ptr(), current_stream(rowptr)),
             "gespmm_csr2csc_f32");
    return out;
}

torch::Tensor sddmm_impl(const torch::Tensor& idx0, const char* name0, bool csr, const torch::Tensor& colind,
                         const torch::Tensor& D1, const torch::Tensor& D2) {
    need(idx0, name0, torch::kInt32, 1);
    need(colind, "colind", torch::kInt32, 1);
    need(D1, "D1", torch::kFloat32, 2);
    need(D2, "D2", torch::kFloat32, 2);
    TORCH_CHECK_VALUE(D1.size(1) == D2.size(1), "D1 and D2 must have the same number of columns");
    same_device(D1, D2);
    same_device(D1, idx0);
    same_device(D1, colind);
    const int64_t nnz = colind.numel(), N = D1.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(D1.device());
    auto out = torch::empty({nnz}, D1.options());
    if (csr) {
        const int64_t M = D1.size(0);
        TORCH_CHECK_VALUE(idx0.numel() == M + 1, "rowptr must have D1.size(0)+1 entries");
        check_rc(gespmm_sddmm_csr_f32(idx0.data_ptr<int32_t>(), colind.data_ptr<int32_t>(), D1.data_ptr<float>(),
                                      D2.data_ptr<float>(), out.data_ptr<float>(), M, nnz, N, current_stream(D1)),
                 "gespmm_sddmm_csr_f32");
    } else {
        TORCH_CHECK_VALUE(idx0.numel() == nnz, "rowind and colind must have the same length");
        check_rc(gespmm_sddmm_coo_f32(idx0.data_ptr<int32_t>(), colind.data_ptr<int32_t>(), D1.data_ptr<float>(),
                                      D2.data_ptr<float>(), out.data_ptr<float>(), nnz, N, current_stream(D1)),
                 "gespmm_sddmm_coo_f32");
    }
    return out;
}

torch::Tensor coo_sddmm(const torch::Tensor& rowind, const torch::Tensor& colind, const torch::Tensor& D1,
                        const torch::Tensor& D2) {
    return sddmm_impl(rowind, "rowind", false, colind, D1, D2);
}

torch::Tensor csr_sddmm(const torch::Tensor& rowptr, const torch::Tensor& colind, const torch::Tensor& D1,
                        const torch::Tensor& D2) {
    return sddmm_impl(rowptr, "rowptr", true, colind, D1, D2);
}

// ---- plans (the analysis stage, gespmm_plan_*): the hot call of a training loop, so it gets the short path as well.
// The Python class (spmm.SpmmPlan) owns the handle and the consistency checks; these two functions only launch.
torch::Tensor plan_spmm(int64_t handle, const torch::Tensor& dense, const c10::optional<torch::Tensor>& out_opt, int64_t M) {
    need(dense, "dense", torch::kFloat32, 2);
    const int64_t N = dense.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dense.device());
    torch::Tensor out;
    if (out_opt.has_value()) {
        out = *out_opt;
        need(out, "out", torch::kFloat32, 2);
        TORCH_CHECK_VALUE(out.size(0) == M && out.size(1) == N && out.device() == dense.device(), "out must be f32[M, N] on the same device");
    } else {
        out = torch::empty({M, N}, dense.options());
    }
    check_rc(gespmm_plan_spmm_f32(reinterpret_cast<gespmm_plan*>(handle), dense.data_ptr<float>(), out.data_ptr<float>(), N,
                                  current_stream(dense)),
             "gespmm_plan_spmm_f32");
    return out;
}

torch::Tensor plan_sddmm(int64_t handle, const torch::Tensor& D1, const torch::Tensor& D2, int64_t nnz) {
    need(D1, "D1", torch::kFloat32, 2);
    need(D2, "D2", torch::kFloat32, 2);
    TORCH_CHECK_VALUE(D1.size(1) == D2.size(1), "D1 and D2 must have the same number of columns");
    same_device(D1, D2);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(D1.device());
    auto out = torch::empty({nnz}, D1.options());
    check_rc(gespmm_plan_sddmm_f32(reinterpret_cast<gespmm_plan*>(handle), D1.data_ptr<float>(), D2.data_ptr<float>(),
                                   out.data_ptr<float>(), D1.size(1), current_stream(D1)),
             "gespmm_plan_sddmm_f32");
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "GE-SpMM for MI355X: spmm in CSR format (csr_spmm with edge values, csr_spmm_no_edge_value "
              "without), csr2csc format transformation, SDDMM in COO and CSR format";
    namespace py = pybind11;
    m.def("csr_spmm", &csr_spmm, "CSR SPMM", py::arg("rowptr"), py::arg("colind"), py::arg("values"),
          py::arg("dense"), py::arg("variant") = -1, py::arg("workspace") = py::none(), py::arg("flags") = 0);
    m.def("csr_spmm_no_edge_value", &csr_spmm_no_edge_value, "CSR SPMM NO EDGE VALUE", py::arg("rowptr"),
          py::arg("colind"), py::arg("dense"), py::arg("variant") = -1, py::arg("workspace") = py::none(),
          py::arg("flags") = 0);
    m.def("csr_spmm_max", &csr_spmm_max, "CSR SPMM, max reducer", py::arg("rowptr"), py::arg("colind"),
          py::arg("dense"), py::arg("empty_value") = -10000.0, py::arg("variant") = -1);
    m.def("csr2csc", &csr2csc, "csr2csc");
    m.def("coo_sddmm", &coo_sddmm, "COO SDDMM");
    m.def("csr_sddmm", &csr_sddmm, "CSR SDDMM");
    m.def("plan_spmm", &plan_spmm, "SpMM through a gespmm_plan handle", py::arg("handle"), py::arg("dense"),
          py::arg("out") = py::none(), py::arg("M") = 0);
    m.def("plan_sddmm", &plan_sddmm, "SDDMM through a gespmm_plan handle");
}
