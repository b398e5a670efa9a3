// Torch-free user of the C-ABI in include/gcn_spmm.h: everything a non-Python caller needs.
// Builds a small skewed CSR on the host, plans it with the native planner, copies it to the
// device with plain hipMalloc/hipMemcpy, runs C = A·B (+bias, ReLU) and the transpose product
// through gcn_csr_transpose_device, and checks both against straightforward CPU loops.
// Build: hipcc -O2 --offload-arch=gfx950 -I include tests/c_abi/c_abi_smoke.cpp \
//        -L pygcn_amd/csrc -lgcn_spmm -Wl,-rpath,$PWD/pygcn_amd/csrc -o c_abi_smoke
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gcn_spmm.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define GCN_OK(x) do { int r_ = (x); if (r_ != 0) { \
    std::printf("gcn error %d (%s) at %s:%d\n", r_, gcn_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <typename T> static T *to_dev(const std::vector<T> &h)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

static int make_plan(const std::vector<int32_t> &rowptr, const int32_t *d_rowptr, const int32_t *d_col,
                     const float *d_val, int64_t n_rows, int64_t n_cols, gcn_csr_plan *plan)
{
    int64_t ni = 0, nc = 0, nl = 0;
    GCN_OK(gcn_plan_count_host(rowptr.data(), 0, n_rows, 0, 0, &ni, &nc, &nl));
    std::vector<int32_t> items(2 * ni + 1), chunk_row(nc + 1), long_row(nl + 1), long_chunk0(nl + 1);
    std::vector<int64_t> chunk_e0(nc + 1);
    GCN_OK(gcn_plan_fill_host(rowptr.data(), 0, n_rows, 0, 0, items.data(), ni, chunk_row.data(),
                              chunk_e0.data(), nc, long_row.data(), long_chunk0.data(), nl));
    plan->n_rows = n_rows; plan->n_cols = n_cols; plan->nnz = rowptr[n_rows];
    plan->rowptr = d_rowptr; plan->rowptr_is64 = 0; plan->long_thresh = GCN_DEFAULT_LONG_THRESH;
    plan->col = d_col; plan->val = d_val;
    plan->n_items = ni; plan->items = to_dev(items);
    plan->n_chunks = nc; plan->chunk_row = to_dev(chunk_row); plan->chunk_e0 = to_dev(chunk_e0);
    plan->n_long = nl; plan->long_row = to_dev(long_row); plan->long_chunk0 = to_dev(long_chunk0);
    return 0;
}

int main()
{
    if (gcn_abi_version() != GCN_ABI_VERSION) { std::printf("ABI mismatch\n"); return 1; }
    const int64_t n_rows = 3000, n_cols = 2000, F = 256;
    std::srand(7);
    std::vector<int32_t> rowptr(n_rows + 1, 0), col;
    std::vector<float> val;
    for (int64_t r = 0; r < n_rows; ++r) {
        int deg = std::rand() % 9;
        if (r == 17) deg = 1500;            // a long row (chunked path)
        if (r % 11 == 0) deg = 0;           // empty rows
        for (int k = 0; k < deg; ++k) {
            col.push_back(std::rand() % n_cols);
            val.push_back((float)(std::rand() % 1000) / 1000.f - 0.3f);
        }
        rowptr[r + 1] = (int32_t)col.size();
    }
    const int64_t nnz = (int64_t)col.size();
    std::vector<float> B(n_cols * F), bias(F), G(n_rows * F);
    for (auto &x : B) x = (float)(std::rand() % 2001) / 1000.f - 1.f;
    for (auto &x : bias) x = (float)(std::rand() % 2001) / 1000.f - 1.f;
    for (auto &x : G) x = (float)(std::rand() % 2001) / 1000.f - 1.f;

    int32_t *d_rowptr = to_dev(rowptr), *d_col = to_dev(col);
    float *d_val = to_dev(val), *d_B = to_dev(B), *d_bias = to_dev(bias), *d_G = to_dev(G);
    float *d_C = nullptr, *d_Ct = nullptr;
    HIP_OK(hipMalloc((void **)&d_C, n_rows * F * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_Ct, n_cols * F * sizeof(float)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    // forward: C = relu(A·B + bias)
    gcn_csr_plan plan;
    if (make_plan(rowptr, d_rowptr, d_col, d_val, n_rows, n_cols, &plan)) return 4;
    size_t ws_bytes = gcn_spmm_workspace_bytes(&plan, F);
    void *ws = nullptr;
    HIP_OK(hipMalloc(&ws, ws_bytes + 16));
    GCN_OK(gcn_spmm_csr(&plan, GCN_DTYPE_F32, d_B, F, d_C, F, F, d_bias, 1, ws, ws_bytes, stream));

    // backward product: CSR(A^T) on the device, then Ct = A^T · G
    int32_t *d_rowptr_t = nullptr, *d_col_t = nullptr;
    float *d_val_t = nullptr;
    HIP_OK(hipMalloc((void **)&d_rowptr_t, (n_cols + 1) * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_col_t, (nnz + 1) * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_val_t, (nnz + 1) * sizeof(float)));
    size_t tws_bytes = gcn_csr_transpose_workspace_bytes(n_rows, n_cols, nnz);
    void *tws = nullptr;
    HIP_OK(hipMalloc(&tws, tws_bytes));
    GCN_OK(gcn_csr_transpose_device(d_rowptr, 0, d_col, d_val, n_rows, n_cols, nnz, d_rowptr_t, d_col_t,
                                    d_val_t, tws, tws_bytes, stream));
    std::vector<int32_t> rowptr_t(n_cols + 1);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(rowptr_t.data(), d_rowptr_t, (n_cols + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    gcn_csr_plan plan_t;
    if (make_plan(rowptr_t, d_rowptr_t, d_col_t, d_val_t, n_cols, n_rows, &plan_t)) return 4;
    size_t ws_t_bytes = gcn_spmm_workspace_bytes(&plan_t, F);
    void *ws_t = nullptr;
    HIP_OK(hipMalloc(&ws_t, ws_t_bytes + 16));
    GCN_OK(gcn_spmm_csr(&plan_t, GCN_DTYPE_F32, d_G, F, d_Ct, F, F, nullptr, 0, ws_t, ws_t_bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<float> C(n_rows * F), Ct(n_cols * F);
    HIP_OK(hipMemcpy(C.data(), d_C, C.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(Ct.data(), d_Ct, Ct.size() * sizeof(float), hipMemcpyDeviceToHost));

    // CPU check (layers.py:34-36 + relu; and the scatter form of A^T·G)
    double err = 0, scale = 0, err_t = 0, scale_t = 0;
    std::vector<float> ref_t(n_cols * F, 0.f);
    for (int64_t r = 0; r < n_rows; ++r) {
        std::vector<float> acc(F, 0.f);
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            for (int64_t f = 0; f < F; ++f) {
                acc[f] += val[e] * B[(int64_t)col[e] * F + f];
                ref_t[(int64_t)col[e] * F + f] += val[e] * G[r * F + f];
            }
        }
        for (int64_t f = 0; f < F; ++f) {
            const float ref = std::fmax(acc[f] + bias[f], 0.f);
            err = std::fmax(err, std::fabs((double)C[r * F + f] - ref));
            scale = std::fmax(scale, std::fabs((double)ref));
        }
    }
    for (size_t i = 0; i < ref_t.size(); ++i) {
        err_t = std::fmax(err_t, std::fabs((double)Ct[i] - ref_t[i]));
        scale_t = std::fmax(scale_t, std::fabs((double)ref_t[i]));
    }
    // argument errors are reported, not crashed on
    const int bad = gcn_spmm_csr(&plan, 9, d_B, F, d_C, F, F, nullptr, 0, ws, ws_bytes, stream);
    std::printf("C_ABI_SMOKE nnz=%lld items=%lld chunks=%lld fwd_err=%.3e/%.3e bwd_err=%.3e/%.3e bad_dtype_rc=%d\n",
                (long long)nnz, (long long)plan.n_items, (long long)plan.n_chunks, err, scale, err_t,
                scale_t, bad);
    // ---- ABI 22: the epilogue's maximum side output, and the fused NLL + log_softmax backward sweep
    float *d_amax = nullptr;
    HIP_OK(hipMalloc((void **)&d_amax, sizeof(float)));
    HIP_OK(hipMemsetAsync(d_amax, 0, sizeof(float), stream));
    gcn_epilogue ep = {};                       // every option off ...
    ep.bias = d_bias;
    ep.log_softmax = 1;                         // ... except: store log_softmax(A·B + bias) rows
    ep.c_absmax = d_amax;                       //             and report max|stored value|
    GCN_OK(gcn_spmm_csr_ep(&plan, GCN_DTYPE_F32, d_B, F, d_C, F, F, &ep, ws, ws_bytes, stream));
    std::vector<int64_t> target(n_rows);
    for (int64_t r = 0; r < n_rows; ++r) target[r] = (r % 13 == 0) ? -100 : (r * 7) % F;   // some ignored rows
    int64_t kept = 0;
    for (int64_t r = 0; r < n_rows; ++r) kept += target[r] >= 0;
    const float coef = -1.f / (float)kept;
    int64_t *d_target = to_dev(target);
    std::vector<float> coef_h(1, coef);
    float *d_coef = to_dev(coef_h), *d_gp = nullptr, *d_colsum = nullptr;
    HIP_OK(hipMalloc((void **)&d_gp, n_rows * F * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_colsum, F * sizeof(float)));
    const size_t sweep_bytes = gcn_bwd_colsum_workspace_bytes(n_rows, F, GCN_DTYPE_F32);
    void *sweep_ws = nullptr;
    HIP_OK(hipMalloc(&sweep_ws, sweep_bytes + 16));
    GCN_OK(gcn_nll_log_softmax_backward_colsum(GCN_DTYPE_F32, d_target, d_coef, d_C, d_gp, d_colsum, n_rows, F,
                                               sweep_ws, sweep_bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> logp(n_rows * F), gp(n_rows * F), colsum(F);
    float amax = 0.f;
    HIP_OK(hipMemcpy(logp.data(), d_C, logp.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gp.data(), d_gp, gp.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(colsum.data(), d_colsum, F * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&amax, d_amax, sizeof(float), hipMemcpyDeviceToHost));
    double err_lse = 0, err_gp = 0, scale_gp = 0, err_cs = 0, scale_cs = 0;
    float want_amax = 0.f;
    std::vector<double> cs(F, 0.0);
    for (int64_t r = 0; r < n_rows; ++r) {
        double se = 0;
        for (int64_t f = 0; f < F; ++f) {
            se += std::exp((double)logp[r * F + f]);
            want_amax = std::fmax(want_amax, std::fabs(logp[r * F + f]));
        }
        err_lse = std::fmax(err_lse, std::fabs(se - 1.0));                   // rows of exp(logp) sum to 1
        for (int64_t f = 0; f < F; ++f) {
            const double ref = target[r] < 0 ? 0.0
                : (double)coef * ((f == target[r] ? 1.0 : 0.0) - std::exp((double)logp[r * F + f]));
            err_gp = std::fmax(err_gp, std::fabs((double)gp[r * F + f] - ref));
            scale_gp = std::fmax(scale_gp, std::fabs(ref));
            cs[f] += ref;
        }
    }
    for (int64_t f = 0; f < F; ++f) {
        err_cs = std::fmax(err_cs, std::fabs((double)colsum[f] - cs[f]));
        scale_cs = std::fmax(scale_cs, std::fabs(cs[f]));
    }
    std::printf("C_ABI_SMOKE abi22 lse_err=%.3e gp_err=%.3e/%.3e colsum_err=%.3e/%.3e amax=%g (want %g)\n", err_lse,
                err_gp, scale_gp, err_cs, scale_cs, amax, want_amax);
    const bool ok22 = err_lse <= 1e-5 && err_gp <= 1e-5 * scale_gp && err_cs <= 2e-5 * scale_cs + 1e-9 &&
                      amax == want_amax;
    // ---- rows as bitmask + non-zero values (wire format of the compressed halo exchange): round trip
    const int64_t pm = 37, pF = 64;
    std::vector<float> dense(pm * pF, 0.f), back(pm * pF, -1.f);
    std::vector<int64_t> offs(pm + 1, 0);
    for (int64_t i = 0; i < pm * pF; ++i) dense[i] = (i * 2654435761u % 7 < 2) ? (float)(i % 11) - 5.5f : 0.f;
    float *d_dense = to_dev(dense), *d_back = nullptr, *d_pv = nullptr;
    uint32_t *d_bits = nullptr;
    int32_t *d_cnt = nullptr;
    HIP_OK(hipMalloc((void **)&d_back, pm * pF * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_pv, pm * pF * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_bits, pm * (pF / 32) * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&d_cnt, pm * sizeof(int32_t)));
    GCN_OK(gcn_rows_pack_count(GCN_DTYPE_F32, d_dense, pF, nullptr, pm, pF, d_bits, d_cnt, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<int32_t> cnt(pm), cnt2(pm);
    HIP_OK(hipMemcpy(cnt.data(), d_cnt, pm * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < pm; ++r) offs[r + 1] = offs[r] + cnt[r];          // the caller's scan
    int64_t *d_offs = to_dev(offs);
    GCN_OK(gcn_rows_pack_values(GCN_DTYPE_F32, d_dense, pF, nullptr, pm, pF, d_offs, d_pv, stream));
    GCN_OK(gcn_bits_row_counts(d_bits, pm, pF / 32, d_cnt, stream));
    GCN_OK(gcn_rows_unpack(GCN_DTYPE_F32, d_bits, d_offs, d_pv, pm, pF, d_back, pF, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(back.data(), d_back, back.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(cnt2.data(), d_cnt, pm * sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t nz = 0;
    for (float v : dense) nz += v != 0.f;
    const bool ok_pack = back == dense && cnt2 == cnt && offs[pm] == nz &&
                         gcn_rows_pack_count(GCN_DTYPE_F32, d_dense, pF, nullptr, pm, 48, d_bits, d_cnt, stream) == GCN_E_BADARG;
    std::printf("C_ABI_SMOKE pack nonzeros=%lld of %lld round_trip=%s\n", (long long)offs[pm], (long long)(pm * pF),
                ok_pack ? "exact" : "MISMATCH");
    const bool ok = ok22 && ok_pack && err <= 1e-5 * scale && err_t <= 1e-5 * scale_t && bad == GCN_E_BADARG && plan.n_chunks > 0;
    std::printf(ok ? "C_ABI_SMOKE OK\n" : "C_ABI_SMOKE FAILED\n");
    return ok ? 0 : 5;
}
