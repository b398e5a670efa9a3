/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C CPU restatement of the sparse x dense products on the GraphConvolution hot path of
 * LinChen-65/pygcn.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product (pygcn_amd/) never does.
 *
 * What it restates.  The reference calls `torch.spmm(adj, support)` (pygcn/layers.py:34) and
 * autograd runs `adj.t() @ grad_out` for it (torch derivatives.yaml `mm`, mat2 formula; call
 * site `loss.backward()` pygcn/train.py:157).  The arithmetic is in the third-party dependency
 * PyTorch (unpinned in the reference's setup.py:12-15; 2.10.0+rocm7.0 in this image):
 * ATen's sparse-COO CPU addmm walks the nnz list in storage order and for every entry does
 *      out[row, :] += value * dense[col, :]           (one axpy per stored entry)
 * after zero-filling `out`; duplicates are therefore summed and unsorted input is legal.
 * `spmm_coo_f32` below follows exactly that published algorithm; `spmm_csr_f32` is the same
 * sum in CSR order (the layout the product uses), and the `_t` variants form A^T·G by
 * scattering, which is what the reference computes in backward.
 *
 * Parity pin: checked against golden vectors produced by importing the reference layer in
 * the build container (tests/golden/make_golden.py -> g2/g3/g4 fixtures); see
 * tests/test_oracle_golden.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* out[n_rows, F] = sum over COO entries in storage order; follows ATen's COO addmm worker. */
void oracle_spmm_coo_f32(int64_t nnz, const int64_t *row, const int64_t *col, const float *val,
                         const float *B, int64_t ldb, float *C, int64_t ldc, int64_t n_rows,
                         int64_t F)
{
    for (int64_t i = 0; i < n_rows; ++i) memset(C + i * ldc, 0, (size_t)F * sizeof(float));
    for (int64_t e = 0; e < nnz; ++e) {
        const float a = val[e];
        const float *b = B + col[e] * ldb;
        float *c = C + row[e] * ldc;
        for (int64_t f = 0; f < F; ++f) c[f] += a * b[f];
    }
}

/* CSR row-parallel form of the same sum: C[i,:] = sum_{k in row i} A[i,k] * B[k,:]
 * (pygcn/layers.py:34 with adj in CSR).  Rows are independent, so OpenMP over rows keeps the
 * per-row summation order fixed. */
void oracle_spmm_csr_f32(int64_t n_rows, const int64_t *rowptr, const int32_t *col,
                         const float *val, const float *B, int64_t ldb, float *C, int64_t ldc,
                         int64_t F)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n_rows; ++i) {
        float *c = C + i * ldc;
        memset(c, 0, (size_t)F * sizeof(float));
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float a = val[e];
            const float *b = B + (int64_t)col[e] * ldb;
            for (int64_t f = 0; f < F; ++f) c[f] += a * b[f];
        }
    }
}

/* Same product with fp64 accumulation (error attribution only; not the parity target). */
void oracle_spmm_csr_f64acc(int64_t n_rows, const int64_t *rowptr, const int32_t *col,
                            const float *val, const float *B, int64_t ldb, double *C,
                            int64_t ldc, int64_t F)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n_rows; ++i) {
        double *c = C + i * ldc;
        for (int64_t f = 0; f < F; ++f) c[f] = 0.0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const double a = val[e];
            const float *b = B + (int64_t)col[e] * ldb;
            for (int64_t f = 0; f < F; ++f) c[f] += a * (double)b[f];
        }
    }
}

/* Backward product of the spmm (torch `mm` derivative, mat2): GS[n_cols, F] = A^T · G, formed
 * by scattering each stored entry: GS[col, :] += value * G[row, :].  Serial: the scatter
 * order is the storage order, as in the reference's COO kernel on the transposed view. */
void oracle_spmm_csr_t_f32(int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                           const int32_t *col, const float *val, const float *G, int64_t ldg,
                           float *GS, int64_t ldgs, int64_t F)
{
    for (int64_t k = 0; k < n_cols; ++k) memset(GS + k * ldgs, 0, (size_t)F * sizeof(float));
    for (int64_t i = 0; i < n_rows; ++i) {
        const float *g = G + i * ldg;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const float a = val[e];
            float *o = GS + (int64_t)col[e] * ldgs;
            for (int64_t f = 0; f < F; ++f) o[f] += a * g[f];
        }
    }
}

/* Explicit CSR transpose (stable counting sort by column), so that the A^T·G product can also
 * be restated as a row-parallel CSR product on CSR(A^T) — the form the product's backward
 * uses.  rowptr_t has n_cols+1 entries; col_t/val_t have nnz entries. */
void oracle_csr_transpose(int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                          const int32_t *col, const float *val, int64_t *rowptr_t,
                          int32_t *col_t, float *val_t)
{
    const int64_t nnz = rowptr[n_rows];
    for (int64_t k = 0; k <= n_cols; ++k) rowptr_t[k] = 0;
    for (int64_t e = 0; e < nnz; ++e) rowptr_t[col[e] + 1]++;
    for (int64_t k = 0; k < n_cols; ++k) rowptr_t[k + 1] += rowptr_t[k];
    for (int64_t i = 0; i < n_rows; ++i) {
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int64_t dst = rowptr_t[col[e]]++;
            col_t[dst] = (int32_t)i;
            val_t[dst] = val[e];
        }
    }
    for (int64_t k = n_cols; k > 0; --k) rowptr_t[k] = rowptr_t[k - 1];
    rowptr_t[0] = 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
