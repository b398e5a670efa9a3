// gcn_ingest.hip — device-side graph ingest for the GraphConvolution path (gfx950).
//
// SURVEY §8 row f4.  What it replaces in the reference (all host-side scipy there):
//   normalize(mx) = D^-1 · mx            pygcn/utils.py:390-397   -> gcn_row_normalize_device
//   the transposed adjacency PyTorch re-derives on every backward call of torch.spmm
//   (autograd of pygcn/layers.py:34)       -> gcn_csr_transpose_device, run ONCE per graph
//
// CSR(A) -> CSR(A^T) is a stable LSD radix sort of the stored entries by column index
// (rocPRIM/hipCUB device primitive) carrying (source row, value) as one 64-bit payload, so that
// every row of A^T lists its entries in increasing source-row order: the backward sums are then
// deterministic.  Row pointers of A^T come from a binary search over the sorted keys — no atomics
// anywhere.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>

#include "gcn_spmm.h"

// error reporting shared with gcn_spmm.hip (one gcn_last_error() for the whole library)
int gcn_internal_fail(int code, const char *msg);
int gcn_internal_fail_hip(int hip_error, const char *where);

namespace {

int ifail(int code, const char *msg) { return gcn_internal_fail(code, msg); }
int ifail_hip(hipError_t e, const char *where) { return gcn_internal_fail_hip((int)e, where); }

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

int key_bits(int64_t n_cols)
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n_cols) ++b;
    return b;
}

// payload[e] = (source row of entry e) << 32 | bits of val[e]; one thread per stored entry,
// the row found by binary search in rowptr (upper bound - 1)
template <typename IdxT>
__global__ __launch_bounds__(256) void pack_entries_kernel(const IdxT *__restrict__ rowptr,
                                                           const float *__restrict__ val,
                                                           int64_t n_rows, int64_t nnz,
                                                           uint64_t *__restrict__ payload)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        int64_t lo = 0, hi = n_rows;   // first row r with rowptr[r+1] > e
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)rowptr[mid + 1] > e)
                hi = mid;
            else
                lo = mid + 1;
        }
        payload[e] = ((uint64_t)(uint32_t)lo << 32) | (uint64_t)__float_as_uint(val[e]);
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void unpack_entries_kernel(const uint64_t *__restrict__ payload,
                                                             const int32_t *__restrict__ keys_sorted,
                                                             int64_t n_cols, int64_t nnz,
                                                             IdxT *__restrict__ rowptr_t,
                                                             int32_t *__restrict__ col_t,
                                                             float *__restrict__ val_t)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = tid; e < nnz; e += stride) {
        const uint64_t v = payload[e];
        col_t[e] = (int32_t)(v >> 32);
        val_t[e] = __uint_as_float((uint32_t)v);
    }
    // rowptr_t[c] = number of stored entries with column < c  (lower bound in the sorted keys)
    for (int64_t c = tid; c <= n_cols; c += stride) {
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys_sorted[mid] < c)
                lo = mid + 1;
            else
                hi = mid;
        }
        rowptr_t[c] = (IdxT)lo;
    }
}

// D^-1 · M: one wavefront per row (grid-stride), lanes stride over the row's entries,
// wavefront shuffle reduction of the row sum, then the scaling pass.  Rows that sum to 0 stay 0
// (the reference turns the infinite reciprocal into 0: pygcn/utils.py:394).
template <typename IdxT>
__global__ __launch_bounds__(256) void row_normalize_kernel(const IdxT *__restrict__ rowptr,
                                                            float *__restrict__ val, int64_t n_rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rows; r += n_waves) {
        const int64_t e0 = (int64_t)rowptr[r], e1 = (int64_t)rowptr[r + 1];
        float s = 0.f;
        for (int64_t e = e0 + lane; e < e1; e += 64) s += val[e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float inv = (s != 0.f) ? 1.f / s : 0.f;
        for (int64_t e = e0 + lane; e < e1; e += 64) val[e] *= inv;
    }
}

size_t sort_temp_bytes(int64_t nnz, int bits)
{
    size_t temp = 0;
    // size query only (d_temp_storage == nullptr): nothing is launched
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const int32_t *)nullptr,
                                             (int32_t *)nullptr, (const uint64_t *)nullptr,
                                             (uint64_t *)nullptr, nnz, 0, bits, (hipStream_t)0);
    return temp;
}

}   // namespace

extern "C" {

size_t gcn_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz)
{
    (void)n_rows;
    if (nnz <= 0) return 256;
    return align_up((size_t)nnz * 4) + 2 * align_up((size_t)nnz * 8) +
           align_up(sort_temp_bytes(nnz, key_bits(n_cols))) + 256;
}

int gcn_csr_transpose_device(const void *rowptr, int rowptr_is64, const int32_t *col,
                             const float *val, int64_t n_rows, int64_t n_cols, int64_t nnz,
                             void *rowptr_t, int32_t *col_t, float *val_t, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    if (rowptr == nullptr || rowptr_t == nullptr || n_rows < 0 || n_cols < 0 || nnz < 0 ||
        n_rows >= INT32_MAX || n_cols >= INT32_MAX)
        return ifail(GCN_E_BADARG, "gcn_csr_transpose_device: bad sizes or NULL row pointers");
    if (nnz > 0 && (col == nullptr || val == nullptr || col_t == nullptr || val_t == nullptr))
        return ifail(GCN_E_BADARG, "gcn_csr_transpose_device: NULL entry arrays");
    if (workspace == nullptr || workspace_bytes < gcn_csr_transpose_workspace_bytes(n_rows, n_cols, nnz))
        return ifail(GCN_E_WORKSPACE, "gcn_csr_transpose_device: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)workspace;
    int32_t *keys_sorted = (int32_t *)w;
    w += align_up((size_t)nnz * 4);
    uint64_t *pay_in = (uint64_t *)w;
    w += align_up((size_t)nnz * 8);
    uint64_t *pay_out = (uint64_t *)w;
    w += align_up((size_t)nnz * 8);
    const int bits = key_bits(n_cols);
    size_t temp = sort_temp_bytes(nnz, bits);
    const unsigned blocks = (unsigned)std::min<int64_t>((std::max<int64_t>(nnz, n_cols + 1) + 255) / 256,
                                                        256 * 32);
    if (nnz > 0) {
        if (rowptr_is64)
            hipLaunchKernelGGL(pack_entries_kernel<int64_t>, dim3(blocks), dim3(256), 0, s,
                               (const int64_t *)rowptr, val, n_rows, nnz, pay_in);
        else
            hipLaunchKernelGGL(pack_entries_kernel<int32_t>, dim3(blocks), dim3(256), 0, s,
                               (const int32_t *)rowptr, val, n_rows, nnz, pay_in);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs((void *)w, temp, col, keys_sorted, pay_in,
                                                          pay_out, nnz, 0, bits, s);
        if (e != hipSuccess) return ifail_hip(e, "gcn_csr_transpose_device: radix sort");
    }
    if (rowptr_is64)
        hipLaunchKernelGGL(unpack_entries_kernel<int64_t>, dim3(blocks), dim3(256), 0, s, pay_out,
                           keys_sorted, n_cols, nnz, (int64_t *)rowptr_t, col_t, val_t);
    else
        hipLaunchKernelGGL(unpack_entries_kernel<int32_t>, dim3(blocks), dim3(256), 0, s, pay_out,
                           keys_sorted, n_cols, nnz, (int32_t *)rowptr_t, col_t, val_t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ifail_hip(e, "gcn_csr_transpose_device: launch");
    return 0;
}

int gcn_row_normalize_device(const void *rowptr, int rowptr_is64, float *val, int64_t n_rows,
                             void *stream)
{
    if (rowptr == nullptr || n_rows < 0)
        return ifail(GCN_E_BADARG, "gcn_row_normalize_device: bad arguments");
    if (n_rows == 0) return 0;
    if (val == nullptr) return ifail(GCN_E_BADARG, "gcn_row_normalize_device: val is NULL");
    const unsigned blocks = (unsigned)std::min<int64_t>((n_rows + 3) / 4, 256 * 64);
    hipStream_t s = (hipStream_t)stream;
    if (rowptr_is64)
        hipLaunchKernelGGL(row_normalize_kernel<int64_t>, dim3(blocks), dim3(256), 0, s,
                           (const int64_t *)rowptr, val, n_rows);
    else
        hipLaunchKernelGGL(row_normalize_kernel<int32_t>, dim3(blocks), dim3(256), 0, s,
                           (const int32_t *)rowptr, val, n_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ifail_hip(e, "gcn_row_normalize_device: launch");
    return 0;
}

}   // extern "C"
