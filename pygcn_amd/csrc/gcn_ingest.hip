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

// ---- COO -> CSR with duplicate reduction (pygcn/utils.py:360-368 on the device) ----------------
// key = row * n_cols + col of every stored entry; a stable radix sort of (key, storage position)
// puts duplicates next to each other IN STORAGE ORDER; the head of every run reduces its run
// sequentially (sum: what scipy's coo->csr and torch.spmm on an uncoalesced COO tensor do; max: the
// elementwise maximum the reference's symmetrization `adj + adj.T*(adj.T > adj) - adj*(adj.T > adj)`
// amounts to), so the result does not depend on thread scheduling.  No atomics.
__global__ __launch_bounds__(256) void coo_keys_kernel(const int64_t *__restrict__ row,
                                                       const int64_t *__restrict__ col, int64_t nnz,
                                                       int64_t n_cols, uint64_t *__restrict__ key,
                                                       uint32_t *__restrict__ pos)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        key[e] = (uint64_t)row[e] * (uint64_t)n_cols + (uint64_t)col[e];
        pos[e] = (uint32_t)e;
    }
}

__global__ __launch_bounds__(256) void coo_heads_kernel(const uint64_t *__restrict__ key, int64_t nnz,
                                                        int64_t *__restrict__ head)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= nnz; e += stride)
        head[e] = (e < nnz && (e == 0 || key[e] != key[e - 1])) ? 1 : 0;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void coo_reduce_kernel(const uint64_t *__restrict__ key,
                                                         const uint32_t *__restrict__ pos,
                                                         const int64_t *__restrict__ outpos,
                                                         const float *__restrict__ val, int64_t nnz,
                                                         int64_t n_rows, int64_t n_cols, int reduce_max,
                                                         IdxT *__restrict__ rowptr,
                                                         int32_t *__restrict__ col_out,
                                                         float *__restrict__ val_out,
                                                         int64_t *__restrict__ nnz_out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = tid; e < nnz; e += stride) {
        const uint64_t k = key[e];
        if (e > 0 && key[e - 1] == k) continue;             // not the head of its run
        float acc = val[pos[e]];
        for (int64_t j = e + 1; j < nnz && key[j] == k; ++j) {
            const float v = val[pos[j]];
            acc = reduce_max ? fmaxf(acc, v) : acc + v;
        }
        const int64_t o = outpos[e];
        col_out[o] = (int32_t)(k % (uint64_t)n_cols);
        val_out[o] = acc;
    }
    // rowptr[r] = number of distinct keys below r * n_cols
    for (int64_t r = tid; r <= n_rows; r += stride) {
        const uint64_t bound = (uint64_t)r * (uint64_t)n_cols;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (key[mid] < bound)
                lo = mid + 1;
            else
                hi = mid;
        }
        rowptr[r] = (IdxT)outpos[lo];
    }
    if (tid == 0) *nnz_out = outpos[nnz];
}

int key_bits64(int64_t n_rows, int64_t n_cols)
{
    const unsigned __int128 top = (unsigned __int128)n_rows * (unsigned __int128)n_cols;
    int b = 1;
    while (b < 64 && ((unsigned __int128)1 << b) < top) ++b;
    return b;
}

size_t coo_sort_temp_bytes(int64_t nnz, int bits)
{
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint64_t *)nullptr,
                                             (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, nnz, 0, bits, (hipStream_t)0);
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (int64_t *)nullptr, (int64_t *)nullptr,
                                           nnz + 1, (hipStream_t)0);
    return temp > scan ? temp : scan;
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

size_t gcn_coo_to_csr_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz)
{
    if (nnz <= 0) return 256;
    return 2 * align_up((size_t)nnz * 8) + 2 * align_up((size_t)nnz * 4) +
           align_up(((size_t)nnz + 1) * 8) +
           align_up(coo_sort_temp_bytes(nnz, key_bits64(n_rows, n_cols))) + 256;
}

int gcn_coo_to_csr_device(const int64_t *row, const int64_t *col, const float *val, int64_t nnz,
                          int64_t n_rows, int64_t n_cols, int reduce, void *rowptr_out,
                          int rowptr_is64, int32_t *col_out, float *val_out, int64_t *nnz_out,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_rows >= INT32_MAX || n_cols >= INT32_MAX ||
        nnz >= (int64_t)UINT32_MAX || rowptr_out == nullptr || nnz_out == nullptr ||
        (reduce != GCN_REDUCE_SUM && reduce != GCN_REDUCE_MAX))
        return ifail(GCN_E_BADARG, "gcn_coo_to_csr_device: bad sizes / NULL outputs / unknown reduce");
    if (nnz > 0 && (row == nullptr || col == nullptr || val == nullptr || col_out == nullptr ||
                    val_out == nullptr))
        return ifail(GCN_E_BADARG, "gcn_coo_to_csr_device: NULL entry arrays");
    if (!rowptr_is64 && nnz >= INT32_MAX)
        return ifail(GCN_E_BADARG, "gcn_coo_to_csr_device: int32 row pointers need nnz < 2^31");
    if (workspace == nullptr || workspace_bytes < gcn_coo_to_csr_workspace_bytes(n_rows, n_cols, nnz))
        return ifail(GCN_E_WORKSPACE, "gcn_coo_to_csr_device: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (nnz == 0) {
        e = hipMemsetAsync(rowptr_out, 0, (size_t)(n_rows + 1) * (rowptr_is64 ? 8 : 4), s);
        if (e == hipSuccess) e = hipMemsetAsync(nnz_out, 0, sizeof(int64_t), s);
        return e == hipSuccess ? 0 : ifail_hip(e, "gcn_coo_to_csr_device: memset");
    }
    char *w = (char *)workspace;
    uint64_t *key_in = (uint64_t *)w;
    w += align_up((size_t)nnz * 8);
    uint64_t *key_sorted = (uint64_t *)w;
    w += align_up((size_t)nnz * 8);
    uint32_t *pos_in = (uint32_t *)w;
    w += align_up((size_t)nnz * 4);
    uint32_t *pos_sorted = (uint32_t *)w;
    w += align_up((size_t)nnz * 4);
    int64_t *outpos = (int64_t *)w;
    w += align_up(((size_t)nnz + 1) * 8);
    const int bits = key_bits64(n_rows, n_cols);
    size_t temp = coo_sort_temp_bytes(nnz, bits);
    const unsigned blocks = (unsigned)std::min<int64_t>((std::max<int64_t>(nnz, n_rows) + 256) / 256,
                                                        256 * 32);
    hipLaunchKernelGGL(coo_keys_kernel, dim3(blocks), dim3(256), 0, s, row, col, nnz, n_cols, key_in,
                       pos_in);
    e = hipcub::DeviceRadixSort::SortPairs((void *)w, temp, key_in, key_sorted, pos_in, pos_sorted,
                                           nnz, 0, bits, s);
    if (e != hipSuccess) return ifail_hip(e, "gcn_coo_to_csr_device: radix sort");
    hipLaunchKernelGGL(coo_heads_kernel, dim3(blocks), dim3(256), 0, s, key_sorted, nnz, outpos);
    e = hipcub::DeviceScan::ExclusiveSum((void *)w, temp, outpos, outpos, nnz + 1, s);
    if (e != hipSuccess) return ifail_hip(e, "gcn_coo_to_csr_device: scan");
    if (rowptr_is64)
        hipLaunchKernelGGL(coo_reduce_kernel<int64_t>, dim3(blocks), dim3(256), 0, s, key_sorted,
                           pos_sorted, outpos, val, nnz, n_rows, n_cols, reduce == GCN_REDUCE_MAX,
                           (int64_t *)rowptr_out, col_out, val_out, nnz_out);
    else
        hipLaunchKernelGGL(coo_reduce_kernel<int32_t>, dim3(blocks), dim3(256), 0, s, key_sorted,
                           pos_sorted, outpos, val, nnz, n_rows, n_cols, reduce == GCN_REDUCE_MAX,
                           (int32_t *)rowptr_out, col_out, val_out, nnz_out);
    e = hipGetLastError();
    if (e != hipSuccess) return ifail_hip(e, "gcn_coo_to_csr_device: launch");
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
