// gcn_plan.hip — the launch schedule of a CSR adjacency built ON THE DEVICE (gfx950).
//
// SURVEY §8 row f4.  The schedule (struct gcn_csr_plan, include/gcn_spmm.h) is what the prepared
// adjacency carries instead of the torch sparse tensor the reference builds once at load time
// (pygcn/utils.py:407-414) and hands to every GraphConvolution.forward (pygcn/layers.py:32).
// gcn_plan_{count,fill}_host walk the row pointer on the CPU; at config C5 that is a 200 MB
// device->host copy and a 50-million-step sequential loop.  This file produces the SAME schedule,
// array for array, without the row pointer ever leaving HBM.
//
// The host planner is a greedy left-to-right segmentation: an item starts at a short row s and
// takes the following short rows while the summed cost (stored entries + 1 per row) stays within
// `item_cost` and the item has fewer than 64 rows; a long row (more than `long_thresh` entries)
// belongs to no item and is cut into chunks.  Greedy segmentation looks sequential, but:
//   1. where an item that STARTS at short row s would END depends on s alone:
//        end(s) = the largest e <= min(s + 64, n) with cost[s, e) <= item_cost  (at least s + 1),
//      a 6-step binary search in the prefix sums of the row costs (a long row is given the cost
//      item_cost + 1, so no item can span it);
//   2. the item after that one starts at next(s) = the first short row >= end(s);
//   3. the item starts of the greedy walk are the orbit of the first short row under `next`.
//      The orbit is marked by pointer doubling: in round k every marked row marks next^(2^k) of
//      itself, then next^(2^k) is squared — ceil(log2 n) rounds of two gathers over n integers.
// Prefix sums are hipCUB device scans; the orbit lives in "short-row rank" space so that long rows
// cost nothing.  Everything is deterministic; no atomics.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>

#include "gcn_spmm.h"

int gcn_internal_fail(int code, const char *msg);
int gcn_internal_fail_hip(int hip_error, const char *where);

namespace {

constexpr int kWaveRows = 64;          // an item holds at most one wavefront of rows

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct PlanWs {          // carved out of the caller's workspace; every array has n_rows + 1 entries
    int64_t *cost;       // row cost, then (in place) its exclusive prefix sum
    int64_t *chunks;     // chunks per row (0 for short rows), then its exclusive prefix sum
    int32_t *long_rank;  // 1 for long rows, then exclusive prefix sum = index among the long rows
    int32_t *short_rank; // 1 for short rows, then exclusive prefix sum = index among the short rows
    int32_t *short_row;  // [rank] -> row
    int32_t *item_end;   // [rank] -> end(row) of the item that would start there
    int32_t *jump[2];    // [rank] -> rank, next^(2^k), double-buffered
    int32_t *mark;       // [rank] 1 if an item of the greedy walk starts at that short row
    int32_t *item_id;    // exclusive prefix sum of mark
    void *scan_temp;
    size_t scan_temp_bytes;
};

size_t scan_temp_bytes_for(int64_t n)
{
    size_t t64 = 0, t32 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t64, (int64_t *)nullptr, (int64_t *)nullptr, n,
                                           (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t32, (int32_t *)nullptr, (int32_t *)nullptr, n,
                                           (hipStream_t)0);
    return std::max(t64, t32);
}

size_t carve(void *workspace, int64_t n_rows, PlanWs *ws)
{
    const size_t m = (size_t)n_rows + 1;
    char *w = (char *)workspace;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = w ? (void *)(w + off) : nullptr;
        off += align_up(bytes);
        return p;
    };
    PlanWs t;
    t.cost = (int64_t *)take(m * 8);
    t.chunks = (int64_t *)take(m * 8);
    t.long_rank = (int32_t *)take(m * 4);
    t.short_rank = (int32_t *)take(m * 4);
    t.short_row = (int32_t *)take(m * 4);
    t.item_end = (int32_t *)take(m * 4);
    t.jump[0] = (int32_t *)take(m * 4);
    t.jump[1] = (int32_t *)take(m * 4);
    t.mark = (int32_t *)take(m * 4);
    t.item_id = (int32_t *)take(m * 4);
    t.scan_temp_bytes = scan_temp_bytes_for((int64_t)m);
    t.scan_temp = take(t.scan_temp_bytes);
    if (ws) *ws = t;
    return off + 256;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void plan_rows_kernel(const IdxT *__restrict__ rp, int64_t n_rows,
                                                        int item_cost, int long_thresh, PlanWs ws)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
        int64_t cost = 0, chunks = 0;
        int32_t is_long = 0, is_short = 0;
        if (r < n_rows) {
            const int64_t deg = (int64_t)rp[r + 1] - (int64_t)rp[r];
            if (deg > long_thresh) {
                is_long = 1;
                chunks = (deg + long_thresh - 1) / long_thresh;
                cost = (int64_t)item_cost + 1;     // no item can contain this row
            } else {
                is_short = 1;
                cost = (deg > 0 ? deg : 0) + 1;
            }
        }
        ws.cost[r] = cost;
        ws.chunks[r] = chunks;
        ws.long_rank[r] = is_long;
        ws.short_rank[r] = is_short;
    }
}

// per short row: its rank -> (row, end of the item that would start here, rank of the next start)
template <typename IdxT>
__global__ __launch_bounds__(256) void plan_short_kernel(const IdxT *__restrict__ rp, int64_t n_rows,
                                                         int item_cost, int long_thresh, PlanWs ws)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int32_t n_short = ws.short_rank[n_rows];
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
        if (r == n_rows) {                      // sentinel rank: past the last short row
            ws.jump[0][n_short] = n_short;
            ws.mark[n_short] = 0;
            continue;
        }
        const int64_t deg = (int64_t)rp[r + 1] - (int64_t)rp[r];
        if (deg > long_thresh) continue;
        const int32_t rank = ws.short_rank[r];
        const int64_t base = ws.cost[r];
        int64_t lo = r + 1, hi = std::min<int64_t>(r + kWaveRows, n_rows);   // answer in [lo, hi]
        while (lo < hi) {                       // largest e with cost[r, e) <= item_cost
            const int64_t mid = (lo + hi + 1) >> 1;
            if (ws.cost[mid] - base <= (int64_t)item_cost)
                lo = mid;
            else
                hi = mid - 1;
        }
        ws.short_row[rank] = (int32_t)r;
        ws.item_end[rank] = (int32_t)lo;
        ws.jump[0][rank] = ws.short_rank[lo];   // rank of the first short row >= end
        ws.mark[rank] = rank == 0 ? 1 : 0;      // the walk starts at the first short row
    }
}

// one doubling round: marked ranks mark next^(2^k) of themselves (marking is monotone and only
// ever reaches ranks on the orbit, so doing it in place is safe), then next^(2^k) is squared
__global__ __launch_bounds__(256) void plan_double_kernel(const int32_t *__restrict__ jump_in,
                                                          int32_t *__restrict__ jump_out,
                                                          int32_t *mark,
                                                          const int32_t *__restrict__ n_short_p)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_short = *n_short_p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_short; i += stride) {
        const int32_t j = jump_in[i];
        if (__hip_atomic_load(&mark[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            __hip_atomic_store(&mark[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        jump_out[i] = jump_in[j];
    }
}

__global__ void plan_counts_kernel(PlanWs ws, int64_t n_rows, int64_t *__restrict__ counts)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int32_t n_short = ws.short_rank[n_rows];
        counts[0] = ws.item_id[n_short];     // marks before the sentinel
        counts[1] = ws.chunks[n_rows];
        counts[2] = ws.long_rank[n_rows];
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void plan_fill_kernel(const IdxT *__restrict__ rp, int64_t n_rows,
                                                        int long_thresh, PlanWs ws,
                                                        int32_t *__restrict__ items, int64_t n_items,
                                                        int32_t *__restrict__ chunk_row,
                                                        int64_t *__restrict__ chunk_e0, int64_t n_chunks,
                                                        int32_t *__restrict__ long_row,
                                                        int32_t *__restrict__ long_chunk0, int64_t n_long)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int32_t n_short = ws.short_rank[n_rows];
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += stride) {
        if (r == n_rows) {
            if (n_long >= 0 && ws.long_rank[n_rows] == n_long) long_chunk0[n_long] = (int32_t)n_chunks;
            continue;
        }
        if (r < n_short && ws.mark[r]) {             // r read as a RANK here
            const int64_t t = ws.item_id[r];
            if (t < n_items) {
                items[2 * t] = ws.short_row[r];
                items[2 * t + 1] = ws.item_end[r];
            }
        }
        const int64_t deg = (int64_t)rp[r + 1] - (int64_t)rp[r];
        if (deg > long_thresh) {                      // r read as a ROW here
            const int64_t l = ws.long_rank[r], o = ws.chunks[r];
            const int64_t k = (deg + long_thresh - 1) / long_thresh;
            if (l < n_long && o + k <= n_chunks) {
                long_row[l] = (int32_t)r;
                long_chunk0[l] = (int32_t)o;
                for (int64_t c = 0; c < k; ++c) {
                    chunk_row[o + c] = (int32_t)r;
                    chunk_e0[o + c] = (int64_t)rp[r] + c * long_thresh;
                }
            }
        }
    }
}

unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32); }

}   // namespace

extern "C" {

size_t gcn_plan_device_workspace_bytes(int64_t n_rows)
{
    if (n_rows < 0 || n_rows >= INT32_MAX) return 0;
    return carve(nullptr, n_rows, nullptr);
}

int gcn_plan_count_device(const void *rowptr, int rowptr_is64, int64_t n_rows, int32_t item_cost,
                          int32_t long_thresh, void *workspace, size_t workspace_bytes,
                          int64_t *counts, void *stream)
{
    if (rowptr == nullptr || counts == nullptr || n_rows < 0 || n_rows >= INT32_MAX)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_plan_count_device: bad rowptr / n_rows / counts");
    if (workspace == nullptr || workspace_bytes < gcn_plan_device_workspace_bytes(n_rows))
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_plan_count_device: workspace too small");
    if (item_cost <= 0) item_cost = GCN_DEFAULT_ITEM_COST;
    if (long_thresh <= 0) long_thresh = GCN_DEFAULT_LONG_THRESH;
    hipStream_t s = (hipStream_t)stream;
    PlanWs ws;
    carve(workspace, n_rows, &ws);
    const int64_t m = n_rows + 1;
    const dim3 grid(grid_for(m)), block(256);
    if (rowptr_is64)
        hipLaunchKernelGGL(plan_rows_kernel<int64_t>, grid, block, 0, s, (const int64_t *)rowptr,
                           n_rows, item_cost, long_thresh, ws);
    else
        hipLaunchKernelGGL(plan_rows_kernel<int32_t>, grid, block, 0, s, (const int32_t *)rowptr,
                           n_rows, item_cost, long_thresh, ws);
    size_t tb = ws.scan_temp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(ws.scan_temp, tb, ws.cost, ws.cost, m, s);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(ws.scan_temp, tb, ws.chunks, ws.chunks, m, s);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(ws.scan_temp, tb, ws.long_rank, ws.long_rank, m, s);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(ws.scan_temp, tb, ws.short_rank, ws.short_rank, m, s);
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_plan_count_device: scan");
    if (rowptr_is64)
        hipLaunchKernelGGL(plan_short_kernel<int64_t>, grid, block, 0, s, (const int64_t *)rowptr,
                           n_rows, item_cost, long_thresh, ws);
    else
        hipLaunchKernelGGL(plan_short_kernel<int32_t>, grid, block, 0, s, (const int32_t *)rowptr,
                           n_rows, item_cost, long_thresh, ws);
    int rounds = 0;
    while (((int64_t)1 << rounds) < m) ++rounds;      // 2^rounds >= n_short + 1
    for (int k = 0; k < rounds; ++k)
        hipLaunchKernelGGL(plan_double_kernel, grid, block, 0, s, ws.jump[k & 1], ws.jump[(k + 1) & 1],
                           ws.mark, ws.short_rank + n_rows);
    e = hipcub::DeviceScan::ExclusiveSum(ws.scan_temp, tb, ws.mark, ws.item_id, m, s);
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_plan_count_device: scan");
    hipLaunchKernelGGL(plan_counts_kernel, dim3(1), dim3(64), 0, s, ws, n_rows, counts);
    e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_plan_count_device: launch");
    return 0;
}

int gcn_plan_fill_device(const void *rowptr, int rowptr_is64, int64_t n_rows, int32_t long_thresh,
                         const void *workspace, size_t workspace_bytes, int32_t *items,
                         int64_t n_items, int32_t *chunk_row, int64_t *chunk_e0, int64_t n_chunks,
                         int32_t *long_row, int32_t *long_chunk0, int64_t n_long, void *stream)
{
    if (rowptr == nullptr || n_rows < 0 || n_rows >= INT32_MAX || long_chunk0 == nullptr ||
        n_items < 0 || n_chunks < 0 || n_long < 0 || (n_items > 0 && items == nullptr) ||
        (n_chunks > 0 && (chunk_row == nullptr || chunk_e0 == nullptr)) ||
        (n_long > 0 && long_row == nullptr))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_plan_fill_device: NULL output or bad sizes");
    if (workspace == nullptr || workspace_bytes < gcn_plan_device_workspace_bytes(n_rows))
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_plan_fill_device: workspace too small");
    if (long_thresh <= 0) long_thresh = GCN_DEFAULT_LONG_THRESH;
    hipStream_t s = (hipStream_t)stream;
    PlanWs ws;
    carve(const_cast<void *>(workspace), n_rows, &ws);
    const dim3 grid(grid_for(n_rows + 1)), block(256);
    if (rowptr_is64)
        hipLaunchKernelGGL(plan_fill_kernel<int64_t>, grid, block, 0, s, (const int64_t *)rowptr, n_rows,
                           long_thresh, ws, items, n_items, chunk_row, chunk_e0, n_chunks, long_row,
                           long_chunk0, n_long);
    else
        hipLaunchKernelGGL(plan_fill_kernel<int32_t>, grid, block, 0, s, (const int32_t *)rowptr, n_rows,
                           long_thresh, ws, items, n_items, chunk_row, chunk_e0, n_chunks, long_row,
                           long_chunk0, n_long);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_plan_fill_device: launch");
    return 0;
}

}   // extern "C"
