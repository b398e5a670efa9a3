// gcn_pack.hip — rows of a mostly-zero activation as bitmask + non-zero values (gfx950).
//
// Used by the multi-GPU path's COMPRESSED hidden-layer exchange (pygcn_amd/sharded.py,
// ShardedGraph(compress_hidden=True)): the input of the second GraphConvolution layer is the
// output of relu + dropout (pygcn/models.py:48,50 upstream), >= 75 % zeros in training — its halo
// rows travel as 1 bit per element + the non-zero values and are expanded on arrival.  The
// reference has no multi-device code; this is part of the scaling design of SURVEY §8(e).
//
// Layout: bit j of bits[r][w] is set iff element 32·w + j of row r is non-zero (F a multiple of
// 32); vals holds the non-zero elements of the rows in row-major order, row r starting at
// offsets[r] (exclusive prefix sums of counts[]).  One wave per row; a lane owns 4 consecutive
// elements, so a 256-element slab is one wave instruction (16 B per lane at fp32).
//   pack, pass 1  gcn_rows_pack_count : bits + counts        (reads the rows once)
//   (exclusive scan of counts by the caller: offsets, total)
//   pack, pass 2  gcn_rows_pack_values: vals                  (reads the rows again; positions
//                                                              from ballots + mbcnt, no atomics)
//   unpack        gcn_rows_unpack     : dense rows from bits + offsets + vals
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gcn_spmm.h"

int gcn_internal_fail(int code, const char *msg);
int gcn_internal_fail_hip(int hip_error, const char *where);

namespace {

constexpr int kWave = 64;
constexpr int kRowsPerBlock = 4;          // one wave per row, 4 waves per workgroup
typedef uint16_t bf16_t;

template <typename T> struct Quad;       // 4 consecutive elements of a row <-> 4 floats
template <> struct Quad<float> {
    typedef float4 Raw;
    static __device__ __forceinline__ void load(const float *p, float (&x)[4])
    {
        const float4 v = *(const float4 *)p;
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&x)[4])
    {
        *(float4 *)p = make_float4(x[0], x[1], x[2], x[3]);
    }
};
template <> struct Quad<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t *p, float (&x)[4])
    {
        const uint2 v = *(const uint2 *)p;
        x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
        x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t *p, const float (&x)[4])
    {
        // (the values ARE bf16 numbers: truncation is exact)
        uint2 v;
        v.x = (__float_as_uint(x[0]) >> 16) | (__float_as_uint(x[1]) & 0xffff0000u);
        v.y = (__float_as_uint(x[2]) >> 16) | (__float_as_uint(x[3]) & 0xffff0000u);
        *(uint2 *)p = v;
    }
};

__device__ __forceinline__ int below(unsigned long long m)      // set bits of m in lanes below this one
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// OR over each group of 8 consecutive lanes (quad xor-1, xor-2, then the half-row mirror)
__device__ __forceinline__ uint32_t or8(uint32_t v)
{
    v |= dpp_u<0xB1>(v);
    v |= dpp_u<0x4E>(v);
    v |= dpp_u<0x141>(v);
    return v;
}

template <typename T>
__global__ __launch_bounds__(kWave *kRowsPerBlock) void pack_count_kernel(
    const T *__restrict__ src, int64_t ld, const int64_t *__restrict__ rows, int64_t m, int F,
    uint32_t *__restrict__ bits, int32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (r >= m) return;
    const T *row = src + (rows ? rows[r] : r) * ld;
    const int words = F >> 5;
    int cnt = 0;
    for (int f0 = 0; f0 < F; f0 += 4 * kWave) {           // a slab of 256 elements per pass
        const int f = f0 + 4 * lane;
        uint32_t nib = 0u;
        if (f < F) {
            float x[4];
            Quad<T>::load(row + f, x);
#pragma unroll
            for (int k = 0; k < 4; ++k) nib |= (x[k] != 0.f ? 1u : 0u) << k;
        }
        cnt += __builtin_popcount(nib);
        const uint32_t w = or8(nib << (4 * (lane & 7)));   // the 8 lanes of a word hold its 8 nibbles
        if ((lane & 7) == 0 && f < F) bits[r * words + (f >> 5)] = w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, kWave);
    if (lane == 0) counts[r] = cnt;
}

template <typename T>
__global__ __launch_bounds__(kWave *kRowsPerBlock) void pack_values_kernel(
    const T *__restrict__ src, int64_t ld, const int64_t *__restrict__ rows, int64_t m, int F,
    const int64_t *__restrict__ offsets, T *__restrict__ vals)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (r >= m) return;
    const T *row = src + (rows ? rows[r] : r) * ld;
    int64_t base = offsets[r];
    for (int f0 = 0; f0 < F; f0 += 4 * kWave) {
        const int f = f0 + 4 * lane;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (f < F) Quad<T>::load(row + f, x);
        // position of element (lane, k) among the non-zeros of the slab, in row order (4·lane + k):
        // all non-zeros of lower lanes + the non-zeros of this lane at lower k
        unsigned long long b[4];
        int pos = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = __ballot(x[k] != 0.f);
            pos += below(b[k]);
            total += __builtin_popcountll(b[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x[k] != 0.f) {
                T out;
                if (sizeof(T) == 4) *(float *)&out = x[k];
                else *(bf16_t *)&out = (bf16_t)(__float_as_uint(x[k]) >> 16);
                vals[base + pos] = out;
                ++pos;
            }
        }
        base += total;
    }
}

template <typename T>
__global__ __launch_bounds__(kWave *kRowsPerBlock) void unpack_kernel(
    const uint32_t *__restrict__ bits, const int64_t *__restrict__ offsets, const T *__restrict__ vals,
    int64_t m, int F, T *__restrict__ dst, int64_t ldd)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (r >= m) return;
    const int words = F >> 5;
    int64_t base = offsets[r];
    for (int f0 = 0; f0 < F; f0 += 4 * kWave) {
        const int f = f0 + 4 * lane;
        uint32_t nib = 0u;
        if (f < F) nib = (bits[r * words + (f >> 5)] >> (4 * (lane & 7))) & 0xFu;
        float x[4];
        int pos = 0, total = 0;
        unsigned long long b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = __ballot(((nib >> k) & 1u) != 0u);
            pos += below(b[k]);
            total += __builtin_popcountll(b[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = 0.f;
            if ((nib >> k) & 1u) {
                const T v = vals[base + pos];
                x[k] = sizeof(T) == 4 ? *(const float *)&v : __uint_as_float((uint32_t)(*(const bf16_t *)&v) << 16);
                ++pos;
            }
        }
        if (f < F) Quad<T>::store(dst + r * ldd + f, x);
        base += total;
    }
}

// counts[r] = set bits of row r of bits [m, words] (the receiver's side of the scan: the wire carries
// no counts).  8 lanes per row, 16 bytes per lane per pass.
__global__ __launch_bounds__(256) void bits_count_kernel(const uint32_t *__restrict__ bits, int64_t m, int words,
                                                         int32_t *__restrict__ counts)
{
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    int c = 0;
    if (r < m)
        for (int w = sub; w < words; w += 8) c += __builtin_popcount(bits[r * words + w]);
    c += __shfl_xor(c, 1, kWave);
    c += __shfl_xor(c, 2, kWave);
    c += __shfl_xor(c, 4, kWave);
    if (r < m && sub == 0) counts[r] = c;
}

int check(const char *who, int dtype, int64_t m, int64_t F, int64_t ld)
{
    if (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16) return gcn_internal_fail(GCN_E_BADARG, who);
    if (m < 0 || F <= 0 || F % 32 != 0 || F > INT32_MAX || ld < F || ld % 4 != 0)
        return gcn_internal_fail(GCN_E_BADARG, who);
    return 0;
}

}   // namespace

extern "C" {

int gcn_rows_pack_count(int dtype, const void *src, int64_t ld, const int64_t *rows, int64_t m, int64_t F,
                        uint32_t *bits, int32_t *counts, void *stream)
{
    if (int rc = check("gcn_rows_pack_count: bad dtype / sizes (F a multiple of 32, ld a multiple of 4)", dtype, m, F, ld))
        return rc;
    if (m == 0) return 0;
    if (src == nullptr || bits == nullptr || counts == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_rows_pack_count: NULL pointer");
    if (((uintptr_t)src) % 16 != 0) return gcn_internal_fail(GCN_E_ALIGN, "gcn_rows_pack_count: 16-byte alignment required");
    const dim3 grid((unsigned)((m + kRowsPerBlock - 1) / kRowsPerBlock)), block(kWave * kRowsPerBlock);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCN_DTYPE_F32)
        hipLaunchKernelGGL(pack_count_kernel<float>, grid, block, 0, s, (const float *)src, ld, rows, m, (int)F, bits, counts);
    else
        hipLaunchKernelGGL(pack_count_kernel<bf16_t>, grid, block, 0, s, (const bf16_t *)src, ld, rows, m, (int)F, bits, counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_rows_pack_count launch");
}

int gcn_rows_pack_values(int dtype, const void *src, int64_t ld, const int64_t *rows, int64_t m, int64_t F,
                         const int64_t *offsets, void *vals, void *stream)
{
    if (int rc = check("gcn_rows_pack_values: bad dtype / sizes (F a multiple of 32, ld a multiple of 4)", dtype, m, F, ld))
        return rc;
    if (m == 0) return 0;
    if (src == nullptr || offsets == nullptr || vals == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_rows_pack_values: NULL pointer");
    if (((uintptr_t)src) % 16 != 0) return gcn_internal_fail(GCN_E_ALIGN, "gcn_rows_pack_values: 16-byte alignment required");
    const dim3 grid((unsigned)((m + kRowsPerBlock - 1) / kRowsPerBlock)), block(kWave * kRowsPerBlock);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCN_DTYPE_F32)
        hipLaunchKernelGGL(pack_values_kernel<float>, grid, block, 0, s, (const float *)src, ld, rows, m, (int)F, offsets, (float *)vals);
    else
        hipLaunchKernelGGL(pack_values_kernel<bf16_t>, grid, block, 0, s, (const bf16_t *)src, ld, rows, m, (int)F, offsets, (bf16_t *)vals);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_rows_pack_values launch");
}

int gcn_rows_unpack(int dtype, const uint32_t *bits, const int64_t *offsets, const void *vals, int64_t m,
                    int64_t F, void *dst, int64_t ldd, void *stream)
{
    if (int rc = check("gcn_rows_unpack: bad dtype / sizes (F a multiple of 32, ldd a multiple of 4)", dtype, m, F, ldd))
        return rc;
    if (m == 0) return 0;
    if (bits == nullptr || offsets == nullptr || dst == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_rows_unpack: NULL pointer");
    if (((uintptr_t)dst) % 16 != 0) return gcn_internal_fail(GCN_E_ALIGN, "gcn_rows_unpack: 16-byte alignment required");
    const dim3 grid((unsigned)((m + kRowsPerBlock - 1) / kRowsPerBlock)), block(kWave * kRowsPerBlock);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCN_DTYPE_F32)
        hipLaunchKernelGGL(unpack_kernel<float>, grid, block, 0, s, bits, offsets, (const float *)vals, m, (int)F, (float *)dst, ldd);
    else
        hipLaunchKernelGGL(unpack_kernel<bf16_t>, grid, block, 0, s, bits, offsets, (const bf16_t *)vals, m, (int)F, (bf16_t *)dst, ldd);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_rows_unpack launch");
}

int gcn_bits_row_counts(const uint32_t *bits, int64_t m, int64_t words, int32_t *counts, void *stream)
{
    if (m < 0 || words <= 0 || words > (1 << 20))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_bits_row_counts: bad sizes");
    if (m == 0) return 0;
    if (bits == nullptr || counts == nullptr) return gcn_internal_fail(GCN_E_BADARG, "gcn_bits_row_counts: NULL pointer");
    const dim3 grid((unsigned)((m * 8 + 255) / 256)), block(256);
    hipLaunchKernelGGL(bits_count_kernel, grid, block, 0, (hipStream_t)stream, bits, m, (int)words, counts);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_bits_row_counts launch");
}

}   // extern "C"
