// gcn_spmm.hip — CSR SpMM for the GraphConvolution hot path, written for gfx950 (MI355X, CDNA4).
//
// Implements the C-ABI of include/gcn_spmm.h.  What it replaces in the reference:
//   torch.spmm(adj, support)          pygcn/layers.py:34      -> gcn_spmm_csr on CSR(adj)
//   adj.t() @ grad_output (autograd)  pygcn/train.py:157      -> gcn_spmm_csr on CSR(adj^T)
//   output + self.bias                pygcn/layers.py:35-36   -> fused epilogue
//
// Execution model (see DESIGN.md §3 for the numbers):
//   * HBM-bound gather: every stored entry pulls one dense row B[col,:] (F*s bytes).  A 64-lane
//     wavefront reads one such row with ONE instruction (16 B per lane -> 1 KiB for fp32 F=256),
//     so all B traffic is full-line coalesced and the column index / value are wave-uniform
//     scalars (v_readlane from a 64-entry register tile, scalar address arithmetic).
//   * Latency is hidden by keeping D row-loads in flight per wave in a register ring that is
//     refilled as soon as a slot is consumed, times 8 waves per SIMD.
//   * Load balance on power-law graphs comes from the static schedule in gcn_csr_plan: short
//     rows are packed into equal-cost row-batch items (one wave each); rows longer than
//     long_thresh are cut into chunks summed by separate waves into an fp32 slab and added in
//     chunk order by a tiny second kernel — no float atomics, bitwise reproducible.
//   * Narrow feature widths (F*s < 1 KiB) put several stored entries of one row side by side in
//     one wave instruction (lane groups) and finish with a wavefront shuffle reduction.
//
// gfx950 only.  No other architecture is supported or intended.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "gcn_spmm.h"

// keep-threshold of the fused dropout: an element is kept iff its 16-bit field >= round(p * 2^16),
// clamped to [1, 65535] for p > 0 (0 = dropout off) — include/gcn_spmm.h, struct gcn_epilogue
static inline uint32_t gcn_dropout_threshold16(float p)
{
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 65536.0 + 0.5;
    return (uint32_t)std::min(65535.0, std::max(1.0, (double)(int64_t)t));
}
// the scale that goes with that threshold: the kernels keep an element with probability
// (65536 - T) / 65536, so 65536 / (65536 - T) — not 1 / (1 - p) of the unquantised p — makes
// E[dropout(x)] = x exactly (the two agree to 2^-17 away from the clamps, and exactly at p = 1/2)
static inline float gcn_dropout_scale16(uint32_t thresh)
{
    return thresh == 0u ? 1.f : 65536.f / (float)(65536u - thresh);
}

namespace {

constexpr int kWave = 64;
#ifndef GCN_WPB
#define GCN_WPB 4
#endif
constexpr int kWavesPerBlock = GCN_WPB;
constexpr int kDefaultItemCost = GCN_DEFAULT_ITEM_COST;
constexpr int kDefaultLongThresh = GCN_DEFAULT_LONG_THRESH;

thread_local char g_err[256] = "";

int fail(int code, const char *msg)
{
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int fail_hip(hipError_t e, const char *where)
{
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

struct KParams {
    const void *rowptr;
    const int32_t *col;
    const float *val;
    const int32_t *items;
    const int32_t *chunk_row;
    const int64_t *chunk_e0;
    const int32_t *long_row;
    const int32_t *long_chunk0;
    const void *B;
    void *C;
    const float *bias;
    float *partial;
    int64_t ldb;   // elements
    int64_t ldc;   // elements
    int32_t F;
    int32_t n_total;    // n_chunks + n_items
    int32_t n_chunks;
    int32_t n_long;
    int32_t long_thresh;
    int32_t relu;
    uint32_t drop_thresh;   // keep an element iff its 16 random bits >= drop_thresh (0: no dropout)
    float drop_scale;       // 65536 / (65536 - drop_thresh): 1 / (1 - p) of the quantised p
    uint32_t seed_lo, seed_hi;
    int64_t drop_row_base;  // added to the row index in the dropout counter (a shard's first row)
    const void *B2;          // optional second block of B: rows >= b_split live here (ldb2)
    int64_t ldb2;
    int32_t b_split;         // INT32_MAX when B is one block
    const uint32_t *bflag;   // optional bitmap [ceil(n_cols/32)]: bit c clear = row c of B is all zero
    const int32_t *bnnz;     // optional device scalar: number of non-zero rows of B
    int32_t n_cols;
    uint8_t *cflag;          // optional output [n_rows], pre-zeroed: 1 = stored row has a non-zero
    int32_t log_softmax;     // store log_softmax of the row (whole row inside one store_out call)
    const uint64_t *seed_dev;   // optional device-resident dropout seed (overrides seed_lo/hi)
    const uint32_t *csel;       // optional bitmap over output rows: clear bit = row not wanted
    int32_t cskip;              // with cflag: all-zero rows are not stored
    uint32_t *cabsmax;          // optional output: max |stored value| as BITS (atomic max; inf / NaN sort on top)
};

// ------------------------------------------------------------------------------------------
// element traits: how VEC elements of T travel between memory and fp32 registers
// ------------------------------------------------------------------------------------------
typedef uint16_t bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <typename T, int VEC> struct Elem;

template <> struct Elem<float, 4> {
    typedef f32x4 Raw;
    static __device__ __forceinline__ void unpack(const Raw &r, float (&x)[4])
    {
        x[0] = r.x; x[1] = r.y; x[2] = r.z; x[3] = r.w;
    }
    static __device__ __forceinline__ Raw pack(const float (&x)[4])
    {
        Raw r = {x[0], x[1], x[2], x[3]};
        return r;
    }
};
template <> struct Elem<float, 1> {
    typedef float Raw;
    static __device__ __forceinline__ void unpack(const Raw &r, float (&x)[1]) { x[0] = r; }
    static __device__ __forceinline__ Raw pack(const float (&x)[1]) { return x[0]; }
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);   // v_cvt_pk_bf16_f32, round-to-nearest-even
    return __builtin_bit_cast(uint32_t, b);
}
template <> struct Elem<bf16_t, 8> {
    typedef u32x4 Raw;
    static __device__ __forceinline__ void unpack(const Raw &r, float (&x)[8])
    {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(w[i] << 16);
            x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ Raw pack(const float (&x)[8])
    {
        Raw r = {pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]),
                 pack_bf16x2(x[6], x[7])};
        return r;
    }
};
template <> struct Elem<bf16_t, 1> {
    typedef bf16_t Raw;
    static __device__ __forceinline__ void unpack(const Raw &r, float (&x)[1])
    {
        x[0] = __uint_as_float(((uint32_t)r) << 16);
    }
    static __device__ __forceinline__ Raw pack(const float (&x)[1])
    {
        return (bf16_t)(pack_bf16x2(x[0], 0.f) & 0xffffu);
    }
};

// Row flags of a row-sparse dense operand are used only when they can pay: a device-side count
// (no host synchronisation) says fewer than num/den of the rows are non-zero.  The wide kernel
// (1-KiB rows, cheap scalar compaction) gains up to 3/4; the narrow kernel's lane compaction only
// below 1/8 (measured at C5: 16 % non-zero rows = 80 % surviving entries ran 47 ms hinted against
// 41 ms dense, 5 % ran 18 ms).
__device__ __forceinline__ bool use_row_flags(const KParams &p, int num, int den)
{
    if (p.bflag == nullptr || p.bnnz == nullptr) return false;
    const int nz = __builtin_amdgcn_readfirstlane(*p.bnnz);
    return (int64_t)nz * den < (int64_t)p.n_cols * num;
}

__device__ __forceinline__ bool row_bit(const uint32_t *__restrict__ bits, int c)
{
    return (bits[c >> 5] >> (c & 31)) & 1u;
}

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so the dropout mask of element (row, f)
// depends only on (seed, row, f) — never on which kernel variant, vector width, wave or launch
// produced the row.  One call yields 128 bits = EIGHT 16-bit keep fields (ABI 22; before: four
// 32-bit words — half the integer multiplies per element now) — or, at p = 1/2 where one bit
// decides, 128 ONE-bit fields (ABI 23, see apply_dropout).  Element (row, f), 16-bit form:
//     block = ((f >> 4) << 1) | ((f >> 2) & 1)          counter word 2
//     field = (((f >> 3) & 1) << 2) | (f & 3)           0..7: word field >> 1, half field & 1
//     w     = philox(counter = (row_lo, row_hi, block, 0), key = seed)   (row incl. drop_row_base)
//     keep  = ((w[field >> 1] >> 16 * (field & 1)) & 0xFFFF) >= round(p * 65536)
// The block groups columns {16c + 4b + (0..3), 16c + 4b + 8 + (0..3)} (b = bit 2 of f): exactly the
// eight columns one lane of the MFMA GEMMs' transposed accumulator tile stores per 16-column
// group (gcn_gemm.hip), so the GEMM epilogues need ONE call per eight stored elements too.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <int VEC>
__device__ __forceinline__ void apply_dropout(const KParams &p, int64_t row, int f, float (&o)[VEC])
{
    uint32_t r[4];
    uint32_t k0 = p.seed_lo, k1 = p.seed_hi;
    if (p.seed_dev != nullptr) {   // wave-uniform scalar load: the seed as of execution time
        const uint64_t sd = *p.seed_dev;
        k0 = (uint32_t)sd;
        k1 = (uint32_t)(sd >> 32);
    }
    row += p.drop_row_base;
    if (p.drop_thresh == 32768u) {
        // p = 1/2 (the reference's default): ONE bit decides, a call yields 128 one-bit fields —
        //     block = ((f >> 8) << 1) | ((f >> 2) & 1),   index = (((f & 255) >> 3) << 2) | (f & 3)
        //     keep  = (w[index >> 5] >> (index & 31)) & 1
        // the same layout rule at a 256-column span: a lane of the GEMMs' transposed accumulator
        // tile finds ALL its columns of a 256-wide row in one call (16 calls per tile before).
#pragma unroll
        for (int q = 0; q < (VEC + 3) / 4; ++q) {
            const int fq = f + 4 * q;
            const uint32_t blk = ((uint32_t)(fq >> 8) << 1) | ((uint32_t)(fq >> 2) & 1u);
            const int idx = (((fq & 255) >> 3) << 2) | (fq & 3);          // (VEC >= 4: fq & 3 == 0)
            philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), blk, 0u, k0, k1, r);
            const int ws = idx >> 5;
            const uint32_t w = (ws & 2) ? ((ws & 1) ? r[3] : r[2]) : ((ws & 1) ? r[1] : r[0]);
            const uint32_t nib = w >> (idx & 31);
#pragma unroll
            for (int j = 0; j < (VEC == 1 ? 1 : 4); ++j)
                o[4 * q + j] = ((nib >> j) & 1u) ? o[4 * q + j] * p.drop_scale : 0.f;
        }
        return;
    }
    if (VEC == 1) {
        const uint32_t blk = ((uint32_t)(f >> 4) << 1) | ((uint32_t)(f >> 2) & 1u);
        const int fld = (((f >> 3) & 1) << 2) | (f & 3);
        philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), blk, 0u, k0, k1, r);
        const uint32_t w = (fld & 4) ? ((fld & 2) ? r[3] : r[2]) : ((fld & 2) ? r[1] : r[0]);
        const uint32_t bits = (fld & 1) ? (w >> 16) : (w & 0xFFFFu);
        o[0] = (bits >= p.drop_thresh) ? o[0] * p.drop_scale : 0.f;
    } else {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {       // 4 consecutive columns = fields 4s .. 4s+3 of one block
            const int fq = f + 4 * q;
            const uint32_t blk = ((uint32_t)(fq >> 4) << 1) | ((uint32_t)(fq >> 2) & 1u);
            philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), blk, 0u, k0, k1, r);
            const bool hi = ((fq >> 3) & 1) != 0;
            const uint32_t w0 = hi ? r[2] : r[0], w1 = hi ? r[3] : r[1];
            o[4 * q + 0] = ((w0 & 0xFFFFu) >= p.drop_thresh) ? o[4 * q + 0] * p.drop_scale : 0.f;
            o[4 * q + 1] = ((w0 >> 16) >= p.drop_thresh) ? o[4 * q + 1] * p.drop_scale : 0.f;
            o[4 * q + 2] = ((w1 & 0xFFFFu) >= p.drop_thresh) ? o[4 * q + 2] * p.drop_scale : 0.f;
            o[4 * q + 3] = ((w1 >> 16) >= p.drop_thresh) ? o[4 * q + 3] * p.drop_scale : 0.f;
        }
    }
}

// xor-butterfly reductions inside groups of LPR consecutive lanes (LPR a power of two).  The first
// four steps are DPP row operations — VALU modifiers, no trip through the LDS crossbar and no
// address registers: quad_perm [1,0,3,2] and [2,3,0,1] are the xor-1 / xor-2 exchanges; after them
// a quad is uniform, so row_half_mirror (lane i <-> 7 - i) and row_mirror (i <-> 15 - i) act as
// xor 4 / xor 8.  max and + are commutative, so every lane of a group ends with the same bits.
// Wider groups finish with wave shuffles (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int LPR>
__device__ __forceinline__ float group_max(float m)
{
    if (LPR >= 2) m = fmaxf(m, dpp_move<0xB1>(m));
    if (LPR >= 4) m = fmaxf(m, dpp_move<0x4E>(m));
    if (LPR >= 8) m = fmaxf(m, dpp_move<0x141>(m));
    if (LPR >= 16) m = fmaxf(m, dpp_move<0x140>(m));
#pragma unroll
    for (int off = 16; off < LPR; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    return m;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float s)
{
    if (LPR >= 2) s += dpp_move<0xB1>(s);
    if (LPR >= 4) s += dpp_move<0x4E>(s);
    if (LPR >= 8) s += dpp_move<0x141>(s);
    if (LPR >= 16) s += dpp_move<0x140>(s);
#pragma unroll
    for (int off = 16; off < LPR; off <<= 1) s += __shfl_xor(s, off, kWave);
    return s;
}

// bias add + ReLU + dropout + conversion + store of one output row segment (VEC floats per lane)
// (a row is held by LPR consecutive lanes, VEC elements each, starting at a multiple of LPR)
// XEPI = 0 compiles the log_softmax / output-row-flag code out, 1 keeps log_softmax only (the last
// layer of every forward pass: without the row-flag code the bf16 narrow instantiation keeps 6
// waves per SIMD), 2 keeps both: the plain instantiations keep
// the register budget they were tuned with (wide fp32: 62 VGPRs; the extras cost 8-15 more)
// Returns (XEPI >= 2 with p.cabsmax set, else 0) the largest |stored value| of this lane as its bit
// pattern: callers keep a running maximum per lane and publish it once per wave (publish_absmax).
template <typename T, int VEC, int LPR, int XEPI, bool ALLOW_SKIP = true>
__device__ __forceinline__ uint32_t store_out(const KParams &p, int64_t row, int f, bool act,
                                              const float (&acc)[VEC], const float (&bias)[VEC])
{
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        o[i] = acc[i] + bias[i];
        if (p.relu) o[i] = fmaxf(o[i], 0.f);
    }
    if (p.drop_thresh != 0u) apply_dropout<VEC>(p, row, f, o);   // wave-uniform branch
    if (XEPI >= 1 && p.log_softmax) {   // wave-uniform; the host guarantees the whole row sits in these LPR lanes
        const bool valid = f < p.F;
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < VEC; ++i) m = valid ? fmaxf(m, o[i]) : m;
        m = group_max<LPR>(m);
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) se += valid ? __expf(o[i] - m) : 0.f;   // arguments <= 0
        se = group_sum<LPR>(se);
        const float lse = m + __logf(se);   // se in [1, F]: hardware exp2 / log2, ~1e-7 absolute
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] -= lse;
    }
    bool do_store = act;
    if (XEPI >= 2 && p.cflag != nullptr) {   // wave-uniform; every lane holding a non-zero stores the same byte
        bool nz = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) nz |= (o[i] != 0.f);
        if (ALLOW_SKIP && p.cskip) {
            // an all-zero row is not written at all: the decision must be the same for the LPR
            // lanes that hold the row -> their slice of the wave's ballot
            const unsigned long long b = __ballot(act && nz);
            const int ln = (int)(threadIdx.x & (kWave - 1));
            const unsigned long long gm =
                LPR >= kWave ? ~0ull : (((1ull << LPR) - 1ull) << (ln & ~(LPR - 1)));
            do_store = act && (b & gm) != 0ull;
        }
        if (act && nz) p.cflag[row] = 1;
    }
    if (do_store) {
        T *dst = (T *)p.C + row * p.ldc + f;
#if SPMM_STORE_NT
        __builtin_nontemporal_store(Elem<T, VEC>::pack(o), (typename Elem<T, VEC>::Raw *)dst);
#else
        *(typename Elem<T, VEC>::Raw *)dst = Elem<T, VEC>::pack(o);
#endif
    }
    uint32_t amax = 0u;
    if (XEPI >= 2 && p.cabsmax != nullptr && do_store) {   // (uniform pointer test)
        float st[VEC];                                       // the values as STORED (bf16: rounded)
        Elem<T, VEC>::unpack(Elem<T, VEC>::pack(o), st);
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (f + i < p.F) amax = max(amax, __float_as_uint(st[i]) & 0x7fffffffu);
    }
    return amax;
}

// One atomic per wave at most, and only when the wave's maximum beats the value already published
// (a plain load at the end of the wave's work: its latency hides nothing that matters).
template <int XEPI>
__device__ __forceinline__ void publish_absmax(const KParams &p, uint32_t amax)
{
    if (XEPI >= 2 && p.cabsmax != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, off, kWave));
        if ((threadIdx.x & (kWave - 1)) == 0 && amax != 0u &&
            amax > __atomic_load_n(p.cabsmax, __ATOMIC_RELAXED))
            atomicMax(p.cabsmax, amax);
    }
}

// ------------------------------------------------------------------------------------------
// WIDE kernel: one dense row segment (64 lanes x VEC elements) per wave instruction.
// Used when F/VEC > 32 lanes; blockIdx.y walks further 64*VEC-wide feature slabs.
// ------------------------------------------------------------------------------------------
// 16-byte load of a dense-row segment through a buffer descriptor built from wave-uniform
// scalars: base = B + col*ldb (SGPR pair), num_records = row bytes.  The hardware range check
// returns zeros for lanes past the end of the row (feature tail) and for whole slots whose
// num_records is 0 (slots past the end of an edge tile), so the gather needs no branches.
#ifndef SPMM_GATHER_AUX    /* experiment builds (tools/build_spmm_variants.sh): cache policy of the gather */
#define SPMM_GATHER_AUX 0  /* 0 default, 2 = nt (streaming) */
#endif
#ifndef SPMM_HUB_TAG       /* 1: bit 31 of a column index marks a HUB column — its row is loaded with the */
#define SPMM_HUB_TAG 0     /*    default policy, every other row with SPMM_GATHER_AUX                     */
#endif
#ifndef SPMM_STORE_NT      /* 1: the result rows are stored non-temporally */
#define SPMM_STORE_NT 0
#endif
template <int AUX = SPMM_GATHER_AUX>
__device__ __forceinline__ u32x4 row_load16(uint64_t base, uint32_t nbytes, uint32_t voff)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    void *pb = (void *)(((uint64_t)hi << 32) | lo);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(pb, 0, nbytes, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, AUX);
}

template <typename T, int VEC, int D, bool ROWS, bool FLAGS, int XEPI>
__device__ __forceinline__ void wide_stream(const KParams &p, const int32_t *__restrict__ colp,
                                            const float *__restrict__ valp, int ne, int lane,
                                            unsigned ld_off_bytes, float (&acc)[VEC],
                                            int rel_end, int nr, int64_t row0, int f, bool act,
                                            const float (&bias)[VEC], uint32_t &amax)
{
    typedef typename Elem<T, VEC>::Raw Raw;
    static_assert(sizeof(Raw) == 16, "wide kernel moves 16 bytes per lane");
    const uint32_t ldb_bytes = (uint32_t)(p.ldb * (int64_t)sizeof(T));   // < 4 GiB, host-checked
    const uint32_t ldb2_bytes = (uint32_t)(p.ldb2 * (int64_t)sizeof(T));
    const uint32_t row_bytes = (uint32_t)p.F * (uint32_t)sizeof(T);
    // row c of the dense operand: in B below the split, in B2 (rebased) from the split on; all
    // scalar arithmetic (the sharded path keeps a rank's own rows and its halo rows apart)
    auto row_base = [&](int c) -> uint64_t {
        return c < p.b_split ? (uint64_t)p.B + (uint64_t)(uint32_t)c * ldb_bytes
                             : (uint64_t)p.B2 + (uint64_t)(uint32_t)(c - p.b_split) * ldb2_bytes;
    };
    int r = 0;
    int rend = ROWS ? readlane_i(rel_end, 0) : INT_MAX;
    const bool flags = FLAGS && use_row_flags(p, 3, 4);   // wave-uniform; FLAGS = false: dense operand,
                                                    // the flag code is compiled out
    const bool sel = FLAGS && ROWS && p.csel != nullptr;  // output-row selection (wave-uniform)
    // lane l < nr: is row row0 + l wanted
    const int want = (sel && lane < nr) ? (int)row_bit(p.csel, (int)row0 + lane) : 1;
    auto emit = [&](int rr) {   // store row rr of the item unless the caller does not want it
        if (!sel || readlane_i(want, rr))
            amax = max(amax, store_out<T, VEC, kWave, XEPI>(p, row0 + rr, f, act, acc, bias));
    };
    auto consume = [&](int e, const u32x4 &raw, float a) {
        if (ROWS) {
            while (e >= rend) {   // row finished (loop: rows without stored entries follow)
                emit(r);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
                ++r;
                rend = readlane_i(rel_end, r);
            }
        }
        float x[VEC];
        Elem<T, VEC>::unpack(__builtin_bit_cast(Raw, raw), x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a, x[i], acc[i]);
    };

    for (int t = 0; t < ne; t += kWave) {
        const int cnt = min(kWave, ne - t);
        int cv = 0;
        float vv = 0.f;   // lanes >= cnt keep value 0: slots past the tile add 0 * 0
        if (lane < cnt) {
            cv = colp[t + lane];
            vv = valp[t + lane];
        }
        // row-sparse operand: one byte gather per tile tells which of the 64 rows are all-zero;
        // their slots get num_records = 0 like the slots past the end of the tile (no traffic)
        if (FLAGS && (flags || sel)) {
            // row-sparse operand: one bitmap probe per lane tells which of the tile's rows of B are
            // all-zero; only the stored entries whose bit is set are visited (ascending order, so
            // the row bookkeeping of consume() is unchanged), D at a time
            int fv = (lane < cnt) ? 1 : 0;
            if (flags) fv = fv && row_bit(p.bflag, cv);
            if (sel) {
                // output-row selection: the row of entry i = t + lane is the first r with
                // rel_end[r] > i (binary search over the row ends held in lanes 0..nr-1); entries
                // of rows the caller does not want are not visited either
                const int i = t + lane;
                int lo = 0, hi = nr - 1;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const int mid = (lo + hi) >> 1;
                    const int v = __shfl(rel_end, mid, kWave);
                    if (v > i) hi = mid; else lo = min(mid + 1, nr - 1);
                }
                // (the shuffle must run in ALL lanes: a lane switched off by a short-circuit would
                //  supply 0 to the lanes that read it)
                const int w = __shfl(want, lo, kWave);
                fv &= w;
            }
            unsigned long long m = __ballot(fv != 0);
            while (m) {
                int kk[D];
                u32x4 x[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    kk[j] = m ? (int)__builtin_ctzll(m) : -1;
                    m &= m - 1;   // (0 stays 0)
                    const int c = readlane_i(cv, kk[j] < 0 ? 0 : kk[j]);
                    x[j] = row_load16(row_base(c), kk[j] >= 0 ? row_bytes : 0u, ld_off_bytes);
                }
#pragma unroll
                for (int j = 0; j < D; ++j)
                    if (kk[j] >= 0) consume(t + kk[j], x[j], readlane_f(vv, kk[j]));
            }
            continue;
        }
        for (int k = 0; k < cnt; k += D) {
            // D row loads in flight, branch-free
            u32x4 x[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const bool ok = k + j < cnt;
#if SPMM_HUB_TAG
                const int ct = readlane_i(cv, k + j);
                const int c = ct & 0x7fffffff;
                if (ct < 0) x[j] = row_load16<0>(row_base(c), ok ? row_bytes : 0u, ld_off_bytes);
                else x[j] = row_load16<SPMM_GATHER_AUX>(row_base(c), ok ? row_bytes : 0u, ld_off_bytes);
#else
                const int c = readlane_i(cv, k + j);   // k + j <= 63 always (k <= 56)
                x[j] = row_load16(row_base(c), ok ? row_bytes : 0u, ld_off_bytes);
#endif
            }
#pragma unroll
            for (int j = 0; j < D; ++j)
                consume(min(t + k + j, ne - 1), x[j], readlane_f(vv, k + j));
        }
    }
    if (ROWS) {
        while (r < nr) {   // last row of the item and any trailing empty rows
            emit(r);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            ++r;
        }
    }
}

template <typename T, int VEC, typename IdxT, int D, bool FLAGS, int XEPI>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_wide_kernel(KParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int item =
        blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= p.n_total) return;
    const int f = blockIdx.y * (kWave * VEC) + lane * VEC;
    const bool act = f < p.F;
    // lanes past the end of the row (F not a multiple of 64*VEC) are zeroed by the range check
    const unsigned ld_off_bytes = (unsigned)f * (unsigned)sizeof(T);
    const IdxT *__restrict__ rp = (const IdxT *)p.rowptr;

    float acc[VEC], bias[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        acc[i] = 0.f;
        bias[i] = 0.f;
    }

    if (item < p.n_chunks) {
        // ---- one chunk of a long row -> fp32 partial slab
        const int row = p.chunk_row[item];
        if (FLAGS && p.csel != nullptr && !row_bit(p.csel, row)) return;   // long row not wanted
        const int64_t e0 = p.chunk_e0[item];
        const int64_t e1 = min(e0 + (int64_t)p.long_thresh, (int64_t)rp[row + 1]);
        uint32_t unused = 0u;
        wide_stream<T, VEC, D, false, FLAGS, XEPI>(p, p.col + e0, p.val + e0, (int)(e1 - e0), lane,
                                      ld_off_bytes, acc, 0, 0, 0, f, act, bias, unused);
        if (act) {
            float *dst = p.partial + (int64_t)item * p.F + f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i] = acc[i];
        }
        return;
    }

    // ---- a row-batch item: rows [ra, rb), <= 64 rows, contiguous stored entries
    const int it = item - p.n_chunks;
    const int ra = p.items[2 * it], rb = p.items[2 * it + 1];
    const int nr = rb - ra;
    const int64_t ea = (int64_t)rp[ra];
    const int rel_end = (lane < nr) ? (int)((int64_t)rp[ra + 1 + lane] - ea) : 0;
    const int ne = readlane_i(rel_end, nr - 1);
    if (p.bias != nullptr && act) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) bias[i] = p.bias[f + i];
    }
    uint32_t amax = 0u;
    wide_stream<T, VEC, D, true, FLAGS, XEPI>(p, p.col + ea, p.val + ea, ne, lane, ld_off_bytes, acc, rel_end,
                                 nr, (int64_t)ra, f, act, bias, amax);
    publish_absmax<XEPI>(p, amax);
}

// ------------------------------------------------------------------------------------------
// NARROW kernel: a dense row segment needs only LPR (< 64, or 64 with scalar elements) lanes,
// so G = 64/LPR stored entries of the SAME output row travel side by side in one wave
// instruction; the G partial sums meet in a wavefront shuffle reduction.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ const char *narrow_row_ptr(const KParams &p, int c)
{
    return c < p.b_split ? (const char *)p.B + (int64_t)c * (p.ldb * (int64_t)sizeof(T))
                         : (const char *)p.B2 + (int64_t)(c - p.b_split) * (p.ldb2 * (int64_t)sizeof(T));
}

template <typename T, int VEC, int LPR, int U>
__device__ __forceinline__ void narrow_row(const KParams &p, const int32_t *__restrict__ col,
                                           const float *__restrict__ val, int64_t e0, int64_t e1,
                                           int g, unsigned ld_off, float (&acc)[VEC],
                                           bool flags = false)
{
    typedef typename Elem<T, VEC>::Raw Raw;
    constexpr int G = kWave / LPR;
    if (flags) {
        // Row-sparse dense operand: per 64-entry tile probe the row bitmap once per lane, move the
        // surviving entries to the front of the wave (a full lane permutation: survivors keep
        // their order in [0, nv), the rest go behind), then split ONLY the survivors over the G
        // lane groups.  Skipped entries cost neither a bitmap-dependent stall per entry nor a load.
        const int lane = (int)(threadIdx.x & (kWave - 1));
        for (int64_t t = e0; t < e1; t += kWave) {
            const int cnt = (int)min((int64_t)kWave, e1 - t);
            int c = 0;
            float a = 0.f;
            bool keep = false;
            if (lane < cnt) {
                c = col[t + lane];
                a = val[t + lane];
                keep = row_bit(p.bflag, c);
            }
            const unsigned long long m = __ballot(keep);
            const int nv = __builtin_popcountll(m);
            if (nv == 0) continue;   // wave-uniform
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int dest = keep ? below : nv + (lane - below);   // a permutation of 0..63
            const int cc = __builtin_amdgcn_ds_permute(dest << 2, c);
            const float aa = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(a)));
            for (int k = 0; k < nv; k += G * U) {   // uniform trip count
                Raw x[U];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = k + u * G + g;
                    const bool ok = idx < nv;
                    const int ck = __shfl(cc, idx & (kWave - 1), kWave);
                    const float ak = __shfl(aa, idx & (kWave - 1), kWave);
                    w[u] = ok ? ak : 0.f;
                    Raw z = {};
                    x[u] = z;
                    if (ok) x[u] = *(const Raw *)(narrow_row_ptr<T>(p, ck) + ld_off);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {   // unloaded slots hold zeros with weight 0: exact
                    float xf[VEC];
                    Elem<T, VEC>::unpack(x[u], xf);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(w[u], xf[i], acc[i]);
                }
            }
        }
    } else
    for (int64_t e = e0; e < e1; e += (int64_t)G * U) {   // uniform trip count
        int c[U];
        float a[U];
        Raw x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ee = e + (int64_t)u * G + g;
            const bool ok = ee < e1;
            c[u] = ok ? col[ee] : 0;
            a[u] = ok ? val[ee] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // entries past the end of the row point at row 0 (a harmless cached read); their
            // products are skipped below, so non-finite values there cannot leak in
            x[u] = *(const Raw *)(narrow_row_ptr<T>(p, c[u]) + ld_off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ee = e + (int64_t)u * G + g;
            if (ee < e1) {
                float xf[VEC];
                Elem<T, VEC>::unpack(x[u], xf);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a[u], xf[i], acc[i]);
            }
        }
    }
    if (G > 1) {
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
        }
    }
}

template <typename T, int VEC, int LPR, typename IdxT, int XEPI>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void spmm_narrow_kernel(KParams p)
{
    constexpr int U = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int item =
        blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= p.n_total) return;
    const int g = lane / LPR, l = lane % LPR;
    const int f = blockIdx.y * (LPR * VEC) + l * VEC;
    const bool act = f < p.F;
    const unsigned ld_off = act ? (unsigned)f * (unsigned)sizeof(T) : 0u;
    const IdxT *__restrict__ rp = (const IdxT *)p.rowptr;
    const bool flags = use_row_flags(p, 1, 8);   // wave-uniform

    float acc[VEC], bias[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        acc[i] = 0.f;
        bias[i] = 0.f;
    }

    const bool sel = p.csel != nullptr;   // output-row selection (wave-uniform)
    uint32_t amax = 0u;                   // running max |stored value| of this lane (see store_out)
    if (item < p.n_chunks) {
        const int row = p.chunk_row[item];
        if (sel && !row_bit(p.csel, row)) return;   // long row not wanted
        const int64_t e0 = p.chunk_e0[item];
        const int64_t e1 = min(e0 + (int64_t)p.long_thresh, (int64_t)rp[row + 1]);
        narrow_row<T, VEC, LPR, U>(p, p.col, p.val, e0, e1, g, ld_off, acc, flags);
        if (act && g == 0) {
            float *dst = p.partial + (int64_t)item * p.F + f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) dst[i] = acc[i];
        }
        return;
    }

    const int it = item - p.n_chunks;
    const int ra = p.items[2 * it], rb = p.items[2 * it + 1];
    const int nr = rb - ra;
    if (p.bias != nullptr && act) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) bias[i] = p.bias[f + i];
    }
    constexpr int G = kWave / LPR;
    const int64_t ea = (int64_t)rp[ra];
    const int rel_end = (lane < nr) ? (int)((int64_t)rp[ra + 1 + lane] - ea) : 0;
    const int ne = readlane_i(rel_end, nr - 1);

    if (G == 1 || nr == 1 || ne > kWave) {
        // one row at a time, its stored entries split over the G lane groups
        int64_t e0 = ea;
        for (int r = ra; r < rb; ++r) {
            const int64_t e1 = (int64_t)rp[r + 1];
            if (sel && !row_bit(p.csel, r)) {   // row not wanted: neither computed nor stored
                e0 = e1;
                continue;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
            narrow_row<T, VEC, LPR, U>(p, p.col, p.val, e0, e1, g, ld_off, acc, flags);
            amax = max(amax, store_out<T, VEC, LPR, XEPI>(p, (int64_t)r, f, act && g == 0, acc, bias));
            e0 = e1;
        }
        publish_absmax<XEPI>(p, amax);
        return;
    }

    // ---- row-per-group: each lane group owns a different SHORT row of the item, so G*RU rows
    // are in flight for the accumulator cost of RU rows and no cross-lane reduction is needed.
    // Row ends and the (col, val) tile of the item (<= 64 entries) sit in lanes and are read
    // with ds_bpermute: no dependent memory round trip per row.
    typedef typename Elem<T, VEC>::Raw Raw;
#ifndef GCN_RU      // tuned on MI355X (C4 graph): RU 2 / UU 2 / kShort 8 beat 1,3,4 / 1,4 / 4,16,32
#define GCN_RU 2
#endif
#ifndef GCN_UU
#define GCN_UU 2
#endif
#ifndef GCN_KSHORT
#define GCN_KSHORT 8
#endif
    constexpr int RU = GCN_RU;                // rounds in flight
    constexpr int UU = GCN_UU;                // entries per row per step
    constexpr int kShort = GCN_KSHORT;        // rows longer than this take the edge-split path
    const int64_t ldb_bytes = p.ldb * (int64_t)sizeof(T);
    int cv = 0;
    float vv = 0.f;
    if (lane < ne) {
        cv = p.col[ea + lane];
        vv = p.val[ea + lane];
        if (flags && !row_bit(p.bflag, cv)) vv = 0.f, cv = -1;   // all-zero row of B: skip its gather
    }
    const int up = __shfl_up(rel_end, 1, kWave);
    const int rel_start = (lane == 0) ? 0 : up;
    // lane r < nr: is row ra + r wanted by the caller (c_row_select)
    const int wantn = (sel && lane < nr) ? (int)row_bit(p.csel, ra + lane) : 1;
    unsigned long long medium = __ballot(lane < nr && rel_end - rel_start > kShort && wantn != 0);

    for (int q0 = 0; q0 < nr; q0 += G * RU) {
        int e0[RU], e1[RU], row[RU];
        float a2[RU][VEC];
#pragma unroll
        for (int ru = 0; ru < RU; ++ru) {
            const int r = q0 + ru * G + g;
            const int s0 = __shfl(rel_start, r & (kWave - 1), kWave);
            const int s1 = __shfl(rel_end, r & (kWave - 1), kWave);
            const int wr = __shfl(wantn, r & (kWave - 1), kWave);   // (all lanes run the shuffle)
            const bool mine = r < nr && s1 - s0 <= kShort && wr != 0;
            row[ru] = mine ? r : -1;
            e0[ru] = s0;
            e1[ru] = mine ? s1 : s0;
#pragma unroll
            for (int i = 0; i < VEC; ++i) a2[ru][i] = 0.f;
        }
        for (int j = 0; j < kShort; j += UU) {
            bool more = false;
#pragma unroll
            for (int ru = 0; ru < RU; ++ru) more |= (e0[ru] + j < e1[ru]);
            if (!__any(more)) break;
            Raw x[RU][UU];
            float av[RU][UU];
#pragma unroll
            for (int ru = 0; ru < RU; ++ru) {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const int idx = e0[ru] + j + u;
                    const int c = __shfl(cv, idx & (kWave - 1), kWave);
                    const bool ok = idx < e1[ru] && c >= 0;
                    const float a = __shfl(vv, idx & (kWave - 1), kWave);
                    av[ru][u] = ok ? a : 0.f;
                    Raw z = {};
                    x[ru][u] = z;
                    if (ok)
                        x[ru][u] = *(const Raw *)(narrow_row_ptr<T>(p, c) + ld_off);
                }
            }
#pragma unroll
            for (int ru = 0; ru < RU; ++ru) {
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    // slots that were not loaded hold zeros with weight 0: adding them is exact
                    float xf[VEC];
                    Elem<T, VEC>::unpack(x[ru][u], xf);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) a2[ru][i] = fmaf(av[ru][u], xf[i], a2[ru][i]);
                }
            }
        }
#pragma unroll
        for (int ru = 0; ru < RU; ++ru) {
            if (row[ru] >= 0)
                amax = max(amax, store_out<T, VEC, LPR, XEPI>(p, (int64_t)(ra + row[ru]), f, act, a2[ru], bias));
        }
    }

    // ---- the item's medium rows (kShort < entries <= long_thresh): entries split over the
    // lane groups, wavefront shuffle reduction
    while (medium) {
        const int rr = __builtin_ctzll(medium);
        medium &= medium - 1;
        const int s0 = readlane_i(rel_start, rr), s1 = readlane_i(rel_end, rr);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        narrow_row<T, VEC, LPR, U>(p, p.col, p.val, ea + s0, ea + s1, g, ld_off, acc, flags);
        amax = max(amax, store_out<T, VEC, LPR, XEPI>(p, (int64_t)(ra + rr), f, act && g == 0, acc, bias));
    }
    publish_absmax<XEPI>(p, amax);
}

// ------------------------------------------------------------------------------------------
// long rows: add the chunk partials in chunk order, apply the epilogue, write the row
// ------------------------------------------------------------------------------------------
// Sum of column f over the chunk partials c0 .. c1-1 of one long row: SIXTEEN independent running
// sums — sixteen loads in flight per thread — combined in a fixed order: the result depends only
// on (c0, c1), never on scheduling.  (One running sum made every load wait for the previous add:
// a hub row of config C5 has > 1 000 chunks, and the serial chain of L2 round trips of that ONE
// row set the kernel's time, 1.0-1.7 ms per launch.)
__device__ __forceinline__ float long_row_sum(const float *__restrict__ partial, int F, int f, int c0, int c1)
{
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] += partial[(int64_t)(c + k) * F + f];
    }
#pragma unroll
    for (int k = 0; k < 15; ++k)
        if (c + k < c1) s[k] += partial[(int64_t)(c + k) * F + f];
    return (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) +
           (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
}

// ONE WAVE per long row, four rows per workgroup: config C5 has 7.4e5 long rows of 4.5 chunks on
// average (long_thresh 256), and a 256-thread workgroup per row — half of its threads idle at
// F = 128 — made the launch dispatch-bound (1.1 ms for 1.7 GB of partials).  A lane owns the
// columns lane, lane + 64, ...: one chunk row is a coalesced 256-byte read per pass.
template <typename T>
__global__ __launch_bounds__(256) void spmm_long_reduce_kernel(KParams p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int j = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (j >= p.n_long) return;
    const int64_t row = p.long_row[j];
    if (p.csel != nullptr && !row_bit(p.csel, (int)row)) return;   // (its chunks were skipped too)
    const int c0 = p.long_chunk0[j], c1 = p.long_chunk0[j + 1];
    uint32_t amax = 0u;
    if (p.log_softmax) {
        // F <= 512 (host check): the lane owns columns lane + 64k, k < 8; wave-wide max and sum
        // (8-byte loads of column pairs were measured: no faster — the launch is bound by its
        //  dependent index loads per row, not by the width of the partial reads)
        auto col = [&](int k) { return lane + kWave * k; };
        float z[8], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int f = col(k);
            z[k] = -INFINITY;
            if (f < p.F) z[k] = long_row_sum(p.partial, p.F, f, c0, c1) + (p.bias ? p.bias[f] : 0.f);
            m = fmaxf(m, z[k]);
        }
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) se += (col(k) < p.F) ? expf(z[k] - m) : 0.f;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) se += __shfl_xor(se, off, kWave);
        const float lse = m + logf(se);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int f = col(k);
            if (f < p.F) {
                const float o[1] = {z[k] - lse};
                ((T *)p.C)[row * p.ldc + f] = Elem<T, 1>::pack(o);
                if (p.cflag != nullptr && o[0] != 0.f) p.cflag[row] = 1;
                float st[1];
                Elem<T, 1>::unpack(Elem<T, 1>::pack(o), st);
                amax = max(amax, __float_as_uint(st[0]) & 0x7fffffffu);
            }
        }
        publish_absmax<2>(p, amax);
        return;
    }
    for (int f = lane; f < p.F; f += kWave) {
        const float s = long_row_sum(p.partial, p.F, f, c0, c1);
        float a[1] = {s};
        float b[1] = {p.bias ? p.bias[f] : 0.f};
        amax = max(amax, store_out<T, 1, 1, 2, false>(p, row, f, true, a, b));   // long rows: always stored
    }
    publish_absmax<2>(p, amax);
}

// ------------------------------------------------------------------------------------------
// backward of the fused ReLU + inverted-dropout epilogue: out = relu(z) * mask / (1 - p), so
// out > 0 <=> (z > 0 and kept), and grad_z = grad_out * scale * [out > 0].  One streaming pass,
// 16 B per lane, grid-stride (HBM-bound: 2 reads + 1 write per element).
// ------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const T *__restrict__ grad_out,
                                                               const T *__restrict__ out,
                                                               T *__restrict__ grad_pre,
                                                               int64_t n_units, float scale)
{
    typedef typename Elem<T, VEC>::Raw Raw;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_units; i += stride) {
        float g[VEC], o[VEC];
        Elem<T, VEC>::unpack(((const Raw *)grad_out)[i], g);
        Elem<T, VEC>::unpack(((const Raw *)out)[i], o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) g[k] = o[k] > 0.f ? g[k] * scale : 0.f;
        ((Raw *)grad_pre)[i] = Elem<T, VEC>::pack(g);
    }
}

// ------------------------------------------------------------------------------------------
// the same pass with the column sums of its result (grad_bias = sum over rows of grad_pre,
// autograd of `output + self.bias`, pygcn/layers.py:35-36) produced on the fly.  A 256-thread
// block owns a contiguous slab of rows; thread t owns the 4 columns 4*(t % CG) (CG = F/4 column
// groups) of every (256/CG)-th row, keeps 4 running sums, and the block writes one partial row
// [F]; a second tiny kernel adds the partial rows in block order (deterministic, no atomics).
// MODE 0 gives a plain column sum (layer without a fused ReLU), MODE 1 the ReLU / dropout mask,
// MODE 2 the backward of a fused log_softmax: grad_pre = g - exp(out) * rowsum(g), the row sum by
// xor-shuffles over the row's CG <= 64 lanes; rows of g that are entirely zero (every vertex
// outside idx_train) are written as zeros without reading `out`.
// MODE 3 is MODE 2 for the gradient of a mean NLL loss over ALL rows (F.nll_loss(output, labels),
// reference pygcn/train.py:153 without the index selection): g has one non-zero per row,
// g[r][target[r]] = coef, so it is never materialised — grad_pre = coef * (onehot(target[r]) -
// exp(out)), straight from `out` and the label vector: 2 instead of 4 full-height streams.
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void bwd_colsum_kernel(const T *__restrict__ grad_out,
                                                         const T *__restrict__ out,
                                                         T *__restrict__ grad_pre,
                                                         float *__restrict__ partial, int64_t n_rows,
                                                         int F, float scale, int rows_per_block,
                                                         uint8_t *__restrict__ rowflag,
                                                         int32_t *__restrict__ nnz_rows, int skip,
                                                         const int64_t *__restrict__ target,
                                                         const float *__restrict__ coef)
{
    typedef typename Elem<T, VEC>::Raw Raw;
    __shared__ float red[256 * VEC];
    const float cf = MODE == 3 ? *coef : 0.f;
    const int CG = F / VEC, RL = 256 / CG;         // column groups (16 B each), row lanes
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + (int64_t)rows_per_block, n_rows);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    int nz_count = 0;
#ifndef GCN_CS_UNROLL   // measured on MI355X: unroll 2 / 4 and 8192 slabs change nothing (+-1 %): the
#define GCN_CS_UNROLL 1 // pass already runs at the device's mixed read+write rate, 5.5 TB/s
#endif
#pragma unroll GCN_CS_UNROLL
    for (int64_t r = r0 + rl; r < r1; r += RL) {
        const int64_t off = r * F + VEC * cg;
        float g[VEC];
        if (MODE != 3) Elem<T, VEC>::unpack(*(const Raw *)(grad_out + off), g);
        Raw packed = {};
        if (MODE == 3) {
            const int64_t t_row = target[r];          // (one address per row: a broadcast load)
            const int64_t t = t_row - VEC * cg;
#pragma unroll
            for (int i = 0; i < VEC; ++i) g[i] = 0.f;
            if (t_row >= 0) {   // a negative label (torch's ignore_index = -100) is an ignored row: zeros
                float o[VEC];
                Elem<T, VEC>::unpack(*(const Raw *)(out + off), o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) g[i] = cf * ((t == i ? 1.f : 0.f) - expf(o[i]));
            }
            packed = Elem<T, VEC>::pack(g);
            Elem<T, VEC>::unpack(packed, g);
        }
        if (MODE == 1) {
            // lanes whose 16 bytes of grad_out are all zero produce zeros whatever `out` holds:
            // they skip its load (row-sparse gradients: most of the `out` traffic disappears)
            bool gnz = false;
#pragma unroll
            for (int i = 0; i < VEC; ++i) gnz |= (g[i] != 0.f);
            if (gnz) {
                float o[VEC];
                Elem<T, VEC>::unpack(*(const Raw *)(out + off), o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) g[i] = o[i] > 0.f ? g[i] * scale : 0.f;
            }
            packed = Elem<T, VEC>::pack(g);
            Elem<T, VEC>::unpack(packed, g);   // sums and flags follow the STORED (rounded) values
        }
        if (MODE == 2) {
            float rs = 0.f;
            bool gnz = false;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                rs += g[i];
                gnz |= (g[i] != 0.f);
            }
            for (int o2 = 1; o2 < CG; o2 <<= 1) rs += __shfl_xor(rs, o2, 64);
            const unsigned long long gb = __ballot(gnz);
            const int ln = threadIdx.x & 63;
            const unsigned long long gm = CG >= 64 ? ~0ull : (((1ull << CG) - 1) << (ln & ~(CG - 1)));
            if (gb & gm) {   // uniform over the row's lanes
                float o[VEC];
                Elem<T, VEC>::unpack(*(const Raw *)(out + off), o);
#pragma unroll
                for (int i = 0; i < VEC; ++i) g[i] -= expf(o[i]) * rs;
            }
            packed = Elem<T, VEC>::pack(g);
            Elem<T, VEC>::unpack(packed, g);
        }
        bool any = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            acc[i] += g[i];
            any |= (g[i] != 0.f);
        }
        bool row_nz = true;
        if (rowflag != nullptr) {
            // a row lives in CG <= 64 consecutive lanes of one wave (all active or all inactive
            // together): its non-zero flag is a slice of the wave's ballot over the active lanes;
            // the row's first lane writes the byte
            const unsigned long long b = __ballot(any);
            const int lane = threadIdx.x & 63;
            const unsigned long long m = CG >= 64 ? ~0ull : (((1ull << CG) - 1) << (lane & ~(CG - 1)));
            row_nz = (b & m) != 0;
            if (cg == 0) {
                rowflag[r] = row_nz ? 1 : 0;
                nz_count += row_nz ? 1 : 0;
            }
        }
        // skip mode: all-zero rows of the result are not written (the consumer reads the flagged
        // rows only) — at the labelled share of the bench that is 84-95 % of the pass's writes
        if (MODE != 0 && (!skip || row_nz)) *(Raw *)(grad_pre + off) = packed;
    }
    if (rowflag != nullptr) {
        // one atomic per wave: lanes that own rows hold their counts
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nz_count += __shfl_xor(nz_count, off, 64);
        if ((threadIdx.x & 63) == 0 && nz_count) atomicAdd(nnz_rows, nz_count);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] = acc[i];
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RL; ++k)   // fixed order
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += red[(k * CG + cg) * VEC + i];
#pragma unroll
        for (int i = 0; i < VEC; ++i) partial[(int64_t)blockIdx.x * F + VEC * cg + i] = acc[i];
    }
}

// per-row bytes -> bitmap words (1 bit per row; fits the XCD L2 where the byte table does not)
__global__ __launch_bounds__(256) void pack_row_flags_kernel(const uint8_t *__restrict__ bytes,
                                                             uint32_t *__restrict__ bits, int64_t n_rows)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_words = (n_rows + 31) >> 5;
    if (w >= n_words) return;
    uint32_t v = 0;
    const int64_t r0 = w << 5;
#pragma unroll 8
    for (int i = 0; i < 32; ++i)
        if (r0 + i < n_rows && bytes[r0 + i]) v |= 1u << i;
    bits[w] = v;
}

// partial[n_blocks][F] -> colsum[F].  A 1024-thread block owns 32 columns; thread (g, c) adds the
// partial rows g, g+32, g+64, ... of column c (4 loads in flight), then the 32 group sums are added
// in group order through LDS: a fixed summation order, so the result is bitwise reproducible.
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float *__restrict__ partial,
                                                             float *__restrict__ colsum, int n_blocks,
                                                             int F)
{
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int f = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (f < F) {
        int b = g;
        for (; b + 96 < n_blocks; b += 128) {
            s0 += partial[(int64_t)b * F + f];
            s1 += partial[(int64_t)(b + 32) * F + f];
            s2 += partial[(int64_t)(b + 64) * F + f];
            s3 += partial[(int64_t)(b + 96) * F + f];
        }
        for (; b < n_blocks; b += 32) s0 += partial[(int64_t)b * F + f];
    }
    red[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && f < F) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += red[k][c];
        colsum[f] = s;
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR>
void launch_narrow(const KParams &kp, bool is64, dim3 grid, hipStream_t s)
{
    const dim3 block(kWave * kWavesPerBlock);
    const int xepi = (kp.cflag != nullptr || kp.cabsmax != nullptr) ? 2 : (kp.log_softmax ? 1 : 0);   // extras
#define GCN_LAUNCH_NARROW(I, X) \
    hipLaunchKernelGGL((spmm_narrow_kernel<T, VEC, LPR, I, X>), grid, block, 0, s, kp)
    if (xepi == 2) {
        if (is64) GCN_LAUNCH_NARROW(int64_t, 2);
        else GCN_LAUNCH_NARROW(int32_t, 2);
    } else if (xepi == 1) {
        if (is64) GCN_LAUNCH_NARROW(int64_t, 1);
        else GCN_LAUNCH_NARROW(int32_t, 1);
    } else {
        if (is64) GCN_LAUNCH_NARROW(int64_t, 0);
        else GCN_LAUNCH_NARROW(int32_t, 0);
    }
#undef GCN_LAUNCH_NARROW
}

template <typename T, int VEC>
void launch_narrow_lpr(const KParams &kp, bool is64, int lpr, dim3 grid, hipStream_t s)
{
    switch (lpr) {
    case 1: launch_narrow<T, VEC, 1>(kp, is64, grid, s); break;
    case 2: launch_narrow<T, VEC, 2>(kp, is64, grid, s); break;
    case 4: launch_narrow<T, VEC, 4>(kp, is64, grid, s); break;
    case 8: launch_narrow<T, VEC, 8>(kp, is64, grid, s); break;
    case 16: launch_narrow<T, VEC, 16>(kp, is64, grid, s); break;
    case 32: launch_narrow<T, VEC, 32>(kp, is64, grid, s); break;
    default: launch_narrow<T, VEC, 64>(kp, is64, grid, s); break;
    }
}

template <typename T, int VEC>
void launch_wide(const KParams &kp, bool is64, dim3 grid, hipStream_t s)
{
    constexpr int D = 8;
    const dim3 block(kWave * kWavesPerBlock);
    const bool xepi = kp.log_softmax || kp.cflag != nullptr || kp.cabsmax != nullptr;
#define GCN_LAUNCH_WIDE(I, FL, X) \
    hipLaunchKernelGGL((spmm_wide_kernel<T, VEC, I, D, FL, X>), grid, block, 0, s, kp)
    if (kp.bflag != nullptr || kp.csel != nullptr) {   // operand hint / row selection: the flag variant
        if (xepi) {
            if (is64) GCN_LAUNCH_WIDE(int64_t, true, 2);
            else GCN_LAUNCH_WIDE(int32_t, true, 2);
        } else {
            if (is64) GCN_LAUNCH_WIDE(int64_t, true, 0);
            else GCN_LAUNCH_WIDE(int32_t, true, 0);
        }
    } else {
        if (kp.log_softmax && kp.cflag == nullptr && kp.cabsmax == nullptr) {
            // the last layer of every forward pass: log_softmax only, without the row-flag / maximum
            // code of the full-extras instantiation
            if (is64) GCN_LAUNCH_WIDE(int64_t, false, 1);
            else GCN_LAUNCH_WIDE(int32_t, false, 1);
        } else if (xepi) {
            if (is64) GCN_LAUNCH_WIDE(int64_t, false, 2);
            else GCN_LAUNCH_WIDE(int32_t, false, 2);
        } else {
            if (is64) GCN_LAUNCH_WIDE(int64_t, false, 0);
            else GCN_LAUNCH_WIDE(int32_t, false, 0);
        }
    }
#undef GCN_LAUNCH_WIDE
}

int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

template <typename T, int VECW>
int spmm_typed(const gcn_csr_plan *plan, KParams &kp, hipStream_t s)
{
    const int F = kp.F;
    const bool is64 = plan->rowptr_is64 != 0;
    const bool vec_ok = (F % VECW == 0) && (((uintptr_t)kp.B) % 16 == 0) &&
                        (((uintptr_t)kp.B2) % 16 == 0) &&
                        ((kp.ldb2 * (int64_t)sizeof(T)) % 16 == 0) &&
                        (((uintptr_t)kp.C) % 16 == 0) &&
                        ((kp.ldb * (int64_t)sizeof(T)) % 16 == 0) &&
                        ((kp.ldc * (int64_t)sizeof(T)) % 16 == 0);
    const unsigned nblk = (unsigned)((kp.n_total + kWavesPerBlock - 1) / kWavesPerBlock);
    // skipping all-zero rows is a per-store decision: only when one store call holds the whole row
    if (kp.cskip && !(F <= kWave || (vec_ok && F / VECW <= kWave))) kp.cskip = 0;
    if (kp.log_softmax && !(F <= kWave || (vec_ok && F / VECW <= kWave)))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: log_softmax needs the row inside one wavefront "
                                  "(F <= 64, or 16-byte aligned operands with F/lane width <= 64)");
    if (kp.n_total > 0) {
        if (vec_ok) {
            const int lanes = F / VECW;
            if (lanes > 32) {
                dim3 grid(nblk, (unsigned)((lanes + kWave - 1) / kWave));
                launch_wide<T, VECW>(kp, is64, grid, s);
            } else {
                dim3 grid(nblk, 1);
                launch_narrow_lpr<T, VECW>(kp, is64, next_pow2(lanes), grid, s);
            }
        } else {
            const int lpr = std::min(kWave, next_pow2(F));
            dim3 grid(nblk, (unsigned)((F + lpr - 1) / lpr));
            launch_narrow_lpr<T, 1>(kp, is64, lpr, grid, s);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "spmm launch");
    }
    if (kp.n_long > 0) {
        hipLaunchKernelGGL((spmm_long_reduce_kernel<T>), dim3((unsigned)((kp.n_long + 3) / 4)), dim3(256), 0,
                           s, kp);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "long-row reduce launch");
    }
    return 0;
}

template <typename IdxT>
int plan_walk(const IdxT *rp, int64_t n_rows, int item_cost, int long_thresh, int32_t *items,
              int64_t cap_items, int32_t *chunk_row, int64_t *chunk_e0, int64_t cap_chunks,
              int32_t *long_row, int32_t *long_chunk0, int64_t cap_long, int64_t *n_items,
              int64_t *n_chunks, int64_t *n_long)
{
    const bool fill = items != nullptr;
    int64_t ni = 0, nc = 0, nl = 0;
    int64_t ra = 0, cost = 0;
    auto close = [&](int64_t rb) -> bool {
        if (rb > ra) {
            if (fill) {
                if (ni >= cap_items) return false;
                items[2 * ni] = (int32_t)ra;
                items[2 * ni + 1] = (int32_t)rb;
            }
            ++ni;
        }
        ra = rb;
        cost = 0;
        return true;
    };
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t deg = (int64_t)rp[r + 1] - (int64_t)rp[r];
        if (deg < 0) return GCN_E_BADARG;
        if (deg > long_thresh) {
            if (!close(r)) return GCN_E_CAPACITY;
            ra = r + 1;   // the long row belongs to no row-batch item
            const int64_t k = (deg + long_thresh - 1) / long_thresh;
            if (fill) {
                if (nl >= cap_long || nc + k > cap_chunks) return GCN_E_CAPACITY;
                long_row[nl] = (int32_t)r;
                long_chunk0[nl] = (int32_t)nc;
                for (int64_t c = 0; c < k; ++c) {
                    chunk_row[nc + c] = (int32_t)r;
                    chunk_e0[nc + c] = (int64_t)rp[r] + c * long_thresh;
                }
            }
            nc += k;
            ++nl;
            continue;
        }
        const int64_t c = deg + 1;
        if (r > ra && (cost + c > item_cost || r - ra >= kWave)) {
            if (!close(r)) return GCN_E_CAPACITY;
        }
        cost += c;
    }
    if (!close(n_rows)) return GCN_E_CAPACITY;
    if (fill && long_chunk0 != nullptr) {
        if (nl > cap_long) return GCN_E_CAPACITY;
        long_chunk0[nl] = (int32_t)nc;
    }
    if (n_items) *n_items = ni;
    if (n_chunks) *n_chunks = nc;
    if (n_long) *n_long = nl;
    return 0;
}


// ------------------------------------------------------------------------------------------
// SDDMM: out[e] = < G[row(e), :], B[col[e], :] > for every stored entry e of the CSR pattern — the
// gradient of the adjacency VALUES when a caller sets adj.requires_grad (PyTorch's `mm` derivative
// for a sparse first operand: grad_self = (grad · mat2ᵀ) sampled on self's pattern,
// derivatives.yaml `mm` / pygcn/layers.py:34).  The reference never needs it (adj is a loaded
// constant, pygcn/train.py:80,123); SURVEY row f4 lists it as optional.
// Same work units as the product (row-batch items and long-row chunks, one per wave); a wave keeps
// the row of G it works on in registers where the row fits one wave instruction, gathers U = 4
// rows of B at a time (16 B per lane, row address wave-uniform) and reduces the 4 partial dot
// products across the wave.  Gather-bound like the product: nnz·(F·s) + n·(F·s) bytes.
// ------------------------------------------------------------------------------------------
struct SddmmParams {
    const void *rowptr;
    const int32_t *col;
    const int32_t *items;
    const int32_t *chunk_row;
    const int64_t *chunk_e0;
    const void *G;
    const void *B;
    float *out;
    int64_t ldg, ldb;      // elements
    int32_t F, n_total, n_chunks, long_thresh;
};

template <typename T, int VEC, typename IdxT>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void sddmm_kernel(SddmmParams p)
{
    typedef typename Elem<T, VEC>::Raw Raw;
    constexpr int U = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int item =
        blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= p.n_total) return;
    const IdxT *__restrict__ rp = (const IdxT *)p.rowptr;
    int ra, rb;
    int64_t e_lo = -1, e_hi = -1;                   // entry window of a chunk (whole rows otherwise)
    if (item < p.n_chunks) {
        ra = p.chunk_row[item];
        rb = ra + 1;
        e_lo = p.chunk_e0[item];
        e_hi = min(e_lo + (int64_t)p.long_thresh, (int64_t)rp[ra + 1]);
    } else {
        const int it = item - p.n_chunks;
        ra = p.items[2 * it];
        rb = p.items[2 * it + 1];
    }
    const int slab = kWave * VEC;                   // elements one wave instruction covers
    for (int r = ra; r < rb; ++r) {                 // (wave-uniform loop)
        const int64_t es = e_lo >= 0 ? e_lo : (int64_t)rp[r];
        const int64_t ee = e_lo >= 0 ? e_hi : (int64_t)rp[r + 1];
        if (es >= ee) continue;
        const T *grow = (const T *)p.G + (int64_t)r * p.ldg;
        for (int64_t e = es; e < ee; e += U) {
            const int cnt = (int)min((int64_t)U, ee - e);
            float part[U];
#pragma unroll
            for (int u = 0; u < U; ++u) part[u] = 0.f;
            for (int f0 = lane * VEC; f0 < p.F; f0 += slab) {
                float g[VEC];
                Elem<T, VEC>::unpack(*(const Raw *)(grow + f0), g);      // (L1-resident across the row)
                Raw braw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = p.col[e + (u < cnt ? u : 0)];          // (uniform: scalar load)
                    braw[u] = *(const Raw *)((const T *)p.B + (int64_t)c * p.ldb + f0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float b[VEC];
                    Elem<T, VEC>::unpack(braw[u], b);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) part[u] = fmaf(g[i], b[i], part[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float s = group_sum<kWave>(part[u]);
                if (lane == u && u < cnt) p.out[e + u] = s;
            }
        }
    }
}

}   // namespace

// shared with gcn_ingest.hip
int gcn_internal_fail(int code, const char *msg) { return fail(code, msg); }
int gcn_internal_fail_hip(int hip_error, const char *where)
{
    return fail_hip((hipError_t)hip_error, where);
}

extern "C" {

int gcn_abi_version(void) { return GCN_ABI_VERSION; }

const char *gcn_last_error(void) { return g_err; }

int gcn_plan_count_host(const void *rowptr_host, int rowptr_is64, int64_t n_rows,
                        int32_t item_cost, int32_t long_thresh, int64_t *n_items,
                        int64_t *n_chunks, int64_t *n_long)
{
    if (rowptr_host == nullptr || n_rows < 0 || n_rows >= INT32_MAX)
        return fail(GCN_E_BADARG, "gcn_plan_count_host: bad rowptr / n_rows");
    if (item_cost <= 0) item_cost = kDefaultItemCost;
    if (long_thresh <= 0) long_thresh = kDefaultLongThresh;
    int rc = rowptr_is64
                 ? plan_walk((const int64_t *)rowptr_host, n_rows, item_cost, long_thresh, nullptr,
                             0, nullptr, nullptr, 0, nullptr, nullptr, 0, n_items, n_chunks, n_long)
                 : plan_walk((const int32_t *)rowptr_host, n_rows, item_cost, long_thresh, nullptr,
                             0, nullptr, nullptr, 0, nullptr, nullptr, 0, n_items, n_chunks,
                             n_long);
    return rc ? fail(rc, "gcn_plan_count_host: rowptr is not monotone") : 0;
}

int gcn_plan_fill_host(const void *rowptr_host, int rowptr_is64, int64_t n_rows,
                       int32_t item_cost, int32_t long_thresh, int32_t *items, int64_t n_items,
                       int32_t *chunk_row, int64_t *chunk_e0, int64_t n_chunks,
                       int32_t *long_row, int32_t *long_chunk0, int64_t n_long)
{
    if (rowptr_host == nullptr || n_rows < 0 || n_rows >= INT32_MAX || items == nullptr ||
        long_chunk0 == nullptr)
        return fail(GCN_E_BADARG, "gcn_plan_fill_host: null output or bad n_rows");
    if (item_cost <= 0) item_cost = kDefaultItemCost;
    if (long_thresh <= 0) long_thresh = kDefaultLongThresh;
    int64_t ni = 0, nc = 0, nl = 0;
    int rc = rowptr_is64 ? plan_walk((const int64_t *)rowptr_host, n_rows, item_cost, long_thresh,
                                     items, n_items, chunk_row, chunk_e0, n_chunks, long_row,
                                     long_chunk0, n_long, &ni, &nc, &nl)
                         : plan_walk((const int32_t *)rowptr_host, n_rows, item_cost, long_thresh,
                                     items, n_items, chunk_row, chunk_e0, n_chunks, long_row,
                                     long_chunk0, n_long, &ni, &nc, &nl);
    if (rc) return fail(rc, "gcn_plan_fill_host: output arrays too small or rowptr not monotone");
    if (ni != n_items || nc != n_chunks || nl != n_long)
        return fail(GCN_E_CAPACITY, "gcn_plan_fill_host: sizes differ from gcn_plan_count_host");
    return 0;
}

size_t gcn_spmm_workspace_bytes(const gcn_csr_plan *plan, int64_t F)
{
    if (plan == nullptr || F <= 0 || plan->n_chunks <= 0) return 0;
    return (size_t)plan->n_chunks * (size_t)F * sizeof(float);
}

int gcn_spmm_csr_ep(const gcn_csr_plan *plan, int dtype, const void *B, int64_t ldb, void *C,
                    int64_t ldc, int64_t F, const gcn_epilogue *ep, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    const float *bias = ep ? ep->bias : nullptr;
    const int relu = ep ? ep->relu : 0;
    const float drop_p = ep ? ep->dropout_p : 0.f;
    if (!(drop_p >= 0.f && drop_p < 1.f))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: dropout_p must be in [0, 1)");
    if (plan == nullptr) return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: plan is NULL");
    if (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16)
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: unknown dtype");
    if (plan->n_rows < 0 || plan->n_cols < 0 || plan->nnz < 0 || F < 0 || F > INT32_MAX)
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: negative size");
    if (plan->n_rows == 0 || F == 0) return 0;
    if (C == nullptr || (B == nullptr && plan->nnz > 0 && !(ep && ep->b2 && ep->b_split == 0)))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: B or C is NULL");
    if (ldb < F || ldc < F) return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: ldb/ldc smaller than F");
    if (ldb * 4 >= ((int64_t)1 << 32))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: row stride of B must be below 4 GiB");
    if (plan->rowptr == nullptr || plan->items == nullptr ||
        (plan->nnz > 0 && (plan->col == nullptr || plan->val == nullptr)))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: plan has NULL arrays");
    if (plan->n_chunks > 0 && (plan->chunk_row == nullptr || plan->chunk_e0 == nullptr ||
                               plan->long_row == nullptr || plan->long_chunk0 == nullptr))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: plan has long rows but NULL chunk arrays");
    if (plan->n_items + plan->n_chunks >= INT32_MAX || plan->n_long >= INT32_MAX)
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: schedule too large");
    const size_t need = gcn_spmm_workspace_bytes(plan, F);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need))
        return fail(GCN_E_WORKSPACE, "gcn_spmm_csr_ep: workspace too small");

    KParams kp;
    kp.rowptr = plan->rowptr;
    kp.col = plan->col;
    kp.val = plan->val;
    kp.items = plan->items;
    kp.chunk_row = plan->chunk_row;
    kp.chunk_e0 = plan->chunk_e0;
    kp.long_row = plan->long_row;
    kp.long_chunk0 = plan->long_chunk0;
    kp.B = B;
    kp.C = C;
    kp.bias = bias;
    kp.partial = (float *)workspace;
    kp.ldb = ldb;
    kp.ldc = ldc;
    kp.F = (int32_t)F;
    kp.n_total = (int32_t)(plan->n_items + plan->n_chunks);
    kp.n_chunks = (int32_t)plan->n_chunks;
    kp.n_long = (int32_t)plan->n_long;
    kp.long_thresh = plan->long_thresh > 0 ? plan->long_thresh : kDefaultLongThresh;
    kp.relu = relu ? 1 : 0;
    // keep iff rand16 >= round(p * 2^16)  (p = 0 -> threshold 0 -> dropout off)
    kp.drop_thresh = gcn_dropout_threshold16(drop_p);
    kp.drop_scale = gcn_dropout_scale16(kp.drop_thresh);
    kp.drop_row_base = ep ? ep->drop_row_base : 0;
    kp.seed_lo = ep ? (uint32_t)ep->seed : 0u;
    kp.seed_hi = ep ? (uint32_t)(ep->seed >> 32) : 0u;
    kp.seed_dev = ep ? ep->seed_dev : nullptr;
    kp.csel = ep ? ep->c_row_select : nullptr;
    kp.cabsmax = ep ? (uint32_t *)ep->c_absmax : nullptr;
    kp.cskip = (ep && ep->c_skip_zero_rows && ep->c_row_nonzero) ? 1 : 0;
    kp.B2 = ep ? ep->b2 : nullptr;
    kp.ldb2 = ep && ep->b2 ? ep->ldb2 : 0;
    kp.b_split = (ep && ep->b2) ? (int32_t)std::min<int64_t>(ep->b_split, INT32_MAX) : INT32_MAX;
    if (ep && ep->b2 && (ep->b_split < 0 || ep->ldb2 < F || ep->ldb2 * 4 >= ((int64_t)1 << 32)))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: bad second block of B");
    kp.bflag = ep ? ep->b_row_nonzero : nullptr;
    kp.bnnz = ep ? ep->b_nnz_rows : nullptr;
    if (kp.bflag == nullptr || kp.bnnz == nullptr) kp.bflag = nullptr, kp.bnnz = nullptr;
    kp.cflag = ep ? ep->c_row_nonzero : nullptr;
    kp.log_softmax = (ep && ep->log_softmax) ? 1 : 0;
    if (kp.log_softmax && (relu || drop_p > 0.f))
        return fail(GCN_E_BADARG, "gcn_spmm_csr_ep: log_softmax cannot be combined with relu / dropout");
    kp.n_cols = (int32_t)std::min<int64_t>(plan->n_cols, INT32_MAX);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCN_DTYPE_F32) return spmm_typed<float, 4>(plan, kp, s);
    return spmm_typed<bf16_t, 8>(plan, kp, s);
}

int gcn_spmm_csr(const gcn_csr_plan *plan, int dtype, const void *B, int64_t ldb, void *C,
                 int64_t ldc, int64_t F, const float *bias, int relu, void *workspace,
                 size_t workspace_bytes, void *stream)
{
    gcn_epilogue ep = {};          // (every option off; fields added later stay zero)
    ep.bias = bias;
    ep.relu = relu;
    return gcn_spmm_csr_ep(plan, dtype, B, ldb, C, ldc, F, &ep, workspace, workspace_bytes, stream);
}

int gcn_relu_dropout_backward(int dtype, const void *grad_out, const void *out, void *grad_pre,
                              int64_t n_elems, float scale, void *stream)
{
    if (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16)
        return fail(GCN_E_BADARG, "gcn_relu_dropout_backward: unknown dtype");
    if (n_elems < 0) return fail(GCN_E_BADARG, "gcn_relu_dropout_backward: negative size");
    if (n_elems == 0) return 0;
    if (grad_out == nullptr || out == nullptr || grad_pre == nullptr)
        return fail(GCN_E_BADARG, "gcn_relu_dropout_backward: NULL pointer");
    const int64_t per = dtype == GCN_DTYPE_F32 ? 4 : 8;           // elements per 16-byte lane
    const bool vec = (n_elems % per == 0) && (((uintptr_t)grad_out | (uintptr_t)out |
                                                (uintptr_t)grad_pre) % 16 == 0);
    const int64_t n_units = vec ? n_elems / per : n_elems;
    const unsigned blocks = (unsigned)std::min<int64_t>((n_units + 255) / 256, 256 * 8 * 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GCN_DTYPE_F32) {
        if (vec)
            hipLaunchKernelGGL((relu_dropout_bwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, s,
                               (const float *)grad_out, (const float *)out, (float *)grad_pre,
                               n_units, scale);
        else
            hipLaunchKernelGGL((relu_dropout_bwd_kernel<float, 1>), dim3(blocks), dim3(256), 0, s,
                               (const float *)grad_out, (const float *)out, (float *)grad_pre,
                               n_units, scale);
    } else {
        if (vec)
            hipLaunchKernelGGL((relu_dropout_bwd_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 0, s,
                               (const bf16_t *)grad_out, (const bf16_t *)out, (bf16_t *)grad_pre,
                               n_units, scale);
        else
            hipLaunchKernelGGL((relu_dropout_bwd_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, s,
                               (const bf16_t *)grad_out, (const bf16_t *)out, (bf16_t *)grad_pre,
                               n_units, scale);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "relu_dropout_backward launch");
    return 0;
}

#ifndef GCN_CS_BLOCKS
#define GCN_CS_BLOCKS 2048
#endif
static constexpr int64_t kColsumBlocks = GCN_CS_BLOCKS;   // slabs of rows = partial column sums

static bool colsum_shape_ok(int64_t F, int dtype)
{
    const int64_t v = dtype == GCN_DTYPE_BF16 ? 8 : 4;   // elements per 16-byte lane
    return F >= v && F <= 256 * v && (F % v) == 0 && (256 % (F / v)) == 0;
}

size_t gcn_bwd_colsum_workspace_bytes(int64_t n_rows, int64_t F, int dtype)
{
    if (n_rows <= 0 || (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16) ||
        !colsum_shape_ok(F, dtype))
        return 0;
    const int64_t blocks = std::min<int64_t>((n_rows + 63) / 64, kColsumBlocks);
    // partial column sums + one byte per row (staging for the row bitmap)
    return (size_t)blocks * (size_t)F * sizeof(float) + (((size_t)n_rows + 15) & ~(size_t)15);
}

static int bwd_colsum_impl(const char *who, int mode, int dtype, const void *grad_out, const void *out,
                           void *grad_pre, float *colsum, int64_t n_rows, int64_t F, float scale,
                           uint32_t *row_bits, int32_t *nnz_rows, int skip_zero_rows,
                           void *workspace, size_t workspace_bytes, void *stream,
                           const int64_t *target = nullptr, const float *coef = nullptr)
{
    char msg[160];
    auto bad = [&](int code, const char *what) {
        std::snprintf(msg, sizeof msg, "%s: %s", who, what);
        return fail(code, msg);
    };
    if (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16) return bad(GCN_E_BADARG, "unknown dtype");
    if ((row_bits == nullptr) != (nnz_rows == nullptr))
        return bad(GCN_E_BADARG, "row_bits and nnz_rows go together");
    const int64_t vec = dtype == GCN_DTYPE_BF16 ? 8 : 4;
    if (n_rows < 0 || !colsum_shape_ok(F, dtype))
        return bad(GCN_E_BADARG, "F must be a multiple of the 16-byte lane width with F/width dividing 256");
    if (mode >= 2 && F / vec > 64)
        return bad(GCN_E_BADARG, "a row must fit one wavefront (F / lane width <= 64)");
    if (row_bits != nullptr && F / vec > 64)   // a row spans several wavefronts: no per-row ballot
        return bad(GCN_E_BADARG, "row_bits / nnz_rows need F / lane width <= 64 (pass NULL for wider rows)");
    if (skip_zero_rows && (row_bits == nullptr || mode == 0))
        return bad(GCN_E_BADARG, "skip_zero_rows needs the row bitmap outputs and a result tensor");
    if (colsum == nullptr || (grad_out == nullptr && mode != 3) || (out != nullptr && grad_pre == nullptr)
        || (mode == 3 && (target == nullptr || coef == nullptr || out == nullptr)))
        return bad(GCN_E_BADARG, "NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (nnz_rows != nullptr) {
        hipError_t e = hipMemsetAsync(nnz_rows, 0, sizeof(int32_t), s);
        if (e != hipSuccess) return fail_hip(e, who);
    }
    if (n_rows == 0) {
        hipError_t e = hipMemsetAsync(colsum, 0, (size_t)F * sizeof(float), s);
        return e == hipSuccess ? 0 : fail_hip(e, who);
    }
    const size_t need = gcn_bwd_colsum_workspace_bytes(n_rows, F, dtype);
    if (workspace == nullptr || workspace_bytes < need) return bad(GCN_E_WORKSPACE, "workspace too small");
    if (((uintptr_t)grad_out | (uintptr_t)out | (uintptr_t)grad_pre | (uintptr_t)workspace) % 16 != 0)
        return bad(GCN_E_ALIGN, "16-byte alignment required");
    const int64_t blocks = std::min<int64_t>((n_rows + 63) / 64, kColsumBlocks);
    const int rows_per_block = (int)((n_rows + blocks - 1) / blocks);
    uint8_t *row_nonzero = row_bits ? (uint8_t *)workspace + (size_t)blocks * (size_t)F * sizeof(float)
                                    : nullptr;
    const dim3 grid((unsigned)blocks), block(256);
    float *part = (float *)workspace;
#define GCN_LAUNCH_COLSUM(T, V, M)                                                                   \
    hipLaunchKernelGGL((bwd_colsum_kernel<T, V, M>), grid, block, 0, s, (const T *)grad_out,         \
                       (const T *)out, (T *)grad_pre, part, n_rows, (int)F, scale, rows_per_block,   \
                       row_nonzero, nnz_rows, skip_zero_rows ? 1 : 0, target, coef)
    if (dtype == GCN_DTYPE_F32) {
        if (mode == 3) GCN_LAUNCH_COLSUM(float, 4, 3);
        else if (mode == 2) GCN_LAUNCH_COLSUM(float, 4, 2);
        else if (mode == 1) GCN_LAUNCH_COLSUM(float, 4, 1);
        else GCN_LAUNCH_COLSUM(float, 4, 0);
    } else {
        if (mode == 3) GCN_LAUNCH_COLSUM(bf16_t, 8, 3);
        else if (mode == 2) GCN_LAUNCH_COLSUM(bf16_t, 8, 2);
        else if (mode == 1) GCN_LAUNCH_COLSUM(bf16_t, 8, 1);
        else GCN_LAUNCH_COLSUM(bf16_t, 8, 0);
    }
#undef GCN_LAUNCH_COLSUM
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((F + 31) / 32)), dim3(1024), 0, s,
                       (const float *)workspace, colsum, (int)blocks, (int)F);
    if (row_bits != nullptr)
        hipLaunchKernelGGL(pack_row_flags_kernel, dim3((unsigned)((((n_rows + 31) >> 5) + 255) / 256)),
                           dim3(256), 0, s, (const uint8_t *)row_nonzero, row_bits, n_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, who);
    return 0;
}

int gcn_relu_dropout_backward_colsum(int dtype, const void *grad_out, const void *out, void *grad_pre,
                                     float *colsum, int64_t n_rows, int64_t F, float scale,
                                     uint32_t *row_bits, int32_t *nnz_rows, int skip_zero_rows,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    return bwd_colsum_impl("gcn_relu_dropout_backward_colsum", out != nullptr ? 1 : 0, dtype, grad_out,
                           out, grad_pre, colsum, n_rows, F, scale, row_bits, nnz_rows, skip_zero_rows,
                           workspace, workspace_bytes, stream);
}

int gcn_log_softmax_backward_colsum(int dtype, const void *grad_out, const void *out, void *grad_pre,
                                    float *colsum, int64_t n_rows, int64_t F, uint32_t *row_bits,
                                    int32_t *nnz_rows, int skip_zero_rows, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (out == nullptr)
        return fail(GCN_E_BADARG, "gcn_log_softmax_backward_colsum: NULL pointer");
    return bwd_colsum_impl("gcn_log_softmax_backward_colsum", 2, dtype, grad_out, out, grad_pre, colsum,
                           n_rows, F, 1.f, row_bits, nnz_rows, skip_zero_rows, workspace,
                           workspace_bytes, stream);
}

int gcn_nll_log_softmax_backward_colsum(int dtype, const int64_t *target, const float *coef,
                                        const void *out, void *grad_pre, float *colsum, int64_t n_rows,
                                        int64_t F, void *workspace, size_t workspace_bytes, void *stream)
{
    return bwd_colsum_impl("gcn_nll_log_softmax_backward_colsum", 3, dtype, nullptr, out, grad_pre,
                           colsum, n_rows, F, 1.f, nullptr, nullptr, 0, workspace, workspace_bytes,
                           stream, target, coef);
}

int gcn_sddmm_csr(const gcn_csr_plan *plan, int dtype, const void *G, int64_t ldg, const void *B,
                  int64_t ldb, int64_t F, float *out_vals, void *stream)
{
    if (plan == nullptr) return fail(GCN_E_BADARG, "gcn_sddmm_csr: plan is NULL");
    if (dtype != GCN_DTYPE_F32 && dtype != GCN_DTYPE_BF16)
        return fail(GCN_E_BADARG, "gcn_sddmm_csr: unknown dtype");
    if (plan->n_rows < 0 || plan->n_cols < 0 || plan->nnz < 0 || F < 0 || F > INT32_MAX)
        return fail(GCN_E_BADARG, "gcn_sddmm_csr: negative size");
    if (plan->nnz == 0) return 0;
    if (G == nullptr || B == nullptr || out_vals == nullptr || plan->rowptr == nullptr ||
        plan->col == nullptr || plan->items == nullptr)
        return fail(GCN_E_BADARG, "gcn_sddmm_csr: NULL pointer");
    if (plan->n_chunks > 0 && (plan->chunk_row == nullptr || plan->chunk_e0 == nullptr))
        return fail(GCN_E_BADARG, "gcn_sddmm_csr: plan has long rows but NULL chunk arrays");
    if (ldg < F || ldb < F) return fail(GCN_E_BADARG, "gcn_sddmm_csr: ldg / ldb smaller than F");
    hipStream_t s = (hipStream_t)stream;
    if (F == 0) {
        hipError_t e = hipMemsetAsync(out_vals, 0, (size_t)plan->nnz * sizeof(float), s);
        return e == hipSuccess ? 0 : fail_hip(e, "gcn_sddmm_csr");
    }
    SddmmParams sp;
    sp.rowptr = plan->rowptr;
    sp.col = plan->col;
    sp.items = plan->items;
    sp.chunk_row = plan->chunk_row;
    sp.chunk_e0 = plan->chunk_e0;
    sp.G = G;
    sp.B = B;
    sp.out = out_vals;
    sp.ldg = ldg;
    sp.ldb = ldb;
    sp.F = (int32_t)F;
    sp.n_total = (int32_t)(plan->n_items + plan->n_chunks);
    sp.n_chunks = (int32_t)plan->n_chunks;
    sp.long_thresh = plan->long_thresh > 0 ? plan->long_thresh : kDefaultLongThresh;
    const bool is64 = plan->rowptr_is64 != 0;
    const size_t es = dtype == GCN_DTYPE_BF16 ? 2 : 4;
    const int64_t vw = dtype == GCN_DTYPE_BF16 ? 8 : 4;
    const bool vec_ok = (F % vw == 0) && (((uintptr_t)G | (uintptr_t)B) % 16 == 0) &&
                        ((ldg * (int64_t)es) % 16 == 0) && ((ldb * (int64_t)es) % 16 == 0);
    const dim3 grid((unsigned)((sp.n_total + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWave * kWavesPerBlock);
#define GCN_LAUNCH_SDDMM(T, V)                                                             \
    do {                                                                                   \
        if (is64) hipLaunchKernelGGL((sddmm_kernel<T, V, int64_t>), grid, block, 0, s, sp); \
        else hipLaunchKernelGGL((sddmm_kernel<T, V, int32_t>), grid, block, 0, s, sp);      \
    } while (0)
    if (dtype == GCN_DTYPE_F32) {
        if (vec_ok) GCN_LAUNCH_SDDMM(float, 4);
        else GCN_LAUNCH_SDDMM(float, 1);
    } else {
        if (vec_ok) GCN_LAUNCH_SDDMM(bf16_t, 8);
        else GCN_LAUNCH_SDDMM(bf16_t, 1);
    }
#undef GCN_LAUNCH_SDDMM
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "gcn_sddmm_csr launch");
    return 0;
}

int gcn_csr_transpose_host(const void *rowptr_host, int rowptr_is64, const int32_t *col,
                           const float *val, int64_t n_rows, int64_t n_cols, void *rowptr_t,
                           int32_t *col_t, float *val_t)
{
    if (rowptr_host == nullptr || rowptr_t == nullptr || n_rows < 0 || n_cols < 0)
        return fail(GCN_E_BADARG, "gcn_csr_transpose_host: bad arguments");
    auto run = [&](auto *rp, auto *rpt) -> int {
        typedef typename std::remove_pointer<decltype(rpt)>::type I;
        const int64_t nnz = (int64_t)rp[n_rows];
        if (nnz > 0 && (col == nullptr || val == nullptr || col_t == nullptr || val_t == nullptr))
            return GCN_E_BADARG;
        for (int64_t k = 0; k <= n_cols; ++k) rpt[k] = 0;
        for (int64_t e = 0; e < nnz; ++e) {
            if (col[e] < 0 || col[e] >= n_cols) return GCN_E_BADARG;
            rpt[col[e] + 1]++;
        }
        for (int64_t k = 0; k < n_cols; ++k) rpt[k + 1] += rpt[k];
        for (int64_t i = 0; i < n_rows; ++i) {
            for (int64_t e = (int64_t)rp[i]; e < (int64_t)rp[i + 1]; ++e) {
                const int64_t dst = (int64_t)rpt[col[e]]++;
                col_t[dst] = (int32_t)i;
                val_t[dst] = val[e];
            }
        }
        for (int64_t k = n_cols; k > 0; --k) rpt[k] = rpt[k - 1];
        rpt[0] = (I)0;
        return 0;
    };
    int rc = rowptr_is64 ? run((const int64_t *)rowptr_host, (int64_t *)rowptr_t)
                         : run((const int32_t *)rowptr_host, (int32_t *)rowptr_t);
    return rc ? fail(rc, "gcn_csr_transpose_host: column index out of range or NULL array") : 0;
}

}   // extern "C"
