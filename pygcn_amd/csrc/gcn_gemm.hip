// gcn_gemm.hip — the dense half of GraphConvolution on gfx950 MFMA:
// `support = torch.mm(input, self.weight)` (reference pygcn/layers.py:33) and the two GEMMs of its
// backward, at the layer shapes of configs C3–C5.  Six kernels:
//
//   gemm_xw256_h2_kernel<EPI, SCH>   Y[M,256] = X[M,256]·W[256,256], fp32 in / out.  gfx950 has no
//                          reduced-precision fp32 MFMA and the exact one runs at 1/16 of the 16-bit
//                          rate, so the operands are split into 16-bit parts.  SCH 1 (the DEFAULT
//                          since round 4, fp32-EQUIVALENT): three bf16 parts, six MFMAs per product,
//                          a 24-bit significand, no scaling; SCH 0 (opt-in): both operands scaled
//                          by exact powers of two and split into TWO fp16 parts, x·w ≈ h·h' + h·m' +
//                          m·h', three MFMAs, 22 bits.  Optional row list on the input, the layer's
//                          forward epilogue (bias, ReLU, Philox dropout) or the ReLU / dropout
//                          backward mask in the store, max|Y| as a side result.  Round 3's pipeline:
//                          256-row tiles, X and W by HBM -> LDS DMA, persistent workgroups.
//   gemm_xw256_s16_kernel<EPI>       the three-part product on CONTIGUOUS rows (round 4): 128-row
//                          tiles on 16x16x32 MFMAs, the previous tile's stores, the DMA and the
//                          operand split between the running tile's MFMA groups — what the default
//                          scheme runs wherever no row list is given (6.2 ms against 7.05 at M = 10^7).
//   gemm_xw256_kernel      the three-part product in round 1's form (VGPR-staged), kept as the
//                          bit-exact reference of SCH 1.
//   gemm_atg256_h2_kernel<SCH>  grad_W[256,256] = Σ_r A[ra[r]]ᵀ ⊗ G[rg[r]] over a row LIST (+ ordered
//                          slab reduction), the weight gradient without compacting copies.
//   gemm_bf16_kernel<K,N>  bf16 storage (config C5: 128 -> 128), W resident in LDS, streaming.
//   gemm_atg128_bf16_kernel the weight gradient at bf16 storage (128 x 128).
//
// A rule all of them follow (round 3, DESIGN §3.7 "wait-count audit"): vector loads and stores
// retire IN ORDER on one counter, so a wait on a young load is a wait on everything older.  No load
// is consumed where it is issued inside a store section (bias and seed are staged once per
// workgroup), no data-dependent branch surrounds a memory instruction inside a pipelined loop
// (buffer descriptors / clamped indices instead), prefetch register sets are swapped by unrolling
// rather than copied, and `sched_barrier` pins the issue order where the wait counts depend on it.
//
// Structure of the fp32 product kernels: a 512-thread workgroup owns 256 rows; each of its 8 waves
// owns 32 rows x all 256 columns (8 accumulator tiles of 32x32 = 128 VGPRs, held TRANSPOSED so a
// lane owns an output row and stores 16 bytes at a time).  The K loop (16 steps of 16) is fully
// unrolled.  Per step a wave multiplies its pre-split X fragment with the pre-split W chunk
// (fragment-ordered, conflict-free) that the workgroup staged in LDS: two steps of W per barrier,
// double-buffered; W fragments of column block nb+1 are read while the MFMAs of block nb run; the
// X fragment comes straight from global memory in MFMA layout (8 consecutive floats per lane,
// fetched three steps ahead) and is split BEHIND the previous step's MFMA run (pinned there by an
// opaque asm, or hipcc sinks it back to just after the barrier, where it idles the pipe).  W is
// split and fragment-ordered once per call by a small prep kernel into the workspace.
//
// Measured on MI355X at M = 10^7 (one process, interleaved): three bf16 parts 6.2 ms plain / 6.45 with
// the layer epilogue (gemm_xw256_s16_kernel; round 3's pipeline 7.05, round 1's kernel 7.5), h2 4.8 /
// 4.9 ms (round 2: 5.2 / 6.9), hipBLASLt fp32 9.95 ms.  The three-part kernels are power-managed (same
// cycles per tile with and without their memory streams; the chip holds 1.57 GHz), the h2 kernel is
// bound by the CU's vector-memory instruction issue (3e7 wave-level loads / stores of 1 KiB per
// launch, already 16 B per lane), not by MFMA (48 % busy) or HBM (4.3 TB/s): DESIGN §3.7, with the
// variants that were measured and rejected in §7 (resident-W column split, stores through LDS,
// 4-wave workgroups, row pitch, X staged through LDS in full lines, a deeper X ring).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "gcn_spmm.h"

int gcn_internal_fail(int code, const char *msg);
int gcn_internal_fail_hip(int hip_error, const char *where);

// keep-threshold of the fused dropout: an element is kept iff its 16-bit field >= round(p * 2^16),
// clamped to [1, 65535] for p > 0 (0 = dropout off) — include/gcn_spmm.h, struct gcn_epilogue
static inline uint32_t gcn_dropout_threshold16(float p)
{
    if (!(p > 0.f)) return 0u;
    const double t = (double)p * 65536.0 + 0.5;
    return (uint32_t)std::min(65535.0, std::max(1.0, (double)(int64_t)t));
}
// (the scale of the QUANTISED keep probability, as in gcn_spmm.hip: E[dropout(x)] = x exactly)
static inline float gcn_dropout_scale16(uint32_t thresh)
{
    return thresh == 0u ? 1.f : 65536.f / (float)(65536u - thresh);
}

namespace {

constexpr int kN = 256, kK = 256;
constexpr int kChunk = 16;                         // K per MFMA 32x32x16
constexpr int kChunks = kK / kChunk;               // 16
constexpr int kFragBytes = 64 * 16;                // one B fragment: 64 lanes x 8 bf16
constexpr int kChunkBytes = 3 * 8 * kFragBytes;    // 3 splits x 8 column blocks = 24 KiB
#ifndef GEMM_WAVES
#define GEMM_WAVES 8
#endif
constexpr int kWaves = GEMM_WAVES;                 // waves per workgroup, 32 rows each
constexpr int kThreads = 64 * kWaves;
constexpr int kTileRows = 32 * kWaves;
#ifndef GEMM_STAGE
#define GEMM_STAGE 2
#endif
constexpr int kStage = GEMM_STAGE;                 // K steps of W staged per barrier
constexpr int kStageBytes = kStage * kChunkBytes;
constexpr int kWLoads = kStageBytes / 16 / kThreads;   // 16-byte pieces of a W stage per thread

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_round(float x)   // value of x rounded to bf16, as a float
{
    f32x2 v = {x, 0.f};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    const uint32_t bits = __builtin_bit_cast(uint32_t, b) << 16;
    return __uint_as_float(bits);
}

// x -> (h, m, l) as raw bf16 bit patterns
__device__ __forceinline__ void split3(float x, uint16_t &h, uint16_t &m, uint16_t &l)
{
    const float fh = bf16_round(x);
    const float r1 = x - fh;
    const float fm = bf16_round(r1);
    const float fl = bf16_round(r1 - fm);
    h = (uint16_t)(__float_as_uint(fh) >> 16);
    m = (uint16_t)(__float_as_uint(fm) >> 16);
    l = (uint16_t)(__float_as_uint(fl) >> 16);
}

// W [K=256][N=256] fp32 row-major -> workspace [chunk 16][split 3][colblock 8][lane 64][8 bf16],
// element j of lane l of fragment (chunk, colblock): k = 16*chunk + 8*(l>>5) + j, n = 32*cb + (l&31)
__global__ __launch_bounds__(256) void split_w_kernel(const float *__restrict__ W, int64_t ldw,
                                                      uint16_t *__restrict__ wsp)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one (chunk, cb, lane) per thread
    if (idx >= kChunks * 8 * 64) return;
    const int lane = idx & 63, cb = (idx >> 6) & 7, chunk = idx >> 9;
    const int n = 32 * cb + (lane & 31);
    const int k0 = kChunk * chunk + 8 * (lane >> 5);
    uint16_t h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(W[(int64_t)(k0 + j) * ldw + n], h[j], m[j], l[j]);
    uint16_t *base = wsp + (size_t)chunk * (kChunkBytes / 2);
    uint16_t *dh = base + ((0 * 8 + cb) * 64 + lane) * 8;
    uint16_t *dm = base + ((1 * 8 + cb) * 64 + lane) * 8;
    uint16_t *dl = base + ((2 * 8 + cb) * 64 + lane) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dh[j] = h[j];
        dm[j] = m[j];
        dl[j] = l[j];
    }
}

// two floats -> packed (h, m, l) bf16 pairs, low half = x0 (one v_cvt_pk_bf16_f32 per part)
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l)
{
    f32x2 v = {x0, x1};
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    f32x2 r = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u)};
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    f32x2 q = {r.x - __uint_as_float(m << 16), r.y - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
}

__device__ __forceinline__ u32x4 pack8(const uint16_t (&v)[8])
{
    u32x4 r = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
               (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
    return r;
}

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(kThreads, 2) void gemm_xw256_kernel(const float *__restrict__ X, int64_t ldx,
                                                            const uint16_t *__restrict__ wsp,
                                                            float *__restrict__ Y, int64_t ldy,
                                                            int64_t M)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = (int64_t)blockIdx.x * kTileRows + 32 * wave + (lane & 31);
    const bool row_ok = row < M;
    const float *xrow = X + (row_ok ? row : 0) * ldx + 8 * (lane >> 5);

    f32x16 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // stage W stage s (kStage chunks of 24 KiB) into LDS buffer b: linear copy by all threads
    u32x4 wreg[kWLoads];
    auto w_load = [&](int st) {
        const u32x4 *src = (const u32x4 *)((const unsigned char *)wsp + (size_t)st * kStageBytes);
#pragma unroll
        for (int i = 0; i < kWLoads; ++i) wreg[i] = src[i * kThreads + tid];
    };
    auto w_store = [&](int b) {
        u32x4 *dst = (u32x4 *)(lds + b * kStageBytes);
#pragma unroll
        for (int i = 0; i < kWLoads; ++i) dst[i * kThreads + tid] = wreg[i];
    };
    // A fragments: a statically indexed ring of 3 steps (this step + two steps of prefetch); the
    // K loop is fully unrolled so that ring slots are compile-time registers and no copies (which
    // would force the prefetched loads to complete) are needed
    f32x4 ar[3][2];
    auto a_fetch = [&](int c, f32x4 &lo, f32x4 &hi) {
        const f32x4 *p = (const f32x4 *)(xrow + c * kChunk);
        lo = p[0];
        hi = p[1];
    };

    // split one prefetched X fragment into its three bf16 parts (rows past M contribute zeros)
    u32x4 Ah, Am, Al;
    auto split_frag = [&](const f32x4 &lo, const f32x4 &hi) {
        const float av[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            split3_pair(row_ok ? av[2 * j] : 0.f, row_ok ? av[2 * j + 1] : 0.f, h[j], m[j], l[j]);
        Ah = u32x4{h[0], h[1], h[2], h[3]};
        Am = u32x4{m[0], m[1], m[2], m[3]};
        Al = u32x4{l[0], l[1], l[2], l[3]};
    };

    w_load(0);
    a_fetch(0, ar[0][0], ar[0][1]);
    a_fetch(1, ar[1][0], ar[1][1]);
    w_store(0);
    split_frag(ar[0][0], ar[0][1]);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        const int st = c / kStage;
        if (c % kStage == 0) {
            __syncthreads();   // W stage st is in lds[st & 1]; everyone is done with lds[(st + 1) & 1]
            if ((st + 1) * kStage < kChunks) w_load(st + 1);   // prefetch the next W stage
        }
        if (c + 2 < kChunks) a_fetch(c + 2, ar[(c + 2) % 3][0], ar[(c + 2) % 3][1]);
        const u32x4 Xh = Ah, Xm = Am, Xl = Al;
        const unsigned char *buf = lds + (st & 1) * kStageBytes + (c % kStage) * kChunkBytes;
        // W fragments of column block nb+1 are read from LDS before the MFMAs of block nb issue
        // (two register sets), so the LDS latency hides under the MFMA run
        u32x4 Bf[2][3];
        auto b_read = [&](int nb, u32x4 (&dst)[3]) {
            dst[0] = *(const u32x4 *)(buf + ((0 * 8 + nb) * 64 + lane) * 16);
            dst[1] = *(const u32x4 *)(buf + ((1 * 8 + nb) * 64 + lane) * 16);
            dst[2] = *(const u32x4 *)(buf + ((2 * 8 + nb) * 64 + lane) * 16);
        };
        b_read(0, Bf[0]);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            if (nb + 1 < 8) b_read(nb + 1, Bf[(nb + 1) & 1]);
            const u32x4 Bh = Bf[nb & 1][0], Bm = Bf[nb & 1][1], Bl = Bf[nb & 1][2];
            // W fragment as the MFMA "A" operand, X fragment as "B": the accumulator then holds
            // the TRANSPOSED 32x32 tile (lane = output row, 4 consecutive registers = 4
            // consecutive output columns), which stores as 16 bytes per lane
            f32x16 t = acc[nb];
            __builtin_amdgcn_s_setprio(1);   // keep the partner wave's VALU/loads out of the MFMA run
            t = mfma(Bh, Xl, t);     // smallest terms first
            t = mfma(Bl, Xh, t);
            t = mfma(Bm, Xm, t);
            t = mfma(Bh, Xm, t);
            t = mfma(Bm, Xh, t);
            t = mfma(Bh, Xh, t);
            __builtin_amdgcn_s_setprio(0);
            acc[nb] = t;
        }
        if (c % kStage == kStage - 1 && c + 1 < kChunks) w_store((st + 1) & 1);
        if (c + 1 < kChunks) {
            // split the NEXT step's X fragment here, behind this step's MFMA run, so that it
            // overlaps the partner wave's MFMAs instead of idling the pipe right after the
            // barrier; the empty asm makes the results opaque so the compiler cannot sink the
            // split back across the barrier to its first use
            split_frag(ar[(c + 1) % 3][0], ar[(c + 1) % 3][1]);
            asm volatile("" : "+v"(Ah), "+v"(Am), "+v"(Al));
        }
    }

    // transposed C/D layout: output row = lane & 31, output column = 32*nb + 8*(reg >> 2) +
    // 4*(lane >> 5) + (reg & 3)
    if (row_ok) {
        float *yrow = Y + row * ldy + 4 * (lane >> 5);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                *(f32x4 *)(yrow + 32 * nb + 8 * g) = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Second scheme: TWO fp16 parts after an exact power-of-two scaling, three MFMAs per product.
//
// fp16 carries 11 significant bits, so x = h + m with h = fp16(x), m = fp16(x - h) represents x
// to 2^-22 relative, and  x·w ≈ h·h' + h·m' + m·h'  drops only the m·m' term (<= 2^-22): half the
// MFMA work of the 3 x bf16 scheme above at an error that stays well inside the path's 1e-5
// contract (measured against an fp64 product: tests/test_gemm_gpu.py).  What fp16 lacks is RANGE
// (max 65504, normals down to 2^-14), so both operands are first multiplied by a power of two
// (exact in fp32): W by 2^(14 - floor(log2 max|W|)), computed by the prep kernel; X by
// 2^(14 - floor(log2 B)) where B is any upper bound of max|X| the caller supplies in DEVICE memory
// — the layer knows one for free (the previous GEMM's output maximum times the adjacency's
// infinity norm, see pygcn_amd/spmm.py) — and the accumulators are scaled back on the way out.
// Elements more than 2^17 below the tensor's maximum lose RELATIVE precision (their m part goes
// subnormal): the absolute error per element is bounded by 2^-38 · max|X|; inputs with a wider
// dynamic range that matters should use the 3 x bf16 entry point.  Optionally the kernel reports
// max|Y| (one atomic max per wave) — the next layer's bound.
#ifndef GEMM_H2_RING
#define GEMM_H2_RING 4
#endif
#ifndef GEMM_H2_GRID
#define GEMM_H2_GRID 256          /* persistent workgroups: one per CU */
#endif
constexpr int kH2Ring = GEMM_H2_RING;                      // X ring: steps in flight + 1
constexpr int kH2ChunkBytes = 2 * 8 * kFragBytes;          // 2 splits x 8 column blocks = 16 KiB
constexpr int kH2StageBytes = kStage * kH2ChunkBytes;
constexpr int kH2WLoads = kH2StageBytes / 16 / kThreads;
static_assert(kH2StageBytes % (16 * kThreads) == 0, "W stage must divide over the workgroup");

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// X THROUGH LDS BY DMA (round 3).  The MFMA wants a lane to own 8 consecutive k of ONE row, so a
// wave's direct global loads touch 32 different rows per instruction: 64 L1 tag look-ups for 1 KiB
// (profiles/r02_gemm_pmc.md).  Since round 3 the X tile of a wave travels HBM -> LDS by
// `global_load_lds_dwordx4` (no VGPRs, asynchronous), in chunks of 32 rows x 32 columns (2 K steps):
// four instructions per chunk, in each of which every QUAD of lanes reads 64 contiguous bytes of
// one row (16 look-ups per KiB: the coalesced count).  The LDS position of a lane's 16 bytes is
// fixed by its lane id (base + 16·lane), but WHICH (row, columns) a lane fetches is free — the
// assignment below makes the later fragment reads (ds_read_b128, lane = MFMA row) conflict-free:
//   instruction j (0..3), lane l: quad Q = l >> 2, p = l & 3;  t = Q & 3, half = (Q >> 2) & 1,
//   cstep = Q >> 3;   row = 16·half + 4·j + t,   columns 32·chunk + 16·cstep + 4·p .. +3,
//   LDS byte = j·1040 + 16·l          (1040 = 1 KiB + one 16-byte slot: slot index ≡ j mod 16)
// MFMA lane (i, h) of K step cstep reads 32 bytes at  (i>>2 & 3)·1040 + 16·(4·(4·(i>>4) + (i&3) +
// 8·cstep) + 2·h): for the 16 lanes of a read phase the slot index mod 16 is j + 4·t + const —
// sixteen different bank groups.
constexpr int kXInstrBytes = 1024 + 16;
constexpr int kXChunkBytes = 4 * kXInstrBytes;            // 32 rows x 32 columns fp32, padded
constexpr int kXRing = 2;                                  // chunks per wave: one in use, one in flight
constexpr int kXLdsBytes = kWaves * kXRing * kXChunkBytes;

// The two decompositions the product kernel is instantiated for (template parameter SCH):
//   0 "h2": two fp16 parts of both operands after an exact power-of-two scaling, 3 MFMAs per product
//           (22-bit significand; needs an upper bound of max|X|);
//   1 "b3": three bf16 parts, 6 MFMAs per product, no scaling and no bound — a 24-bit significand,
//           the fp32-equivalent form of `torch.mm(input, self.weight)` (pygcn/layers.py:33).
// Same pipeline (persistent workgroups, X and W by HBM -> LDS DMA, cross-tile prefetch, compile-time
// store sections); b3's W stage is ONE K step (24 KiB: three parts x 8 column blocks) so that two
// stages + the X rings fit the 160 KiB of LDS — its 48 MFMAs per barrier equal h2's 2 x 24.
template <int SCH> struct SchemeK {
    static constexpr int NS = SCH == 0 ? 2 : 3;                    // parts per operand
    static constexpr int KS = SCH == 0 ? kStage : 1;               // K steps of W per LDS stage (= per barrier)
    static constexpr int ChunkBytes = NS * 8 * kFragBytes;         // one K step of W: 16 / 24 KiB
    static constexpr int StageBytes = KS * ChunkBytes;             // 32 / 24 KiB
    static constexpr int WShare = StageBytes / kWaves;             // a wave's part of a stage: 4 / 3 KiB
};
static_assert(SchemeK<0>::StageBytes == 2 * 8 * kFragBytes * kStage, "h2 stage");

// exponent e with 2^e <= v < 2^(e+1) for finite v > 0 (0 for zero / non-finite: no scaling)
__device__ __forceinline__ int floor_log2f(float v)
{
    const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
    const int e = (int)(b >> 23);
    if (b == 0 || e == 255) return 0;
    if (e == 0) return -126 - __clz(b << 9) - 1;          // fp32 subnormal
    return e - 127;
}

__device__ __forceinline__ float pow2f(int e)              // 2^e for -126 <= e <= 127
{
    return __uint_as_float((uint32_t)(e + 127) << 23);
}

// header of the workspace the prep kernel writes in front of the W image
struct H2Header {
    int w_exp;        // W was multiplied by 2^w_exp
    int pad[3];
};
constexpr int kH2HeaderBytes = 256;

// one workgroup: max|W|, scale, split into fp16 (h, m), fragment-ordered like the bf16 image
__global__ __launch_bounds__(1024) void split_w_h2_kernel(const float *__restrict__ W, int64_t ldw,
                                                          unsigned char *__restrict__ ws)
{
    __shared__ float red[16];
    __shared__ int s_exp;
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int i = tid; i < kK * kN; i += 1024) mx = fmaxf(mx, fabsf(W[(int64_t)(i >> 8) * ldw + (i & 255)]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        float m = 0.f;
        for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i]);
        int e = 14 - floor_log2f(m);
        e = e > 126 ? 126 : (e < -126 ? -126 : e);
        if (!(m > 0.f) || !(m <= 3.4028235e38f)) e = 0;
        s_exp = e;
        ((H2Header *)ws)->w_exp = e;
    }
    __syncthreads();
    const float sc = pow2f(s_exp);
    uint16_t *img = (uint16_t *)(ws + kH2HeaderBytes);
    for (int idx = tid; idx < kChunks * 8 * 64; idx += 1024) {          // one (chunk, cb, lane)
        const int lane = idx & 63, cb = (idx >> 6) & 7, chunk = idx >> 9;
        const int n = 32 * cb + (lane & 31);
        const int k0 = kChunk * chunk + 8 * (lane >> 5);
        uint16_t *base = img + (size_t)chunk * (kH2ChunkBytes / 2);
        uint16_t *dh = base + ((0 * 8 + cb) * 64 + lane) * 8;
        uint16_t *dm = base + ((1 * 8 + cb) * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = W[(int64_t)(k0 + j) * ldw + n] * sc;
            const _Float16 h = (_Float16)x;
            const _Float16 m = (_Float16)(x - (float)h);
            dh[j] = __builtin_bit_cast(uint16_t, h);
            dm[j] = __builtin_bit_cast(uint16_t, m);
        }
    }
}

__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a),
                                                  __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// store-side options of gemm_xw256_h2_kernel (host: struct gcn_gemm_epilogue)
struct H2Epi {
    const float *bias;          // [256] or NULL
    int relu;
    uint32_t drop_thresh;       // keep an element iff its 16-bit field >= drop_thresh (0: none)
    float drop_scale;           // 65536 / (65536 - drop_thresh)
    uint32_t seed_lo, seed_hi;
    int64_t drop_row_base;      // added to the row index in the dropout counter
    const uint64_t *seed_dev;   // optional: the seed as of execution time (hipGraph replays)
    const float *mask_src;      // optional backward mask: y = mask_src > 0 ? y * mask_scale : 0
    int64_t ld_mask;
    const int32_t *mask_rows;   // optional: mask row of output row r (default: its input row)
    float mask_scale;
    // (gemm_xw256_s16_kernel only) the ReLU / dropout result as ONE BIT per element, 32 bytes per row, in the
    // kernel's own lane order: row r, lane quarter q (0..3), word w (0..1): bit 4 (cb & 7) + j of word
    // 2 q + w at r * 8 words <-> column 16 (8 w + (cb & 7)) + 4 q + j.  Written by the forward epilogue
    // (keep_bits_out), read by the backward mask (mask_bits, instead of 1 KiB of mask_src per row).
    uint32_t *keep_bits_out;
    const uint32_t *mask_bits;
};

// Philox4x32-10 — the SAME keep function of (seed, row, f) as the SpMM epilogue (gcn_spmm.hip,
// include/gcn_spmm.h): block = ((f >> 4) << 1) | ((f >> 2) & 1), eight 16-bit fields per call.  A
// lane of the transposed accumulator tile stores columns 32nb + 8g + 4h + (0..3) (h = lane half),
// so groups g = 2q and 2q + 1 are fields 0..3 and 4..7 of ONE block: one call per 8 elements.
__device__ __forceinline__ void h2_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                          uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (one 32 x 32 -> 64 multiply per product — v_mad_u64_u32 — instead of a mul_hi + mul_lo
        //  pair: integer multiplies run at a quarter of the VALU rate and dominate the epilogue)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 16 bytes per lane straight from global memory into LDS (asynchronous, counted by vmcnt like a
// load): the wave's 64 x 16 bytes land contiguously at LDS byte address `lds_addr` (wave-uniform,
// through M0).  Issued as inline assembly ON PURPOSE: hipcc's wait-count pass
// treats every LDS access after a `__builtin_amdgcn_global_load_lds` as dependent on it and drains
// the whole prefetch with `s_waitcnt vmcnt(0)`; here the kernel places its own counted waits
// (dma_wait<N>: N = DMA instructions issued AFTER the youngest one that must have landed — vector
// loads complete in order, so "at most N outstanding" means everything older is in LDS; stores
// that complete early only make the wait longer, never shorter).
__device__ __forceinline__ void dma16(const void *g, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void dma16_x(const void *g, uint32_t lds_addr)     // (the X stream)
{
    dma16(g, lds_addr);
}
template <int N> __device__ __forceinline__ void dma_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// EPI 0: plain product; EPI 2: backward mask in the store (its own instantiation, so that no branch
// surrounds the mask loads: they are requested one column block AHEAD, before the current block's
// stores, and the wait for them leaves only those stores outstanding — fetched where they were used
// they cost one drain of all earlier stores per column block); both persistent with the cross-tile pipeline.
// EPI 1 (FWD_EPI): bias + ReLU + Philox dropout in the store; the Philox state would not fit next to
// the next tile's prefetched fragments (spills), so this instantiation runs one tile per workgroup.
#ifdef GEMM_PROFILE_STAMPS   /* experiment builds only (tools/gemm_stamps_probe.py): where a tile's cycles go */
__device__ unsigned long long g_gemm_stamps[8];
__device__ unsigned long long g_gemm_step_stamps[16];
// (inside K steps 4 and 5 of every tile: cycles between seven points of the step, summed per wave)
#define GEMM_STEP_STAMP(i)                                                             \
    do {                                                                               \
        if (c == 4 || c == 5) {                                                        \
            unsigned long long t_;                                                     \
            GEMM_STAMP(t_);                                                            \
            if ((i) > 0) ph[8 * (c - 4) + (i)] += t_ - t_prev;                         \
            t_prev = t_;                                                               \
        }                                                                              \
    } while (0)
#define GEMM_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); (v) = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GEMM_STAMP(v) do { } while (0)
#define GEMM_STEP_STAMP(i) do { } while (0)
#endif

template <int EPI, int SCH = 0>
__global__ __launch_bounds__(kThreads, 2) void gemm_xw256_h2_kernel(
    const float *__restrict__ X, int64_t ldx, const int32_t *__restrict__ x_rows,
    const unsigned char *__restrict__ ws, const float *__restrict__ x_bound, float *__restrict__ Y,
    int64_t ldy, int64_t M, uint32_t *__restrict__ y_absmax, const H2Epi ep)
{
    // SCH: 0 two scaled fp16 parts (x_bound required), 1 three bf16 parts (x_bound unused) — SchemeK
    typedef SchemeK<SCH> SK;
    constexpr int NS = SK::NS, KS = SK::KS, kSchChunkBytes = SK::ChunkBytes, kSchStageBytes = SK::StageBytes;
    // EPI: 0 plain, 2 backward mask; forward epilogues (bias always, zeros when there is none):
    // 1 bias only, 4 + ReLU, 5 + ReLU + dropout at p = 1/2 (one-bit keep fields), 6 + ReLU + dropout
    // at any other p (16-bit fields).  Compile-time options: a uniform branch per column group in
    // the store section costs this issue-bound kernel more than the arithmetic it skips.
    constexpr bool FWD_EPI = EPI == 1 || EPI >= 4, MASKED = EPI == 2;
    constexpr bool RELU = EPI >= 4, DROP1 = EPI == 5, DROP16 = EPI == 6;
    const float *__restrict__ mask_src = ep.mask_src;
    const int64_t ld_mask = ep.ld_mask;
    const float mask_scale = ep.mask_scale;
    // PERSISTENT: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, … and the software
    // pipeline runs ACROSS tile boundaries — the X fragments of the next tile's first steps and
    // its first W stage are already in flight while the current tile's last steps are multiplied
    // and its 128 accumulator registers are stored, so no tile pays a cold prologue (with one
    // workgroup per CU nothing else would cover it).  The X ring has kH2Ring slots with static
    // indices; kChunks % kH2Ring == 0 keeps slot = step % kH2Ring valid across the boundary.
    static_assert(kChunks % kH2Ring == 0, "the X ring must divide the K steps of a tile");
    static_assert(kChunks % KS == 0 && (kChunks / KS) % 2 == 0, "W stages must alternate evenly");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kSchStageBytes];    // the W stage buffers
    // (a separate LDS object, so the compiler's wait-count pass can tell a DMA into an X ring from
    //  a store into a W stage and does not drain the X prefetch at every W stage)
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];              // X rings, per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char *wimg = ws + kH2HeaderBytes;
    const int64_t n_tiles = (M + kTileRows - 1) / kTileRows;
    // Forward epilogue: the bias and the device-resident seed are read ONCE here, the bias into LDS.
    // Read where they are used (in the store section) they compile to vector loads — per-lane
    // addresses after the half-wave select — and a vector load that is consumed at once is waited
    // for with `s_waitcnt vmcnt(0)`: loads and stores retire in order on one counter, so every one
    // of the 32 column groups of a tile drained all earlier stores AND the next tile's prefetch
    // (what held this instantiation at 0.45 of HBM against 0.54 for the plain product).
    __shared__ __attribute__((aligned(16))) float bias_lds[FWD_EPI ? kN : 4];
    uint32_t seed_k0 = ep.seed_lo, seed_k1 = ep.seed_hi;
    if (FWD_EPI) {
        if (tid < kN) bias_lds[tid] = ep.bias != nullptr ? ep.bias[tid] : 0.f;
        if (ep.seed_dev != nullptr) {
            const uint64_t sd = *ep.seed_dev;
            seed_k0 = (uint32_t)sd;
            seed_k1 = (uint32_t)(sd >> 32);
        }
        __syncthreads();
    }

    // scales (wave-uniform scalars; the three-part bf16 scheme has fp32's range and needs none)
    float xs = 1.f, back_a = 1.f, back_b = 1.f;
    bool one_step = true;
    if constexpr (SCH == 0) {
        int x_exp = 14 - floor_log2f(*x_bound);
        x_exp = x_exp > 126 ? 126 : (x_exp < -126 ? -126 : x_exp);
        bool poisoned = false;
        {
            const float b = *x_bound;
            if (!(b > 0.f)) x_exp = 0;                    // zero bound (an all-zero operand): no scaling
            // A bound that is inf / NaN is the sentinel of an overflow upstream (y_absmax of a launch
            // whose own bound was too small): it must not turn into silent zeros — poison the result.
            poisoned = !(b <= 3.4028235e38f);
            if (poisoned) x_exp = 0;
        }
        xs = pow2f(x_exp);
        const int back = -(x_exp + ((const H2Header *)ws)->w_exp);       // result * 2^back, in two
        // (one multiply when 2^back is a normal float — always, outside the edges of fp32 —, else two)
        one_step = back >= -126 && back <= 127;
        back_a = poisoned ? __uint_as_float(0x7fc00000u) : pow2f(one_step ? back : back / 2);
        back_b = one_step ? 1.f : pow2f(back - back / 2);    // exact steps
    }

    static_assert(kStage == 2 && kWaves == 8, "the DMA pipeline is written for 2-step W stages and 8 waves");
    static_assert(SK::WShare % 1024 == 0, "a wave's part of a W stage is whole DMA instructions");
    const unsigned char *wl = wimg;
    const uint32_t w_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds +
                           __builtin_amdgcn_readfirstlane(wave) * SK::WShare;
    // this wave's eighth (h2: 4 KiB = 4 DMA instructions, b3: 3) of W stage `st` -> W buffer `b`
#ifndef GEMM_ABLATE_WPIECES    /* ablation builds only (results garbage): DMA instructions per wave and W stage */
#define GEMM_ABLATE_WPIECES (-1)
#endif
    [[maybe_unused]] bool ablate_first_tile = GEMM_ABLATE_WPIECES >= 0;
    auto w_issue = [&](int st, int b) {
        const unsigned char *src = wl + (size_t)st * kSchStageBytes + wave * SK::WShare + lane * 16;
        constexpr int kPieces = GEMM_ABLATE_WPIECES >= 0 ? GEMM_ABLATE_WPIECES : SK::WShare / 1024;
#pragma unroll
        for (int i = 0; i < SK::WShare / 1024; ++i)      // (ablation: all pieces during a workgroup's FIRST tile, so
            if (i < kPieces || ablate_first_tile)        //  that the stale W stages the later tiles multiply are real data)
                dma16(src + i * 1024, w_lds + b * kSchStageBytes + i * 1024);
    };
    unsigned char *xl = xlds + __builtin_amdgcn_readfirstlane(wave) * (kXRing * kXChunkBytes);
    // (what this lane FETCHES in instruction j of a chunk, and where this MFMA lane READS)
    const int ld_row0 = 16 * ((lane >> 4) & 1) + ((lane >> 2) & 3);       // + 4·j
    const int ld_col = 16 * (lane >> 5) + 4 * (lane & 3);                 // + 32·chunk
    const int rd_off = (((lane & 31) >> 2) & 3) * kXInstrBytes +
                       16 * (4 * (4 * ((lane & 31) >> 4) + (lane & 3)) + 2 * (lane >> 5));   // + 512·cstep
    auto x_issue = [&](const float *const (&src)[4], int chunk, int slot) {
        const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)xl + slot * kXChunkBytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_x(src[j] + 32 * chunk, base + j * kXInstrBytes);
    };
    auto a_read = [&](int cstep, int slot, f32x4 &lo, f32x4 &hi) {
        const unsigned char *q = xl + slot * kXChunkBytes + rd_off + 512 * cstep;
        lo = *(const f32x4 *)q;
        hi = *(const f32x4 *)(q + 16);
    };
    u32x4 Ah, Am;
    [[maybe_unused]] u32x4 Al;                     // (third part: SCH 1 only)
    auto split_frag = [&](const f32x4 &lo, const f32x4 &hi, bool ok) {
        const float av[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t h[4], m[4];
        if constexpr (SCH == 1) {
            uint32_t l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                split3_pair(ok ? av[2 * j] : 0.f, ok ? av[2 * j + 1] : 0.f, h[j], m[j], l[j]);
            Ah = u32x4{h[0], h[1], h[2], h[3]};
            Am = u32x4{m[0], m[1], m[2], m[3]};
            Al = u32x4{l[0], l[1], l[2], l[3]};
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = ok ? av[2 * j] * xs : 0.f, x1 = ok ? av[2 * j + 1] * xs : 0.f;
            f32x2 v = {x0, x1};
            const h16x2 hh = __builtin_convertvector(v, h16x2);
            const f32x2 hb = __builtin_convertvector(hh, f32x2);
            f32x2 r = {x0 - hb.x, x1 - hb.y};
            const h16x2 mm = __builtin_convertvector(r, h16x2);
            h[j] = __builtin_bit_cast(uint32_t, hh);
            m[j] = __builtin_bit_cast(uint32_t, mm);
        }
        Ah = u32x4{h[0], h[1], h[2], h[3]};
        Am = u32x4{m[0], m[1], m[2], m[3]};
    };
    // (row of this lane in tile t, is it inside the matrix, the input row it reads)
    auto tile_rows = [&](int64_t t, int64_t &row, bool &ok, int64_t &src) {
        row = t * kTileRows + 32 * wave + (lane & 31);
        ok = t < n_tiles && row < M;
        // optional gather: output row `row` is the product of input row x_rows[row]
        src = ok ? (x_rows ? (int64_t)x_rows[row] : row) : 0;
    };

    int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    int64_t row, src_row;
    bool row_ok;
    tile_rows(tile, row, row_ok, src_row);
    // the four rows this lane fetches per chunk (one per DMA instruction), as row pointers
    auto load_rows = [&](int64_t t, const float *(&src)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = t * kTileRows + 32 * wave + ld_row0 + 4 * j;
            const bool ok = t < n_tiles && r < M;
            const int64_t sr = ok ? (x_rows ? (int64_t)x_rows[r] : r) : 0;
            src[j] = X + sr * ldx + ld_col;
        }
    };
    const float *xsrc[4];
    load_rows(tile, xsrc);
    x_issue(xsrc, 0, 0);
    w_issue(0, 0);
    x_issue(xsrc, 1, 1);
    dma_wait<4>();                 // X chunk 0 and this wave's part of W stage 0 are in LDS
    {
        f32x4 lo, hi;
        a_read(0, 0, lo, hi);
        split_frag(lo, hi, row_ok);
    }
    uint32_t vmax = 0u;   // max of |y| as BITS: unsigned order = float order for finite values, and
                          // inf / NaN patterns sort above every finite one (an overflow is never lost)

    [[maybe_unused]] unsigned long long st_k = 0, st_s = 0, st_n = 0, st_a = 0, st_b = 0, st_c = 0, st_begin = 0, st_f2 = 0, st_f8 = 0;
    [[maybe_unused]] unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    GEMM_STAMP(st_begin);
    for (; tile < n_tiles; tile += gridDim.x) {
        GEMM_STAMP(st_a);
#ifndef GEMM_H2_EPI_PERSIST
#define GEMM_H2_EPI_PERSIST 1
#endif
        if (GEMM_ABLATE_WPIECES >= 0 && tile != (int64_t)blockIdx.x) ablate_first_tile = false;
        const bool has_next = (!FWD_EPI || GEMM_H2_EPI_PERSIST) && tile + gridDim.x < n_tiles;   // (uniform)
        // the W image's 32 load addresses are loop-invariant; left visible, hipcc hoists all of
        // them out of the tile loop as 64-bit VGPR pairs and spills 50 registers
        asm volatile("" : "+s"(wl));
        int64_t row_n, src_n;
        bool ok_n;
        tile_rows(tile + gridDim.x, row_n, ok_n, src_n);
        const float *xsrc_n[4];
        load_rows(tile + gridDim.x, xsrc_n);
        // (masked form: the row of the mask this lane will read in the store section — looked up
        //  here, a whole K loop ahead of its use)
        int64_t mask_row = src_row;
        if (MASKED && ep.mask_rows != nullptr) mask_row = row_ok ? (int64_t)ep.mask_rows[row] : 0;
        f32x16 acc[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
        // ---- the store section of a tile, as two pieces: what is computed once per tile, and one
        // column block's scaling / epilogue / four 16-byte stores.  They run after the K loop: 32
        // store instructions per wave at one point of the tile, the matrix pipe idle for 12 000 cycles
        // (gemm_xw256_s16_kernel hides them under the next tile; measured and dropped here: the
        // blocks inside the tile's last K step — DESIGN §7).
        float *yrow = nullptr;
        const float *mrow = nullptr;
        uint32_t r1[4] = {0u, 0u, 0u, 0u};
        f32x4 mk[2][4];                 // the mask of column block nb in mk[nb & 1]
        auto store_prelude = [&]() {
            yrow = Y + row * ldy + 4 * (lane >> 5);
            // optional fused backward of ReLU / dropout: y = mask_src[src_row] > 0 ? y * scale : 0
            mrow = MASKED ? mask_src + mask_row * ld_mask + 4 * (lane >> 5) : nullptr;
            // dropout at p = 1/2: 128 one-bit keep fields per Philox call (gcn_spmm.hip,
            // apply_dropout) — ONE call covers all of this lane's 128 columns of the row
            // (block = this lane's half h; column 32nb + 8g + 4h + j is bit 16(nb & 1) + 4g + j of
            // word nb >> 1); other p: eight 16-bit fields per call, 16 calls per lane and tile
            if (DROP1) {
                const int64_t drow = row + ep.drop_row_base;
                uint32_t cw = (uint32_t)(lane >> 5);
                asm volatile("" : "+v"(cw));
                h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, seed_k0, seed_k1, r1);
            }
            if (MASKED) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mk[0][g] = *(const f32x4 *)(mrow + 8 * g);
            }
        };
        auto store_block = [&](int nb) {

                // dropout: the two Philox calls of this column block (8 keep fields each: groups 0-1 and
                // 2-3) are computed TOGETHER, so that their two dependent chains of quarter-rate
                // 32 x 32 -> 64 multiplies interleave; the scheduling barrier below then sits between
                // column blocks, not between the chains
                uint32_t r8[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
                if (DROP16) {
                    const uint32_t k0 = seed_k0, k1 = seed_k1;
                    const int64_t drow = row + ep.drop_row_base;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // (opaque: the first Philox round's product of this counter word with its
                        //  constant does not depend on the tile, and hipcc hoists all of them out
                        //  of the tile loop — spilled registers)
                        uint32_t cw = ((uint32_t)(2 * nb + q) << 1) | (uint32_t)(lane >> 5);
                        asm volatile("" : "+v"(cw));
                        h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, k0, k1, r8[q]);
                    }
                }
                // backward mask: the NEXT column block's four 16-byte pieces are requested here,
                // ahead of this block's stores (loads and stores retire in order on one counter: the
                // wait for them at the top of the next block then leaves exactly these 4 stores out)
                if (MASKED && nb + 1 < 8) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) mk[(nb + 1) & 1][g] = *(const f32x4 *)(mrow + 32 * (nb + 1) + 8 * g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                    if constexpr (SCH == 0) {                                  // undo the operand scaling
                        v.x *= back_a; v.y *= back_a; v.z *= back_a; v.w *= back_a;
                        if (!one_step) {                                       // (wave-uniform, rare)
                            v.x *= back_b; v.y *= back_b; v.z *= back_b; v.w *= back_b;
                        }
                    }
                    // forward epilogue of the layer when the GEMM is its LAST stage
                    // ((Â·X)·W + b, pygcn/layers.py:33-36 reassociated): bias, ReLU, inverted dropout
                    const int f = 32 * nb + 8 * g + 4 * (lane >> 5);          // first of 4 columns
                    if (FWD_EPI) {
                        // this lane's 4 bias values from LDS (a broadcast read: lgkmcnt, not vmcnt)
                        const f32x4 b4 = *(const f32x4 *)(bias_lds + f);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
                    if (RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (DROP1) {
                        const uint32_t nib = r1[nb >> 1] >> (16 * (nb & 1) + 4 * g);
                        v.x = (nib & 1u) ? v.x * ep.drop_scale : 0.f;
                        v.y = (nib & 2u) ? v.y * ep.drop_scale : 0.f;
                        v.z = (nib & 4u) ? v.z * ep.drop_scale : 0.f;
                        v.w = (nib & 8u) ? v.w * ep.drop_scale : 0.f;
                    } else if (DROP16) {
                        const uint32_t *rq = r8[g >> 1];
                        const uint32_t w0 = (g & 1) ? rq[2] : rq[0], w1 = (g & 1) ? rq[3] : rq[1];
                        v.x = (w0 & 0xFFFFu) >= ep.drop_thresh ? v.x * ep.drop_scale : 0.f;
                        v.y = (w0 >> 16) >= ep.drop_thresh ? v.y * ep.drop_scale : 0.f;
                        v.z = (w1 & 0xFFFFu) >= ep.drop_thresh ? v.z * ep.drop_scale : 0.f;
                        v.w = (w1 >> 16) >= ep.drop_thresh ? v.w * ep.drop_scale : 0.f;
                    }
                    if (MASKED) {
                        const f32x4 m = mk[nb & 1][g];
                        v.x = m.x > 0.f ? v.x * mask_scale : 0.f;
                        v.y = m.y > 0.f ? v.y * mask_scale : 0.f;
                        v.z = m.z > 0.f ? v.z * mask_scale : 0.f;
                        v.w = m.w > 0.f ? v.w * mask_scale : 0.f;
                    }
#ifdef GEMM_H2_ABLATE_STORES      /* ablation build only: keeps the arithmetic, drops the traffic */
                    if (v.x == 1.2345e-30f)
#endif
                    *(f32x4 *)(yrow + 32 * nb + 8 * g) = v;
                    if (y_absmax != nullptr) {                                 // (wave-uniform)
                        if (RELU)                    // stored values are >= 0 (or NaN): the bits as they are
                            vmax = max(max(vmax, max(__float_as_uint(v.x), __float_as_uint(v.y))),
                                       max(__float_as_uint(v.z), __float_as_uint(v.w)));
                        else
                            vmax = max(max(vmax, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu)),
                                       max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
                    }
                }
                // (keeps hipcc from running all 16 Philox chains of the tile side by side — 64 live
                //  registers on top of the accumulators)
                if (DROP16) __builtin_amdgcn_sched_barrier(0);
            
        };
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int st = c / KS;
            GEMM_STEP_STAMP(0);
            if (c % KS == 0) {
                // every wave waited for ITS part of W stage st before it got here (end of the
                // previous stage / prologue) and has read its last fragment of stage st - 1
                __builtin_amdgcn_s_barrier();
                GEMM_STEP_STAMP(1);
                if ((st + 1) * KS < kChunks)
                    w_issue(st + 1, (st + 1) & 1);
                else if (has_next)
                    w_issue(0, 0);                                // the next tile's first stage
            }
            if (c & 1) {
                // chunk m = (c + 3) / 2 goes into the ring slot chunk m - 2 has just left (its last
                // fragment was read at the end of step c - 1): three K steps of flight time
                constexpr int kCh = kChunks / 2;
                const int m = (c + 3) / 2;
                if (m < kCh)
                    x_issue(xsrc, m, m & 1);
                else if (has_next)
                    x_issue(xsrc_n, m - kCh, m & 1);
            }
            GEMM_STEP_STAMP(2);
            const u32x4 Xh = Ah, Xm = Am;
            [[maybe_unused]] const u32x4 Xl = Al;
            const unsigned char *buf = lds + (st & 1) * kSchStageBytes + (c % KS) * kSchChunkBytes;
            u32x4 Bf[2][NS];
            auto b_read = [&](int nb, u32x4 (&dst)[NS]) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) dst[sp] = *(const u32x4 *)(buf + ((sp * 8 + nb) * 64 + lane) * 16);
            };
            b_read(0, Bf[0]);
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                if (nb + 1 < 8) b_read(nb + 1, Bf[(nb + 1) & 1]);
                const u32x4 Bh = Bf[nb & 1][0], Bm = Bf[nb & 1][1];
                f32x16 t = acc[nb];
                __builtin_amdgcn_s_setprio(1);
                if constexpr (SCH == 1) {
                    // (the order of gemm_xw256_kernel, smallest terms first: results are bit-identical)
                    const u32x4 Bl = Bf[nb & 1][NS - 1];
                    t = mfma(Bh, Xl, t);
                    t = mfma(Bl, Xh, t);
                    t = mfma(Bm, Xm, t);
                    t = mfma(Bh, Xm, t);
                    t = mfma(Bm, Xh, t);
                    t = mfma(Bh, Xh, t);
                } else {
                    t = mfma_h(Bm, Xh, t);       // smaller terms first
                    t = mfma_h(Bh, Xm, t);
                    t = mfma_h(Bh, Xh, t);
                }
                __builtin_amdgcn_s_setprio(0);
                acc[nb] = t;
                if (nb == 0) GEMM_STEP_STAMP(3);
                if (nb == 3) GEMM_STEP_STAMP(4);
                if (nb == 7) GEMM_STEP_STAMP(5);
            }
            if (c % KS == KS - 1) {
                // before the next stage: this wave's part of W stage st + 1 and the X chunk of the
                // next K step must be in LDS.  Younger than both: only the 4 DMA instructions of an
                // X chunk issued at the top of THIS step (odd steps) — if they were.  (b3, one-step
                // stages: on even steps the W stage just issued is the youngest — a full wait, 48
                // MFMAs after its issue, the flight time h2's stages have too.)
                constexpr int kCh = kChunks / 2;
                const bool issued_x = (c & 1) && (((c + 3) / 2 < kCh) || has_next);
                if (c + 1 < kChunks || has_next) {
                    if (issued_x) dma_wait<4>();
                    else dma_wait<0>();
                }
            }
            GEMM_STEP_STAMP(6);
#ifdef GEMM_PROFILE_STAMPS
            if (c == 1) { unsigned long long t2; GEMM_STAMP(t2); st_f2 += t2 - st_a; }
            if (c == 7) { unsigned long long t8; GEMM_STAMP(t8); st_f8 += t8 - st_a; }
#endif
            if (c + 1 < kChunks || has_next) {
                const int cn = (c + 1) % kChunks;                 // (chunk slots alternate: 8 chunks per tile)
                f32x4 lo, hi;
                a_read(cn & 1, (cn >> 1) & 1, lo, hi);
                split_frag(lo, hi, c + 1 < kChunks ? row_ok : ok_n);
                if constexpr (SCH == 1) asm volatile("" : "+v"(Ah), "+v"(Am), "+v"(Al));
                else asm volatile("" : "+v"(Ah), "+v"(Am));
            }
            GEMM_STEP_STAMP(7);
        }

        GEMM_STAMP(st_b);
        if (row_ok) {
            store_prelude();
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) store_block(nb);
        }
        GEMM_STAMP(st_c);
#ifdef GEMM_PROFILE_STAMPS
        st_k += st_b - st_a;
        st_s += st_c - st_b;
        st_n += 1;
#endif
        row = row_n;
        row_ok = ok_n;
        src_row = src_n;
#pragma unroll
        for (int j = 0; j < 4; ++j) xsrc[j] = xsrc_n[j];
    }
    if (y_absmax != nullptr) {                 // |y| >= 0: float order == unsigned order of the bits
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off, 64));
        if (lane == 0 && vmax != 0u) atomicMax(y_absmax, vmax);
    }
#ifdef GEMM_PROFILE_STAMPS
    if (lane == 0) {           // per wave: cycles in K loops / in store sections / tiles / whole lifetime
        unsigned long long st_end;
        GEMM_STAMP(st_end);
        atomicAdd(&g_gemm_stamps[0], st_k);
        atomicAdd(&g_gemm_stamps[1], st_s);
        atomicAdd(&g_gemm_stamps[2], st_n);
        atomicAdd(&g_gemm_stamps[3], st_end - st_begin);
        atomicAdd(&g_gemm_stamps[4], 1ull);
        atomicAdd(&g_gemm_stamps[5], st_f2);
        atomicAdd(&g_gemm_stamps[6], st_f8);
        for (int i = 0; i < 16; ++i) atomicAdd(&g_gemm_step_stamps[i], ph[i]);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// The three-part (fp32-equivalent) product with its STORES UNDER THE NEXT TILE'S MULTIPLICATIONS —
// `gemm_xw256_s16_kernel` (round 4; contiguous rows; 6.2 ms against 7.05 at M = 10^7).
//
// Same arithmetic as gemm_xw256_h2_kernel<·, 1> (three bf16 parts per operand, six MFMAs per product,
// smallest terms first).  What the ablations of that kernel say (profiles/r04_gemm_stamps.md): of its
// 7.1 ms at M = 10^7 the W staging is worth 0.2 ms and the STORE SECTION 1.1 ms — 32 store
// instructions per wave issued at one point of a tile, all eight waves at once, the matrix pipe idle
// for 12 000 cycles; moving them inside the tile's last K step changed nothing (the step then takes
// as long as the stores).  They can only hide under the multiplications of ANOTHER tile, which needs
// the finished tile's result registers AND the running tile's accumulators — 2 x 128 — unless a
// wave's tile is half as large.  So here:
//
//   * a workgroup tile is 128 rows, 16 per wave, on 16x16x32 MFMA tiles: lane (n = lane & 15,
//     q = lane >> 4) owns output row n and, in column block cb (16 columns), columns 16cb + 4q .. +3
//     (the transposed product again: W fragment as the A operand, X fragment as B; a lane stores
//     16 bytes, a store instruction covers 16 rows x 64 contiguous bytes): 64 accumulator registers;
//   * the previous tile's results stay in 64 registers (`prev`) and leave one column block at a time,
//     two per K stage, BETWEEN the stage's MFMA groups;
//   * NOTHING is issued as a burst (S16_SPREAD; the burst form of the same kernel runs 6.7 ms): behind
//     column block 0 / 2 / 4 of a stage go two W pieces each, behind 6 the X chunk, behind 8 and 10 one
//     store each, behind 12 the next chunk is waited for (a counted wait: it was issued a stage ago) and
//     split into the other fragment set, under the last column blocks' MFMAs;
//   * a W stage is one K chunk of 32 x all 256 columns x three parts (48 KiB, two buffers), so a
//     128-row tile has 8 stages and a row costs as many barriers as before; the W image is re-read
//     per 128 rows instead of per 256 (L2 hits; its whole cost was the 0.2 ms above);
//   * X chunks are 16 rows x 32 columns (2 DMA instructions per wave), private to the wave, ring of 2,
//     split ONCE per chunk into the three parts (12 registers, two sets).  Layout of one DMA
//     instruction's KiB: [q 4][row 8][32 bytes] — lane (n, q) reads its 32 bytes at
//     1040 (n >> 3) + 256 q + 32 (n & 7): the 16-byte slot index mod 16 is (n >> 3) + 2 (n & 7),
//     distinct over each of ds_read_b128's 16-lane groups (MI355X_MICROARCH.md, LDS table);
//   * every address is a uniform base (SGPR pair: tile origin, W stage) plus a 32-bit per-lane offset
//     that never changes — no 64-bit address registers (the 16x16x32 form of the old pipeline spilled
//     on them).  This needs contiguous X rows: row lists stay on gemm_xw256_h2_kernel<·, 1>, and so
//     does dropout at p != 1/2 (sixteen Philox calls per tile keep hipcc from unrolling the stages).
//
// Sums run over 8 K chunks of 32 instead of 16 of 16: results differ from gemm_xw256_h2_kernel<·, 1>
// (and round 1's kernel) in fp32 summation order — not bitwise, same accuracy against fp64.
//
// Measured and dropped on the way (profiles/r04_gemm_stamps.md): the same schedule on 32x32x16 MFMA
// tiles with a wave tile of 32 rows x 128 columns (bit-identical to gemm_xw256_h2_kernel<·, 1>; an X
// chunk shared by the two waves of a row group, ring of 3 behind the barrier): 7.8 ms as bursts,
// 7.15 spread — no better than the kernel it was to replace; the 16x16x32 tiles on the OLD schedule
// (8 waves x 32 rows, 256-row tiles): 7.03 against 7.06.
constexpr int kT16StageBytes = 3 * 8 * kFragBytes;           // 24 KiB: K chunk of 32 x 128 columns x 3 parts
constexpr int kT16Stages = 16;
constexpr int kS16Rows = 16 * kWaves;                        // 128
constexpr int kS16StageBytes = 2 * kT16StageBytes;           // 48 KiB: K chunk of 32 x 256 columns x 3 parts
constexpr int kS16Stages = 8;
constexpr int kS16XInstr = 1024 + 16;
constexpr int kS16XChunk = 2 * kS16XInstr;                   // 16 rows x 32 columns fp32, padded
constexpr int kS16XLds = kWaves * 2 * kS16XChunk;
constexpr int kS16LdsBytes = 2 * kS16StageBytes + kS16XLds + kN * 4;      // W stages, X rings, bias

// W [256][256] fp32 row-major -> [stage 16][part 3][cbi 8][lane 64][8 bf16]; stage = 2 kc + half;
// element j of lane l: k = 32 kc + 8 (l >> 4) + j, n = 16 (8 half + cbi) + (l & 15)
__global__ __launch_bounds__(256) void split_w_t16_kernel(const float *__restrict__ W, int64_t ldw,
                                                          uint16_t *__restrict__ img)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one (stage, cbi, lane)
    if (idx >= kT16Stages * 8 * 64) return;
    const int lane = idx & 63, cbi = (idx >> 6) & 7, stage = idx >> 9;
    const int n = 16 * (8 * (stage & 1) + cbi) + (lane & 15);
    const int k0 = 32 * (stage >> 1) + 8 * (lane >> 4);
    uint16_t h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(W[(int64_t)(k0 + j) * ldw + n], h[j], m[j], l[j]);
    uint16_t *base = img + (size_t)stage * (kT16StageBytes / 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        base[((0 * 8 + cbi) * 64 + lane) * 8 + j] = h[j];
        base[((1 * 8 + cbi) * 64 + lane) * 8 + j] = m[j];
        base[((2 * 8 + cbi) * 64 + lane) * 8 + j] = l[j];
    }
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
}

// (HBM -> LDS, 16 bytes per lane, from a UNIFORM base + a 32-bit lane offset + an immediate.  The
//  instruction's immediate offset is added to the global address AND to the LDS address: a piece at
//  +OFF of its source lands at +OFF of `lds_addr`)
template <int OFF> __device__ __forceinline__ void dma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr)
{
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr), "n"(OFF) : "memory", "m0");
}

template <int EPI, bool BITS = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_xw256_s16_kernel(
    const float *__restrict__ X, int64_t ldx, const unsigned char *__restrict__ ws, float *__restrict__ Y,
    int64_t ldy, int64_t M, uint32_t *__restrict__ y_absmax, const H2Epi ep)
{
    // EPI as in gemm_xw256_h2_kernel: 0 plain, 2 backward mask, 1 bias, 4 + ReLU, 5 + dropout at 1/2, 6 + dropout at p;
    // 3: backward mask from KEEP BITS (H2Epi::mask_bits: 8 bytes per lane and tile instead of 16 x 16)
    // BITS (with ReLU): the launch also writes `out > 0` as one bit per element (H2Epi::keep_bits_out)
    constexpr bool FWD_EPI = EPI == 1 || EPI >= 4, MASKED = EPI == 2, MASK_BITS = EPI == 3;
    static_assert(!BITS || EPI == 4 || EPI == 5, "keep bits belong to the ReLU epilogues");
    constexpr bool RELU = EPI >= 4, DROP1 = EPI == 5, DROP16 = EPI == 6;
    static_assert(kWaves == 8, "written for eight waves");
    constexpr int kSt = kS16StageBytes, kWShare = kSt / kWaves;          // a wave's part of a stage: 6 KiB
    extern __shared__ __attribute__((aligned(16))) unsigned char s16_lds[];   // [W stage 0][W stage 1][X rings][bias]
    unsigned char *lds = s16_lds;
    float *bias_lds = (float *)(s16_lds + 2 * kSt + kS16XLds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;
    const unsigned char *wl = ws + kH2HeaderBytes;
    const int64_t n_tiles = (M + kS16Rows - 1) / kS16Rows;
    const float *__restrict__ mask_src = ep.mask_src;
    const int64_t ld_mask = ep.ld_mask;
    const float mask_scale = ep.mask_scale;
    uint32_t seed_k0 = ep.seed_lo, seed_k1 = ep.seed_hi;
    if (FWD_EPI) {          // (bias into LDS, device seed read once: see gemm_xw256_h2_kernel)
        if (tid < kN) bias_lds[tid] = ep.bias != nullptr ? ep.bias[tid] : 0.f;
        if (ep.seed_dev != nullptr) {
            const uint64_t sd = *ep.seed_dev;
            seed_k0 = (uint32_t)sd;
            seed_k1 = (uint32_t)(sd >> 32);
        }
        __syncthreads();
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s16_lds;
    const uint32_t w_lds = lds0 + wave * kWShare;
    const uint32_t lane16 = (uint32_t)lane * 16u;
#ifndef S16_ABLATE        /* ablation builds only (results garbage): 1 no stores, 2 W staged during the first tile only, 4 X likewise */
#define S16_ABLATE 0
#endif
#ifndef S16_SPREAD       /* 0 (experiment builds): a stage's DMA and stores as a burst behind the barrier, its split at the end */
#define S16_SPREAD 1
#endif
    // this wave's eighth of W stage `st` (6 DMA instructions; the immediate moves BOTH ends, see dma16_s): pieces 2 pr, 2 pr + 1
    auto w_issue_pair = [&](int st, int b, int pr) {
        const unsigned char *src = wl + (size_t)st * kSt + wave * kWShare;
        const uint32_t dst = w_lds + b * kSt;
        if (pr == 0) { dma16_s<0>(src, lane16, dst); dma16_s<1024>(src, lane16, dst); }
        else if (pr == 1) { dma16_s<2048>(src, lane16, dst); dma16_s<3072>(src, lane16, dst); }
        else { dma16_s<0>(src + 4096, lane16, dst + 4096); dma16_s<1024>(src + 4096, lane16, dst + 4096); }
    };
    auto w_issue = [&](int st, int b) { w_issue_pair(st, b, 0); w_issue_pair(st, b, 1); w_issue_pair(st, b, 2); };
    unsigned char *xl = s16_lds + 2 * kSt + wave * (2 * kS16XChunk);
    const uint32_t x_lds = lds0 + 2 * kSt + wave * (2 * kS16XChunk);
    // what this lane FETCHES in DMA instruction j of a chunk: row 8j + ((lane >> 1) & 7) of the wave's 16,
    // columns 8 (lane >> 4) + 4 (lane & 1) .. + 3 of the chunk's 32 — as a byte offset from the tile origin
    const int ld_row = (lane >> 1) & 7;
    const uint32_t xoff0 = (uint32_t)(((16 * wave + ld_row) * ldx + 8 * (lane >> 4) + 4 * (lane & 1)) * 4);
    const uint32_t xoff1 = xoff0 + (uint32_t)(8 * ldx * 4);
    auto x_issue = [&](int64_t t, int chunk, int slot) {
        const float *src = X + t * kS16Rows * ldx + 32 * chunk;                         // (uniform)
        const int lim = (int)min((int64_t)kS16Rows, M - t * kS16Rows) - 16 * wave;      // rows of this wave that exist
        // (rows past the end of the matrix fetch the tile's first row instead: read, never stored)
        dma16_s<0>(src, ld_row < lim ? xoff0 : 0u, x_lds + slot * kS16XChunk);
        dma16_s<0>(src, ld_row + 8 < lim ? xoff1 : 0u, x_lds + slot * kS16XChunk + kS16XInstr);
    };
    // (where lane (n, q) READS its 32 bytes of a chunk)
    const int rd16 = (n16 >> 3) * kS16XInstr + 256 * q + 32 * (n16 & 7);
    u32x4 Xq[2][3];           // a chunk's fragment (parts h, m, l): stage c multiplies Xq[c & 1]
    auto split_chunk = [&](int slot, u32x4 (&Xp)[3]) {
        const unsigned char *p = xl + slot * kS16XChunk + rd16;
        const f32x4 lo = *(const f32x4 *)p, hi = *(const f32x4 *)(p + 16);
        const float av[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_pair(av[2 * j], av[2 * j + 1], h[j], m[j], l[j]);
        Xp[0] = u32x4{h[0], h[1], h[2], h[3]};
        Xp[1] = u32x4{m[0], m[1], m[2], m[3]};
        Xp[2] = u32x4{l[0], l[1], l[2], l[3]};
    };
    const int lrow = 16 * wave + n16;                                       // this lane's row in a tile
    const uint32_t yoff = (uint32_t)(lrow * ldy + 4 * q);                   // floats from the tile origin

    int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    [[maybe_unused]] unsigned long long st_begin = 0, st_bw = 0, st_n = 0, st_a = 0, st_b = 0, st_loop_end = 0;
    GEMM_STAMP(st_begin);
    x_issue(tile, 0, 0);
    w_issue(0, 0);
    x_issue(tile, 1, 1);
    dma_wait<2>();                 // X chunk 0 and this wave's part of W stage 0 are in LDS
    split_chunk(0, Xq[0]);
    uint32_t vmax = 0u;

    // ---- one column block of a finished tile: epilogue + one 16-byte store per lane
    f32x4 prev[16];               // the previous tile's results, leaving two column blocks per stage
    int64_t ptile = 0;
    bool have_prev = false;
    [[maybe_unused]] uint32_t r1[4] = {0u, 0u, 0u, 0u};          // dropout at 1/2: the row's keep bits (one Philox call)
    [[maybe_unused]] uint32_t kb[2] = {0u, 0u};                  // forward: this lane's 64 result bits of the tile being stored
    // backward mask from bits: the 8 bytes of this lane for the tile being STORED (mb_prev), being multiplied
    // (mb_cur) and the one after (mb_next, requested at the top of a tile: a whole tile ahead of its use)
    [[maybe_unused]] uint32_t mb_prev[2] = {0u, 0u}, mb_cur[2] = {0u, 0u}, mb_next[2] = {0u, 0u};
    [[maybe_unused]] int32_t mrow_next = 0;                      // the mask row of this lane's row in the NEXT tile
    auto mask_row_of = [&](int64_t t) -> int32_t {               // (rows past the end: row 0 of the mask, never used)
        const int64_t row = t * kS16Rows + lrow;
        return (t >= n_tiles || row >= M) ? 0 : (ep.mask_rows != nullptr ? ep.mask_rows[row] : (int32_t)row);
    };
    auto bits_load = [&](int32_t mrow, uint32_t (&dst)[2]) {
        const uint2 w = *(const uint2 *)(ep.mask_bits + (int64_t)mrow * 8 + 2 * q);
        dst[0] = w.x;
        dst[1] = w.y;
    };
    auto bits_store = [&](int64_t t) {                           // after the tile's 16th column block
        if (BITS) *(uint2 *)(ep.keep_bits_out + (t * kS16Rows + lrow) * 8 + 2 * q) = uint2{kb[0], kb[1]};
    };
    [[maybe_unused]] int64_t pmask_row = 0;                      // masked form: the mask row of this lane's prev row
    auto finish = [&](int cb, f32x4 v, const f32x4 &mk, int64_t drow) __attribute__((always_inline)) -> f32x4 {
        const int f = 16 * cb + 4 * q;                           // first of this lane's 4 columns
        if (FWD_EPI) {
            const f32x4 b4 = *(const f32x4 *)(bias_lds + f);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        if (RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (DROP1) {
            // ONE Philox call per row and parity of bit 2 of the column covers this lane's 64 columns: column
            // 16cb + 4q + j is bit 4 (2cb + (q >> 1)) + j of the call with block q & 1 (gcn_spmm.hip apply_dropout)
            const uint32_t nib = r1[cb >> 2] >> (8 * (cb & 3) + 4 * (q >> 1));
            v.x = (nib & 1u) ? v.x * ep.drop_scale : 0.f;
            v.y = (nib & 2u) ? v.y * ep.drop_scale : 0.f;
            v.z = (nib & 4u) ? v.z * ep.drop_scale : 0.f;
            v.w = (nib & 8u) ? v.w * ep.drop_scale : 0.f;
        } else if (DROP16) {
            // eight 16-bit fields per call: block = (cb << 1) | (q & 1), this lane's four columns are fields 4 (q >> 1) .. + 3
            uint32_t r4[4];
            uint32_t cw = ((uint32_t)cb << 1) | (uint32_t)(q & 1);
            asm volatile("" : "+v"(cw));
            h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, seed_k0, seed_k1, r4);
            const uint32_t w0 = (q & 2) ? r4[2] : r4[0], w1 = (q & 2) ? r4[3] : r4[1];
            v.x = (w0 & 0xFFFFu) >= ep.drop_thresh ? v.x * ep.drop_scale : 0.f;
            v.y = (w0 >> 16) >= ep.drop_thresh ? v.y * ep.drop_scale : 0.f;
            v.z = (w1 & 0xFFFFu) >= ep.drop_thresh ? v.z * ep.drop_scale : 0.f;
            v.w = (w1 >> 16) >= ep.drop_thresh ? v.w * ep.drop_scale : 0.f;
        }
        if (MASKED) {
            v.x = mk.x > 0.f ? v.x * mask_scale : 0.f;
            v.y = mk.y > 0.f ? v.y * mask_scale : 0.f;
            v.z = mk.z > 0.f ? v.z * mask_scale : 0.f;
            v.w = mk.w > 0.f ? v.w * mask_scale : 0.f;
        }
        if (MASK_BITS) {
            const uint32_t nib = mb_prev[cb >> 3] >> (4 * (cb & 7));
            v.x = (nib & 1u) ? v.x * mask_scale : 0.f;
            v.y = (nib & 2u) ? v.y * mask_scale : 0.f;
            v.z = (nib & 4u) ? v.z * mask_scale : 0.f;
            v.w = (nib & 8u) ? v.w * mask_scale : 0.f;
        }
        if (BITS) {                                                // what the backward mask will ask: out > 0
            // (out is max(., 0) times a positive scale or 0: never NaN — so out > 0 <=> its bits, as a SIGNED integer,
            //  are >= 1: one v_med3_i32 per element — hipcc makes a compare + select + wait states of the C form)
            auto pos = [](float f) {
                uint32_t r;
                asm("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(__float_as_uint(f)));
                return r;
            };
            const uint32_t nib = pos(v.x) | (pos(v.y) << 1) | (pos(v.z) << 2) | (pos(v.w) << 3);
            kb[cb >> 3] = (cb & 7) == 0 ? nib : (kb[cb >> 3] | (nib << (4 * (cb & 7))));
        }
        if (y_absmax != nullptr) {                                 // (wave-uniform)
            if (RELU)
                vmax = max(max(vmax, max(__float_as_uint(v.x), __float_as_uint(v.y))),
                           max(__float_as_uint(v.z), __float_as_uint(v.w)));
            else
                vmax = max(max(vmax, max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu)),
                           max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
        }
        return v;
    };
    // what is needed once per finished tile: the row's dropout bits / its mask row
    auto capture = [&](int64_t t) {
        const int64_t row = t * kS16Rows + lrow;
        if (DROP1) {
            const int64_t drow = row + ep.drop_row_base;
            uint32_t cw = (uint32_t)(q & 1);
            asm volatile("" : "+v"(cw));
            h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, seed_k0, seed_k1, r1);
        }
        if (MASKED) pmask_row = row >= M ? 0 : (ep.mask_rows != nullptr ? (int64_t)ep.mask_rows[row] : row);
    };
    [[maybe_unused]] f32x4 mk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // masked form: the next pair's masks
    auto mask_load = [&](int pair) {
        const float *mrow = mask_src + pmask_row * ld_mask + 4 * q + 32 * pair;
        mk[0] = *(const f32x4 *)mrow;
        mk[1] = *(const f32x4 *)(mrow + 16);
    };

    if (MASK_BITS) {                                           // (once: a dependent pair of loads)
        bits_load(mask_row_of(tile), mb_cur);
        mrow_next = mask_row_of(tile + gridDim.x);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < n_tiles;      // (uniform)
        asm volatile("" : "+s"(wl));                           // (no hoisting of the W image's addresses)
        f32x4 acc[16];
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < kS16Stages; ++c) {
            // every wave waited for ITS part of stage c before it got here and has read its last
            // fragment of stage c - 1
            GEMM_STAMP(st_a);
            __builtin_amdgcn_s_barrier();
            GEMM_STAMP(st_b);
            st_bw += st_b - st_a;
            if (MASK_BITS && c == 0) {
                // the NEXT tile's bits (its mask row was looked up a tile ago) and the mask row of the one after
                bits_load(mrow_next, mb_next);
                mrow_next = mask_row_of(tile + 2 * (int64_t)gridDim.x);
            }
            auto store_one = [&](int u) __attribute__((always_inline)) {  // column block 2c + u of the previous tile (always a full tile)
                if (have_prev) {
                    float *ybase = Y + ptile * kS16Rows * ldy;                          // (uniform)
                    const int64_t drow = ptile * kS16Rows + lrow + ep.drop_row_base;
                    const int cb = 2 * c + u;
                    const f32x4 v = finish(cb, prev[cb], mk[u], drow);
                    if (!(S16_ABLATE & 1) || v.x == 1.2345e-30f) *(f32x4 *)(ybase + yoff + 16 * cb) = v;
                    if (cb == 15) bits_store(ptile);
                }
            };
            auto stores = [&]() __attribute__((always_inline)) { store_one(0); store_one(1); };
            // masked form: the masks were loaded a stage ago, and the compiler's wait for them must not
            // see this stage's DMA (it cannot count inline-assembly loads): stores first, then the next
            // pair's mask loads, then the DMA.  Other forms: DMA first, the stores are the youngest.
            if (MASKED) {
                stores();
                if (c + 1 < kS16Stages) { if (have_prev) mask_load(c + 1); }
            }
            const bool issued_w = (c + 1 < kS16Stages) || has_next;
            const bool issued_x = (c + 2 < kS16Stages) || has_next;
            constexpr bool SPREAD = S16_SPREAD && !MASKED;          // (MASK_BITS loads nothing in its stages: it spreads)
            auto w_issue_all = [&]() {
                if (c + 1 < kS16Stages) w_issue(c + 1, (c + 1) & 1);
                else if (has_next) w_issue(0, 0);                // the next tile's first stage
            };
            const bool first_tile = tile == (int64_t)blockIdx.x;
            auto issue_w_pair = [&](int pr) {
                if ((S16_ABLATE & 2) && !first_tile) return;
                if (c + 1 < kS16Stages) w_issue_pair(c + 1, (c + 1) & 1, pr);
                else if (has_next) w_issue_pair(0, 0, pr);
            };
            auto issue_x = [&]() {
                if ((S16_ABLATE & 4) && !first_tile) return;
                if (c + 2 < kS16Stages)       // chunk c + 2 into the ring slot chunk c left (split a stage ago)
                    x_issue(tile, c + 2, c & 1);
                else if (has_next)
                    x_issue(tile + gridDim.x, c + 2 - kS16Stages, c & 1);
            };
            if (!SPREAD) {
                w_issue_all();
                issue_x();
                if (!MASKED) stores();
            }
            const unsigned char *buf = lds + (c & 1) * kSt;
            const u32x4 (&Xp)[3] = Xq[c & 1];
            u32x4 Bf[2][3];
            auto b_read = [&](int cb, u32x4 (&dst)[3]) {
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    dst[sp] = *(const u32x4 *)(buf + (cb >> 3) * kT16StageBytes + ((sp * 8 + (cb & 7)) * 64 + lane) * 16);
            };
            b_read(0, Bf[0]);
#pragma unroll
            for (int cb = 0; cb < 16; ++cb) {
                if (cb + 1 < 16) b_read(cb + 1, Bf[(cb + 1) & 1]);
                const u32x4 Bh = Bf[cb & 1][0], Bm = Bf[cb & 1][1], Bl = Bf[cb & 1][2];
                __builtin_amdgcn_s_setprio(1);
                f32x4 t = acc[cb];
                t = mfma16(Bh, Xp[2], t);      // smallest terms first
                t = mfma16(Bl, Xp[0], t);
                t = mfma16(Bm, Xp[1], t);
                t = mfma16(Bh, Xp[1], t);
                t = mfma16(Bm, Xp[0], t);
                t = mfma16(Bh, Xp[0], t);
                acc[cb] = t;
                __builtin_amdgcn_s_setprio(0);
                // SPREAD: nothing is issued as a burst.  Behind column block 0 / 2 / 4: two W pieces each;
                // 6: the X chunk; 8, 10: one store each; 12: the next chunk (issued a stage ago) is waited
                // for and split into the other fragment set, under the last column blocks' MFMAs.
#ifndef S16_SCHED         /* experiment builds: other placements of the same pieces (all issue in the same ORDER) */
#define S16_SCHED 0
#endif
                // (column block behind which go: W pairs 0-2, the X chunk, store 0, store 1, the split)
                constexpr int at[7] = {S16_SCHED == 0 ? 0 : S16_SCHED == 1 ? 1 : S16_SCHED == 2 ? 0 : 2,
                                       S16_SCHED == 0 ? 2 : S16_SCHED == 1 ? 3 : S16_SCHED == 2 ? 1 : 4,
                                       S16_SCHED == 0 ? 4 : S16_SCHED == 1 ? 5 : S16_SCHED == 2 ? 2 : 6,
                                       S16_SCHED == 0 ? 6 : S16_SCHED == 1 ? 7 : S16_SCHED == 2 ? 4 : 8,
                                       S16_SCHED == 0 ? 8 : S16_SCHED == 1 ? 9 : S16_SCHED == 2 ? 7 : 10,
                                       S16_SCHED == 0 ? 10 : S16_SCHED == 1 ? 11 : S16_SCHED == 2 ? 10 : 12,
                                       S16_SCHED == 0 ? 12 : S16_SCHED == 1 ? 13 : S16_SCHED == 2 ? 13 : 14};
                if (SPREAD) {
                    if (cb == at[0]) issue_w_pair(0);
                    if (cb == at[1]) issue_w_pair(1);
                    if (cb == at[2]) issue_w_pair(2);
                    if (cb == at[3]) issue_x();
                    if (cb == at[4]) store_one(0);
                    if (cb == at[5]) store_one(1);
                    if (cb == at[6] && issued_w) {
                        // younger than the chunk's two DMA instructions (at least): this stage's 6 W pieces,
                        // its X chunk and its stores
                        if (issued_x) { if (have_prev) dma_wait<10>(); else dma_wait<8>(); }
                        else { if (have_prev) dma_wait<8>(); else dma_wait<6>(); }
                        split_chunk((c + 1) & 1, Xq[(c + 1) & 1]);
                    }
                }
            }
            // before the next stage: this wave's part of stage c + 1 and the next chunk (issued a stage
            // ago) must be in LDS; younger than both: the X chunk issued in THIS stage and (unmasked
            // forms) this stage's two stores
            if (issued_w) {
                if (!MASKED && have_prev) {
                    if (issued_x) dma_wait<4>();
                    else dma_wait<2>();
                } else {
                    if (issued_x) dma_wait<2>();
                    else dma_wait<0>();
                }
                if (!SPREAD) split_chunk((c + 1) & 1, Xq[(c + 1) & 1]);
                asm volatile("" : "+v"(Xq[(c + 1) & 1][0]), "+v"(Xq[(c + 1) & 1][1]), "+v"(Xq[(c + 1) & 1][2]));
            }
        }
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) prev[cb] = acc[cb];
        ptile = tile;
        have_prev = true;
        st_n += 1;
        if (MASK_BITS) {
            mb_prev[0] = mb_cur[0]; mb_prev[1] = mb_cur[1];
            mb_cur[0] = mb_next[0]; mb_cur[1] = mb_next[1];
        }
        capture(tile);
        if (MASKED && has_next) mask_load(0);
    }

    GEMM_STAMP(st_loop_end);
    // ---- the last tile of this workgroup (the only one that can be partial): a plain store section
    if (ptile * kS16Rows + lrow < M) {
        float *ybase = Y + ptile * kS16Rows * ldy;
        const int64_t drow = ptile * kS16Rows + lrow + ep.drop_row_base;
#pragma unroll
        for (int pair = 0; pair < 8; ++pair) {
            if (MASKED) mask_load(pair);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cb = 2 * pair + u;
                *(f32x4 *)(ybase + yoff + 16 * cb) = finish(cb, prev[cb], mk[u], drow);
            }
            if (DROP16) __builtin_amdgcn_sched_barrier(0);
        }
        bits_store(ptile);
    }
    if (y_absmax != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, off, 64));
        if (lane == 0 && vmax != 0u) atomicMax(y_absmax, vmax);
    }
#ifdef GEMM_PROFILE_STAMPS
    if (lane == 0) {           // per wave: cycles waiting at the stage barriers / in the final store section / tiles / lifetime
        unsigned long long st_end;
        GEMM_STAMP(st_end);
        atomicAdd(&g_gemm_stamps[0], st_bw);
        atomicAdd(&g_gemm_stamps[1], st_end - st_loop_end);
        atomicAdd(&g_gemm_stamps[2], st_n);
        atomicAdd(&g_gemm_stamps[3], st_end - st_begin);
        atomicAdd(&g_gemm_stamps[4], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// bf16 storage (config C5): Y[M,N] = X[M,K] · W[K,N], bf16 in / out, fp32 accumulate, (K, N) in
// {(128,128), (128,256), (256,128)} (256 x 256 would need more registers than two waves per SIMD
// leave; it stays on hipBLASLt).  One bf16 MFMA per product — at 2·M·K·N flops against (K + N)·2 bytes per row this
// kernel is bound by HBM, not by the matrix pipe (C5: 1.6 TFLOP vs 25.6 GB), so it is built as a
// STREAM: the whole W (32–128 KiB, fragment-ordered by the prep kernel) is resident in LDS for the
// lifetime of the workgroup, every wave walks its own 32-row tiles (grid-stride, no barrier after
// the W load), the X fragments of the NEXT tile are in flight while the current one is multiplied
// and stored, and the bf16 output rows leave as 16-byte stores (the two half-waves exchange their
// 4-column groups with v_permlane32_swap first).
// EPI: 0 plain product, 1 forward epilogue (bias / ReLU / dropout), 2 backward mask read at the
// output row, 3 backward mask read through a row list.
//
// NO DATA-DEPENDENT BRANCH AROUND A MEMORY INSTRUCTION inside the tile loop: the X fragments and the
// Y rows move through BUFFER instructions whose descriptor covers exactly the rows of the tile that
// exist (hardware range check: reads past the end give zeros, writes are dropped; a tile past the
// last one has zero records), and the loop runs tiles in pairs with the two fragment register sets
// swapped instead of copied.  Both matter for the prefetch: hipcc's wait-count pass merges
// the pending-operation state of the branches of an `if` pessimistically, and placed each copy
// `a_cur[c] = a_nxt[c]` right after the last MFMA reading a_cur[c] — either way the wave waited
// (`s_waitcnt vmcnt(7 - c)`) on loads issued a few instructions earlier and the "prefetch" hid
// nothing (0.40 of HBM at C5).  The mask of the backward form is requested BEFORE the next tile's
// fragments (loads retire in order: a wait on the mask must not drain the prefetch), its row-list
// entry one tile ahead.
template <int K, int N, int EPI>
__global__ __launch_bounds__(256, ((K <= 128 && N <= 128) ? ((EPI == 2 || EPI == 3) ? 2 : 3) : 2)) void gemm_bf16_kernel(
    const uint16_t *__restrict__ X, int64_t ldx, const uint16_t *__restrict__ wimg,
    uint16_t *__restrict__ Y, int64_t ldy, int64_t M, int64_t n_tiles, const H2Epi ep)
{
    constexpr int KC = K / 16, NB = N / 32;
    // EPI: 0 plain, 2 / 3 backward mask (own row / row list); forward epilogues as in
    // gemm_xw256_h2_kernel: 1 bias, 4 + ReLU, 5 + dropout at p = 1/2, 6 + dropout at another p
    constexpr bool FWD_EPI = EPI == 1 || EPI >= 4, MASK = EPI == 2 || EPI == 3;
    constexpr bool RELU = EPI >= 4, DROP1 = EPI == 5, DROP16 = EPI == 6;
    // the whole tile's mask (NB x 32 bytes per lane) is requested ahead of the prefetch where the
    // registers allow it — the 128 x 128 shape of config C5; the wider shapes fetch it per column
    // block in the store section (a wait on such a load drains the prefetch, but nothing spills)
    constexpr bool MASK_EARLY = MASK && K <= 128 && N <= 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];    // K * N * 2 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const u32x4 *src = (const u32x4 *)wimg;
        u32x4 *dst = (u32x4 *)wlds;
        for (int i = tid; i < K * N * 2 / 16; i += 256) dst[i] = src[i];
    }
    // (forward epilogue: bias into LDS, device-resident seed read once — in the store section they
    //  would be vector loads waited for with vmcnt(0), see gemm_xw256_h2_kernel)
    __shared__ __attribute__((aligned(16))) float bias_lds[FWD_EPI ? N : 4];
    uint32_t seed_k0 = ep.seed_lo, seed_k1 = ep.seed_hi;
    if (FWD_EPI) {
        if (tid < N) bias_lds[tid] = ep.bias != nullptr ? ep.bias[tid] : 0.f;
        if (ep.seed_dev != nullptr) {
            const uint64_t sd = *ep.seed_dev;
            seed_k0 = (uint32_t)sd;
            seed_k1 = (uint32_t)(sd >> 32);
        }
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;                  // (wave-uniform: scalar registers)
    if (t >= n_tiles) return;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t ldx_b = (uint32_t)ldx * 2u, ldy_b = (uint32_t)ldy * 2u;   // (host: 32 rows < 2 GiB)
    // descriptor of the rows of tile `tile` that exist, `width_b` bytes used per row
    auto tile_rsrc = [&](const uint16_t *base, int64_t ld, uint32_t ld_b, int64_t tile, uint32_t width_b) {
        const int64_t left = M - tile * 32;
        const uint32_t rows = left >= 32 ? 32u : (left > 0 ? (uint32_t)left : 0u);
        const uint64_t p = (uint64_t)(base + (rows ? tile : 0) * 32 * ld);
        // (all wave-uniform; said so explicitly — a descriptor word the compiler keeps in a vector
        //  register turns every access into a per-lane "waterfall" loop)
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        const uint32_t nrec = __builtin_amdgcn_readfirstlane(rows ? (rows - 1u) * ld_b + width_b : 0u);
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, nrec, 0x00020000);
    };
    auto fetch = [&](int64_t tile, u32x4 (&dst)[KC]) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(X, ldx, ldx_b, tile, K * 2);
        const uint32_t voff = (uint32_t)r * ldx_b + 16u * h;
#pragma unroll
        for (int c = 0; c < KC; ++c)                             // k = 16c + 8h .. +8
            dst[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 32 * c, 0, 0));
    };
    // (mask variants) the row of the mask that belongs to output row `row` of tile `tile`
    // (EPI 3: the raw 32-bit list entry — widened where it is used, one tile later; a conversion
    //  at the load would be scheduled right behind it and wait for it there)
    auto mask_index = [&](int64_t tile) -> int32_t {
        int64_t row = tile * 32 + r;
        row = row < M ? row : M - 1;                             // (clamped: the store is dropped)
        return EPI == 3 ? ep.mask_rows[row] : 0;
    };
    u32x4 a0[KC], a1[KC];
    int32_t m0 = 0, m1 = 0;
    auto one_tile = [&](const int64_t t, u32x4 (&a_cur)[KC], u32x4 (&a_nxt)[KC], const int32_t m_cur, int32_t &m_nxt) {
        // this lane's 8 stored columns of every (nb, g pair): 16 bytes of the mask each
        u32x4 mk[MASK_EARLY ? NB : 1][2];
        const unsigned char *mrow_late = nullptr;
        if (MASK && !MASK_EARLY) {
            int32_t mc = m_cur;
            asm volatile("" : "+v"(mc));
            int64_t mi = t * 32 + r;
            mi = EPI == 3 ? (int64_t)mc : (mi < M ? mi : M - 1);
            mrow_late = (const unsigned char *)((const uint16_t *)ep.mask_src + mi * ep.ld_mask) + 16 * h;
        }
        if (MASK_EARLY) {
            // (scheduling fences around the mask requests: nothing of this tile — the widening of
            //  m_cur, the swaps of the mask words in the store section — may move to where it would
            //  wait on a load that is younger than the prefetch)
            __builtin_amdgcn_sched_barrier(0);
            int32_t mc = m_cur;
            asm volatile("" : "+v"(mc));         // (the list entry is first LOOKED AT here, one tile after its load)
            int64_t mi = t * 32 + r;
            mi = EPI == 3 ? (int64_t)mc : (mi < M ? mi : M - 1);
            const unsigned char *mrow = (const unsigned char *)((const uint16_t *)ep.mask_src + mi * ep.ld_mask) + 16 * h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 2; ++q) mk[nb][q] = *(const u32x4 *)(mrow + (32 * nb + 16 * q) * 2);
            // (the order matters and the scheduler does not know it: loads retire in order, so a
            //  mask requested AFTER the prefetch could only be waited for by draining the prefetch)
            __builtin_amdgcn_sched_barrier(0);
        }
        fetch(t + stride, a_nxt);
        if (MASK) m_nxt = mask_index(t + stride < n_tiles ? t + stride : t);
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
        // (opaque per tile: otherwise hipcc hoists every W fragment out of the tile loop into
        //  registers — 128..512 VGPRs — and the kernel runs at one wave per SIMD)
        //  (an opaque OFFSET: an opaque pointer would leave the LDS address space and turn the
        //  reads into flat loads)
        uint32_t wofs = 0;
        asm volatile("" : "+v"(wofs));
        const unsigned char *wl = wlds + wofs;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const u32x4 b = *(const u32x4 *)(wl + ((c * NB + nb) * 64 + lane) * 16);
                acc[nb] = mfma(b, a_cur[c], acc[nb]);      // transposed tile: lane = output row
            }
        }
        if (MASK_EARLY) __builtin_amdgcn_sched_barrier(0);
        const int64_t row = t * 32 + r;
        uint32_t r1[4] = {0u, 0u, 0u, 0u};                                   // p = 1/2: one-bit fields
        if (DROP1) {
            const int64_t drow = row + ep.drop_row_base;
            uint32_t cw = (uint32_t)h;
            asm volatile("" : "+v"(cw));
            h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, seed_k0, seed_k1, r1);
        }
        const __amdgpu_buffer_rsrc_t ys = tile_rsrc(Y, ldy, ldy_b, t, N * 2);
        const uint32_t yoff = (uint32_t)r * ldy_b + 16u * h;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                if (FWD_EPI) {
                    // the layer's forward epilogue on the fp32 accumulators (the layer evaluated
                    // as (Â·X)·W + b, its last stage being this GEMM): bias, ReLU, Philox
                    // dropout — the keep function of gcn_spmm.hip, one Philox call per 8 columns
                    // (at p = 1/2: one call per lane and tile, r1 above)
                    uint32_t r4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int gg = g; gg < g + 2; ++gg) {
                        float v[4] = {acc[nb][4 * gg], acc[nb][4 * gg + 1], acc[nb][4 * gg + 2],
                                      acc[nb][4 * gg + 3]};
                        {
                            const f32x4 b4 = *(const f32x4 *)(bias_lds + 32 * nb + 8 * gg + 4 * h);
                            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                        }
                        if (RELU) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                        }
                        if (DROP1) {
                            const uint32_t nib = r1[nb >> 1] >> (16 * (nb & 1) + 4 * gg);
                            v[0] = (nib & 1u) ? v[0] * ep.drop_scale : 0.f;
                            v[1] = (nib & 2u) ? v[1] * ep.drop_scale : 0.f;
                            v[2] = (nib & 4u) ? v[2] * ep.drop_scale : 0.f;
                            v[3] = (nib & 8u) ? v[3] * ep.drop_scale : 0.f;
                        } else if (DROP16) {
                            if (gg == g) {       // one call for the pair of groups g, g + 1
                                const uint32_t k0 = seed_k0, k1 = seed_k1;
                                uint32_t cw = ((uint32_t)(2 * nb + (g >> 1)) << 1) | (uint32_t)h;
                                asm volatile("" : "+v"(cw));       // (no hoisting of the first round)
                                const int64_t drow = row + ep.drop_row_base;
                                h2_philox((uint32_t)drow, (uint32_t)(drow >> 32), cw, 0u, k0, k1, r4);
                            }
                            const uint32_t w0 = (gg & 1) ? r4[2] : r4[0], w1 = (gg & 1) ? r4[3] : r4[1];
                            v[0] = (w0 & 0xFFFFu) >= ep.drop_thresh ? v[0] * ep.drop_scale : 0.f;
                            v[1] = (w0 >> 16) >= ep.drop_thresh ? v[1] * ep.drop_scale : 0.f;
                            v[2] = (w1 & 0xFFFFu) >= ep.drop_thresh ? v[2] * ep.drop_scale : 0.f;
                            v[3] = (w1 >> 16) >= ep.drop_thresh ? v[3] * ep.drop_scale : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nb][4 * gg + j] = v[j];
                    }
                }
                // groups k = 4nb + g and k + 1: this lane's 4 columns of each, packed to bf16
                f32x2 p0 = {acc[nb][4 * g], acc[nb][4 * g + 1]}, p1 = {acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                f32x2 q0 = {acc[nb][4 * g + 4], acc[nb][4 * g + 5]}, q1 = {acc[nb][4 * g + 6], acc[nb][4 * g + 7]};
                if (MASK) {
                    // backward of a fused ReLU / dropout epilogue in the grad_input GEMM's own
                    // store: y = mask > 0 ? y * scale : 0 on the fp32 accumulators, before the
                    // rounding.  The mask words were fetched in STORE layout (8 consecutive columns
                    // per lane); the accumulators are still in MFMA layout (4 + 4 columns, the
                    // other half-wave holding the columns in between), so the mask words take the
                    // inverse of the swap the packed results take below.
                    const u32x4 m = MASK_EARLY ? mk[MASK_EARLY ? nb : 0][g >> 1]
                                               : *(const u32x4 *)(mrow_late + (32 * nb + 8 * g) * 2);
                    auto ux = __builtin_amdgcn_permlane32_swap(m[0], m[2], false, false);
                    auto uy = __builtin_amdgcn_permlane32_swap(m[1], m[3], false, false);
                    const uint32_t mw[4] = {ux[0], uy[0], ux[1], uy[1]};     // columns of p0, p1, q0, q1
                    auto keep = [&](f32x2 &v, uint32_t w) {
                        v.x = __uint_as_float(w << 16) > 0.f ? v.x * ep.mask_scale : 0.f;
                        v.y = __uint_as_float(w & 0xffff0000u) > 0.f ? v.y * ep.mask_scale : 0.f;
                    };
                    keep(p0, mw[0]);
                    keep(p1, mw[1]);
                    keep(q0, mw[2]);
                    keep(q1, mw[3]);
                }
                uint32_t ax = __builtin_bit_cast(uint32_t, __builtin_convertvector(p0, bf16x2));
                uint32_t ay = __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, bf16x2));
                uint32_t bx = __builtin_bit_cast(uint32_t, __builtin_convertvector(q0, bf16x2));
                uint32_t by = __builtin_bit_cast(uint32_t, __builtin_convertvector(q1, bf16x2));
                auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                // lower half-wave: columns 8k .. 8k+7, upper: 8k+8 .. 8k+15 (16 bytes each)
                const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v),
                                                       ys, yoff + (32 * nb + 8 * g) * 2, 0, 0);
            }
        }
    };
    fetch(t, a0);
    if (MASK) m0 = mask_index(t);
    {
        // NB x 2 stores into a zero-record descriptor (dropped by the range check): the loop's first
        // pass then sees the same sequence of pending memory operations as every later one
        // (fragments, a tile's stores, next fragments), and the counted waits hipcc places in the
        // loop are those of the steady state instead of the minimum over the loop's two entries —
        // which made every tile wait for the previous tile's stores before its first MFMA
        const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, 0, 0x00020000);
        const __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NB * 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, none, 16 * i, 0, 0);
    }
    for (; t < n_tiles; t += 2 * stride) {
        one_tile(t, a0, a1, m0, m1);
        one_tile(t + stride, a1, a0, m1, m0);      // (past the last tile: zero records, a no-op)
    }
}

// W [K][N] bf16 row-major -> [chunk K/16][colblock N/32][lane 64][8 bf16]; element j of lane l:
// k = 16*chunk + 8*(l>>5) + j, n = 32*cb + (l&31)
__global__ __launch_bounds__(256) void order_w_bf16_kernel(const uint16_t *__restrict__ W, int64_t ldw,
                                                           uint16_t *__restrict__ img, int K, int N)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb_count = N / 32;
    if (idx >= (K / 16) * nb_count * 64) return;
    const int lane = idx & 63, cb = (idx >> 6) % nb_count, chunk = (idx >> 6) / nb_count;
    const int n = 32 * cb + (lane & 31), k0 = 16 * chunk + 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) img[(size_t)idx * 8 + j] = W[(int64_t)(k0 + j) * ldw + n];
}


// ------------------------------------------------------------------------------------------------
// Weight gradient: grad_W[256, 256] = Σ_r A[ra[r], :]ᵀ ⊗ G[rg[r], :]  over a LIST of rows
// (`inputᵀ · grad_support`, the backward of pygcn/layers.py:33), fp32 in / out, scaled two-part
// fp16 scheme (three MFMAs per product) like gemm_xw256_h2_kernel.
//
// The reduction runs over the graph's vertices, so both operands are "k-major" for the MFMA: a
// lane needs 8 consecutive ROWS of one column.  No transpose through LDS is needed: a wave loads
// ONE WHOLE ROW per instruction — 256 floats = 1 KiB, 16 bytes per lane, row address on the scalar
// unit, which makes the row GATHER free (ra / rg list the rows on which the gradient can be
// non-zero, pygcn_amd/fused.py) — and 8 consecutive list entries per SUPER-STEP of 32 rows: lane l
// then holds k = 0..7 of its four columns 4l..4l+3, i.e. one complete 16-byte MFMA fragment for
// each of four 32-column tiles
//         tile T = 2q + (l >> 5), fragment lane (l & 31) + 32·(k-half),  column 128·(T & 1) + 4·(l & 31) + (T >> 1)
// — a permutation of the columns that only decides where a result element is stored.  Waves 0-3
// load, split into (h, m) fp16 parts and publish the A fragments (8 rows of the super-step each),
// waves 4-7 the G fragments (LDS, double-buffered, ONE barrier per 32 rows); every wave then
// multiplies its 64 x 128 block of the result (2 x 4 tiles) over the super-step's two MFMA steps.
// (Round 3, first form: 64 columns of one row per instruction, 4 bytes per lane.  The address unit
//  takes as long for such a wave instruction as for a 16-byte one, all waves issue their loads at
//  the same point of the step, and the ablation builds showed load phase and arithmetic adding up
//  instead of overlapping: 5.5 ms = 2.9 ms without loads + 2.6 ms — tools/atg_variant_sweep.py.)
// Loads run kAtgDepth super-steps ahead in a register ring.  The row list is cut into slabs, one
// per workgroup; the slabs' partial products are added in slab order by a second kernel
// (deterministic, no atomics).  Index lists are padded to a multiple of 16 entries with valid
// indices (n_list counts the real entries): the 8 indices of a wave's share are one scalar load.
constexpr int kAtgSuperMin = 4;                 // at least this many 32-row super-steps per workgroup
constexpr int kAtgMaxWgs = 256;
constexpr int kAtgDepth = 2;
constexpr int kAtgBufBytes = 2 * 2 * 8 * 2 * kFragBytes;   // [operand][MFMA step][tile][split] = 64 KiB (h2)
// SCH (SchemeK): 0 two scaled fp16 parts, two LDS buffers (one barrier per super-step); 1 three bf16
// parts — the fp32-equivalent form, no bounds: its 96 KiB of fragments per super-step exist ONCE
// (two buffers would not fit the LDS), so a second barrier separates multiply and publish.
template <int SCH> constexpr int atg_buf_bytes() { return 2 * 2 * 8 * SchemeK<SCH>::NS * kFragBytes; }
#ifndef ATG_B3_STEP16
#define ATG_B3_STEP16 1
#endif
template <int SCH> constexpr int atg_lds_bytes() { return SCH == 0 ? 2 * atg_buf_bytes<SCH>() : (ATG_B3_STEP16 ? 3 * atg_buf_bytes<SCH>() / 2 : atg_buf_bytes<SCH>()); }

template <int SCH>
__global__ __launch_bounds__(512, 2) void gemm_atg256_h2_kernel(
    const float *__restrict__ A, int64_t lda, const int32_t *__restrict__ ra,
    const float *__restrict__ G, int64_t ldg, const int32_t *__restrict__ rg, int64_t n_list,
    const float *__restrict__ a_bound, const float *__restrict__ g_bound,
    float *__restrict__ partial, int64_t supers_per_wg, float *__restrict__ cs_partial)
{
    // cs_partial (three-part ring form only, else NULL): [workgroups][256] column sums of the listed G rows —
    // the bias gradient of the layer whose weight gradient this is, from the rows the kernel loads anyway
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // 2 buffers of kAtgBufBytes
    const int tid = threadIdx.x, lane = tid & 63;
    [[maybe_unused]] unsigned long long st_begin = 0;
    GEMM_STAMP(st_begin);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int op = wave >> 2, w8 = wave & 3;            // operand streamed (0 A, 1 G); rows 8·w8 .. +7 of a super-step
    const int iw = wave & 3, jh = wave >> 2;            // result block: A tiles 2iw.., G tiles 4jh..
    auto scale_exp = [](float b) {
        int e = 14 - floor_log2f(b);
        e = e > 126 ? 126 : (e < -126 ? -126 : e);
        return (!(b > 0.f) || !(b <= 3.4028235e38f)) ? 0 : e;
    };
    constexpr int NS = SchemeK<SCH>::NS;
    float my_scale = 1.f, back_a = 1.f, back_b = 1.f;
    if constexpr (SCH == 0) {
        const int a_exp = scale_exp(*a_bound), g_exp = scale_exp(*g_bound);
        my_scale = pow2f(op ? g_exp : a_exp);
        const int back = -(a_exp + g_exp);
        // (an inf / NaN bound = an overflow upstream: poison the result instead of scaling by 1)
        const bool poisoned = !(*a_bound <= 3.4028235e38f) || !(*g_bound <= 3.4028235e38f);
        back_a = poisoned ? __uint_as_float(0x7fc00000u) : pow2f(back / 2);
        back_b = pow2f(back - back / 2);
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ib][jb][i] = 0.f;

    const int64_t total_supers = (n_list + 31) >> 5;
    const int64_t s0 = (int64_t)blockIdx.x * supers_per_wg;
    const int64_t s1 = s0 + supers_per_wg < total_supers ? s0 + supers_per_wg : total_supers;
    const int64_t padded = (n_list + 15) & ~(int64_t)15;                       // length of the index lists
    const float *src = (op ? G : A) + 4 * lane;
    const int64_t ld = op ? ldg : lda;
    const int32_t *rows = op ? rg : ra;

    // NO BRANCH AROUND A LOAD in the loop: a super-step past the end of the slab re-reads the slab's
    // last one (its rows are then multiplied as zeros) instead of being skipped — with loads under
    // an `if` hipcc's wait-count pass merges the two paths pessimistically and every pass waits for
    // nearly all outstanding loads.
    f32x4 ring[kAtgDepth][8];
    auto fetch = [&](int64_t ss, f32x4 (&v)[8]) {       // loads only — no arithmetic on the results
        ss = ss < s1 - 1 ? ss : s1 - 1;
        int64_t pos = ss * 32 + 8 * w8;                  // (a group of 8 entries lies inside the padded
        pos = pos < padded - 8 ? pos : padded - 8;       //  list or wholly past it: then any valid group)
        const int32_t *idx = rows + pos;                 // (wave-uniform: one scalar load)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f32x4 *)(src + (int64_t)idx[j] * ld);
    };
#ifndef ATG_ABLATE      /* experiment builds only (tools/build_gemm_variants.sh atg): 4 = no loads in the loop */
#define ATG_ABLATE 0
#endif
    auto publish = [&](int64_t ss, f32x4 (&v)[8], unsigned char *buf) {
        // scale, zero the rows past the end of the list (and of the slab), split each column's 8
        // k-values into (h, m) fp16 parts: one 16-byte fragment per tile and split
        const int64_t end = ss < s1 ? n_list : 0;        // (a super-step of the next slab: all zeros)
        const int64_t k0 = ss * 32 + 8 * w8;
        float x[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = k0 + j < end;              // (uniform)
            x[j][0] = live ? v[j].x * my_scale : 0.f;
            x[j][1] = live ? v[j].y * my_scale : 0.f;
            x[j][2] = live ? v[j].z * my_scale : 0.f;
            x[j][3] = live ? v[j].w * my_scale : 0.f;
        }
        // fragment lane: this lane's column slot in the k-half this wave loads; tiles 2q + (lane >> 5)
        unsigned char *mine = buf + ((size_t)((op * 2 + (w8 >> 1)) * 8 + (lane >> 5)) * NS) * kFragBytes +
                              ((lane & 31) + 32 * (w8 & 1)) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t hh[4], mm[4];
            [[maybe_unused]] uint32_t ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SCH == 1) {
                    split3_pair(x[2 * j][q], x[2 * j + 1][q], hh[j], mm[j], ll[j]);
                } else {
                    f32x2 p = {x[2 * j][q], x[2 * j + 1][q]};
                    const h16x2 ph = __builtin_convertvector(p, h16x2);
                    const f32x2 pb = __builtin_convertvector(ph, f32x2);
                    f32x2 r = {p.x - pb.x, p.y - pb.y};
                    const h16x2 pm = __builtin_convertvector(r, h16x2);
                    hh[j] = __builtin_bit_cast(uint32_t, ph);
                    mm[j] = __builtin_bit_cast(uint32_t, pm);
                }
            }
            unsigned char *tile = mine + (size_t)(2 * q) * NS * kFragBytes;      // tile 2q + (lane >> 5)
            *(u32x4 *)(tile) = u32x4{hh[0], hh[1], hh[2], hh[3]};
            *(u32x4 *)(tile + kFragBytes) = u32x4{mm[0], mm[1], mm[2], mm[3]};
            if constexpr (SCH == 1) *(u32x4 *)(tile + 2 * kFragBytes) = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
    };

#ifndef ATG_B3_STEP16     /* 0 (experiment builds): the three-part scheme on 32-row super-steps in ONE buffer, two barriers each */
#define ATG_B3_STEP16 1
#endif
    if constexpr (SCH == 1 && ATG_B3_STEP16) {
        // Three parts per operand: a 32-row super-step's fragments are 96 KiB, two of them do not fit the LDS,
        // and in one buffer publish (270 vector instructions per wave: the split into three bf16 parts) and
        // multiply (96 MFMAs) alternate between two barriers with the matrix pipe idle for 45 % of the cycles
        // (stamps: 11 200 cycles per 32 rows against 6 144 of MFMA time; this form: 9 170 — and 7.25 -> 7.05 ms,
        // because the chip answers the denser pipe with a lower clock, 1.72 -> 1.59 GHz).  Here the unit of the LDS is ONE MFMA
        // step of 16 rows (48 KiB) in a ring of THREE: a super-step's two steps are multiplied one per barrier
        // interval, and during the SECOND interval every wave publishes its 8 rows of the NEXT super-step into
        // the two buffers that are free then — two pair splits (12 VALU each) behind every block of six MFMAs,
        // a tile's three fragment stores behind its fourth pair, fenced so that the order survives the
        // scheduler (hipcc, left alone, gathers the vector work in one place): it issues in the shadow of the
        // wave's own MFMAs and of its SIMD partner's.  Same MFMA order over the rows: the same bits as the
        // one-buffer form.  One code path for every wave and every super-step (a step past the slab publishes
        // zeros into a free buffer): the accumulators never cross a branch.
        constexpr int kStepBytes = atg_buf_bytes<SCH>() / 2;            // [operand 2][tile 8][part 3][fragment]
#ifndef ATG_B3_STAGGER    /* 1 (experiment builds): waves 4-7 split BEFORE a block's MFMAs, waves 0-3 behind them */
#define ATG_B3_STAGGER 0
#endif
        const int st_of = w8 >> 1, kh = w8 & 1;                          // this wave's rows: step of the super-step, k-half
        const bool late = wave >= 4;                                      // (uniform) the SIMD partner of wave - 4
        const int64_t n_sup = s1 - s0;
        f32x4 v[8];
        auto fetch1 = [&](int64_t ss) {                        // loads only (rows of a super-step past the slab: its last)
            ss = ss < s1 - 1 ? ss : s1 - 1;
            int64_t pos = ss * 32 + 8 * w8;
            pos = pos < padded - 8 ? pos : padded - 8;
            const int32_t *idx = rows + pos;                    // (wave-uniform: one scalar load)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4 *)(src + (int64_t)idx[j] * ld);
        };
        // (column sums of G: lane l of a G wave holds columns 4l .. 4l + 3 of every row it loads; the 8 rows
        //  of a publish are summed first, then added to the running sum with a compensation term)
        const bool do_cs = cs_partial != nullptr && op == 1;                              // (uniform)
        float cs_sum[4] = {0.f, 0.f, 0.f, 0.f}, cs_comp[4] = {0.f, 0.f, 0.f, 0.f}, cs_loc[4] = {0.f, 0.f, 0.f, 0.f};
        auto pair_of = [&](int q, int j, int64_t k0, int64_t end, uint32_t &h_, uint32_t &m_, uint32_t &l_) __attribute__((always_inline)) {
            const bool l0 = k0 + 2 * j < end, l1 = k0 + 2 * j + 1 < end;                  // (uniform)
            const float a = q == 0 ? v[2 * j].x : (q == 1 ? v[2 * j].y : (q == 2 ? v[2 * j].z : v[2 * j].w));
            const float c = q == 0 ? v[2 * j + 1].x : (q == 1 ? v[2 * j + 1].y : (q == 2 ? v[2 * j + 1].z : v[2 * j + 1].w));
            const float am = l0 ? a : 0.f, cm = l1 ? c : 0.f;
            split3_pair(am, cm, h_, m_, l_);
            if (do_cs) {
                cs_loc[q] = j == 0 ? am + cm : cs_loc[q] + (am + cm);
                if (j == 3) {
                    const float y = cs_loc[q] - cs_comp[q], t = cs_sum[q] + y;
                    cs_comp[q] = (t - cs_sum[q]) - y;
                    cs_sum[q] = t;
                }
            }
        };
        // where this wave's fragments of a step buffer go: tile 2q + (lane >> 5) at + 2 q NS fragments
        const int mine_off = (int)(((size_t)(op * 8 + (lane >> 5)) * NS) * kFragBytes) + ((lane & 31) + 32 * kh) * 16;
        auto tile_store = [&](unsigned char *mine, int q, const uint32_t (&hh)[4], const uint32_t (&mm)[4], const uint32_t (&ll)[4]) __attribute__((always_inline)) {
            unsigned char *tile = mine + (size_t)(2 * q) * NS * kFragBytes;
            *(u32x4 *)(tile) = u32x4{hh[0], hh[1], hh[2], hh[3]};
            *(u32x4 *)(tile + kFragBytes) = u32x4{mm[0], mm[1], mm[2], mm[3]};
            *(u32x4 *)(tile + 2 * kFragBytes) = u32x4{ll[0], ll[1], ll[2], ll[3]};
        };
        auto publish1 = [&](int64_t ss, unsigned char *dst) {   // (the prologue's: nothing to hide under yet)
            const int64_t end = ss < s1 ? n_list : 0, k0 = ss * 32 + 8 * w8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t hh[4], mm[4], ll[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pair_of(q, j, k0, end, hh[j], mm[j], ll[j]);
                tile_store(dst + mine_off, q, hh, mm, ll);
            }
        };
        // one 16-row step: 48 MFMAs per wave; PUBLISH: super-step `ss` goes out between the blocks
        auto multiply1 = [&](const unsigned char *buf, const bool publish, int64_t ss, unsigned char *dst) __attribute__((always_inline)) {
            const int64_t end = ss < s1 ? n_list : 0, k0 = ss * 32 + 8 * w8;
            const unsigned char *gbase = buf + (size_t)8 * NS * kFragBytes;
            uint32_t hh[4], mm[4], ll[4];
            // (A tile outermost; the G fragments of block b + 1 are requested BEFORE block b's MFMAs — the fences
            //  around the publish chunks would otherwise leave every block waiting for its own LDS reads)
            u32x4 Af[NS], Bf[2][NS];
            auto a_read = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    Af[sp] = *(const u32x4 *)(buf + (((2 * iw + ib) * NS + sp) * 64 + lane) * 16);
            };
            auto b_read = [&](int blk, u32x4 (&dst)[NS]) __attribute__((always_inline)) {
                const unsigned char *gb = gbase + ((4 * jh + (blk & 3)) * NS) * kFragBytes;
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) dst[sp] = *(const u32x4 *)(gb + (sp * 64 + lane) * 16);
            };
            a_read(0);
            b_read(0, Bf[0]);
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                const int ib = blk >> 2, jb = blk & 3;
                if (blk == 4) a_read(1);
                if (blk + 1 < 8) b_read(blk + 1, Bf[(blk + 1) & 1]);
                const u32x4 Bh = Bf[blk & 1][0], Bm = Bf[blk & 1][1], Bl = Bf[blk & 1][2];
                // block b = 4 ib + jb: for tile q = b >> 1 the pairs (rows 2j, 2j + 1), j = 2 (b & 1), + 1.
                // (STAGGER, measured and left off — 10 280 cycles per 32 rows against 9 170: waves 4-7 take a
                //  block's vector work BEFORE its MFMAs, waves 0-3 after, so that SIMD partners alternate;
                //  the branches are uniform and hold no accumulator)
                const int q = blk >> 1;
                auto chunk = [&]() __attribute__((always_inline)) {
                    pair_of(q, 2 * (blk & 1), k0, end, hh[2 * (blk & 1)], mm[2 * (blk & 1)], ll[2 * (blk & 1)]);
                    pair_of(q, 2 * (blk & 1) + 1, k0, end, hh[2 * (blk & 1) + 1], mm[2 * (blk & 1) + 1], ll[2 * (blk & 1) + 1]);
                    if (blk & 1) tile_store(dst + mine_off, q, hh, mm, ll);
                };
                if (publish) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ATG_B3_STAGGER && late) chunk();
                    __builtin_amdgcn_sched_barrier(0);
                }
                f32x16 t = acc[ib][jb];
                t = mfma(Af[0], Bl, t);        // smallest terms first
                t = mfma(Af[2], Bh, t);
                t = mfma(Af[1], Bm, t);
                t = mfma(Af[0], Bm, t);
                t = mfma(Af[1], Bh, t);
                t = mfma(Af[0], Bh, t);
                acc[ib][jb] = t;
                if (publish) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ATG_B3_STAGGER && late)) chunk();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (n_sup > 0) {                                        // (uniform over the workgroup)
            // ring of three step buffers (byte offsets): e = the super-step's first step, o = its second, p = free
            int e = 0, o = kStepBytes, p = 2 * kStepBytes;
            fetch1(s0);
            publish1(s0, lds + (st_of ? o : e));
            fetch1(s0 + 1);
            __syncthreads();
            for (int64_t i = 0; i < n_sup; ++i) {
                multiply1(lds + e, false, 0, nullptr);
                __syncthreads();                                // (everyone has left buffer e: the next super-step's second step goes there)
                multiply1(lds + o, true, s0 + i + 1, lds + (st_of ? e : p));
                fetch1(s0 + i + 2);
                __syncthreads();
                const int e2 = p, o2 = e, p2 = o;
                e = e2; o = o2; p = p2;
            }
        }
        if (cs_partial != nullptr) {                            // (uniform; the ring is free: the loop ends with a barrier)
            float *red = (float *)lds;                          // [G wave 4][256]
            if (op == 1) *(f32x4 *)(red + w8 * 256 + 4 * lane) = f32x4{cs_sum[0], cs_sum[1], cs_sum[2], cs_sum[3]};
            __syncthreads();
            if (tid < 256)
                cs_partial[(size_t)blockIdx.x * 256 + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
        }
    } else
    if (s0 < s1) {                                               // (uniform over the workgroup)
#pragma unroll
        for (int d = 0; d < kAtgDepth; ++d) {
            fetch(s0 + d, ring[d]);
            // (in THIS order: loads retire in order, and the scheduler, left alone, may issue the
            //  groups last-first — the loop's first wait would then be for the youngest load)
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int64_t base = s0; base < s1; base += kAtgDepth) {
#pragma unroll
            for (int d = 0; d < kAtgDepth; ++d) {
                const int64_t ss = base + d;                     // (ss >= s1 in the last pass: zeros)
                unsigned char *buf = lds + (SCH == 0 ? (int)((ss - s0) & 1) * atg_buf_bytes<SCH>() : 0);
                // (a scheduling fence: the arithmetic of THIS super-step's publish must not move up
                //  into the previous one — it would wait there for loads that are one step younger)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SCH == 1) __syncthreads();     // (one buffer: everyone has multiplied the previous super-step)
                publish(ss, ring[d], buf);
                if (!(ATG_ABLATE & 4)) fetch(ss + kAtgDepth, ring[d]);
                __syncthreads();
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const unsigned char *ab = buf + (size_t)((0 * 2 + s2) * 8) * NS * kFragBytes;
                    const unsigned char *gbase = buf + (size_t)((1 * 2 + s2) * 8) * NS * kFragBytes;
                    u32x4 Af[2][NS];
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
                            Af[ib][sp] = *(const u32x4 *)(ab + (((2 * iw + ib) * NS + sp) * 64 + lane) * 16);
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb) {
                        const unsigned char *gb = gbase + ((4 * jh + jb) * NS) * kFragBytes;
                        const u32x4 Bh = *(const u32x4 *)(gb + (0 * 64 + lane) * 16);
                        const u32x4 Bm = *(const u32x4 *)(gb + (1 * 64 + lane) * 16);
                        [[maybe_unused]] u32x4 Bl;
                        if constexpr (SCH == 1) Bl = *(const u32x4 *)(gb + (2 * 64 + lane) * 16);
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) {
                            f32x16 t = acc[ib][jb];
                            if constexpr (SCH == 1) {          // smallest terms first
                                t = mfma(Af[ib][0], Bl, t);
                                t = mfma(Af[ib][2], Bh, t);
                                t = mfma(Af[ib][1], Bm, t);
                                t = mfma(Af[ib][0], Bm, t);
                                t = mfma(Af[ib][1], Bh, t);
                                t = mfma(Af[ib][0], Bh, t);
                            } else {
                                t = mfma_h(Af[ib][1], Bh, t);
                                t = mfma_h(Af[ib][0], Bm, t);
                                t = mfma_h(Af[ib][0], Bh, t);
                            }
                            acc[ib][jb] = t;
                        }
                    }
                }
            }
        }
    }
#ifdef GEMM_PROFILE_STAMPS
    if (lane == 0) {           // per wave: lifetime of the row loop (the 256 x 256 partial result is stored after it)
        unsigned long long st_end;
        GEMM_STAMP(st_end);
        atomicAdd(&g_gemm_stamps[3], st_end - st_begin);
        atomicAdd(&g_gemm_stamps[4], 1ull);
    }
#endif
    // D[i][j] of tiles (Ta, Tb): i = (reg&3) + 8*(reg>>2) + 4*(lane>>5), j = lane&31;
    // column of tile T, slot u: 128*(T&1) + 4*u + (T>>1)
    float *out = partial + (size_t)blockIdx.x * (kK * kN);
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int ta = 2 * iw + ib, tb = 4 * jh + jb;
            const int col = 128 * (tb & 1) + 4 * c + (tb >> 1);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int row = 128 * (ta & 1) + 4 * i + (ta >> 1);
                out[row * kN + col] = SCH == 0 ? acc[ib][jb][reg] * back_a * back_b : acc[ib][jb][reg];
            }
        }
}

__global__ __launch_bounds__(256) void atg_colsum_reduce_kernel(const float *__restrict__ cs_partial, int n_wg,
                                                                float *__restrict__ colsum)
{
    double s = 0.0;                                          // (workgroup order: deterministic)
    for (int w = 0; w < n_wg; ++w) s += (double)cs_partial[(size_t)w * 256 + threadIdx.x];
    colsum[threadIdx.x] = (float)s;
}

__global__ __launch_bounds__(256) void atg_reduce_kernel(const float *__restrict__ partial, int n_wg,
                                                         float *__restrict__ out, int64_t ldo)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;          // one output element
    float s = 0.f;
    for (int w = 0; w < n_wg; ++w) s += partial[(size_t)w * (kK * kN) + idx];
    out[(int64_t)(idx >> 8) * ldo + (idx & 255)] = s;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient at bf16 storage (config C5): grad_W[128, 128] = Σ_r A[ra[r], :]ᵀ ⊗ G[rg[r], :]
// over a LIST of rows, bf16 operands, fp32 accumulation and fp32 result (`inputᵀ · grad_support`,
// the backward of pygcn/layers.py:33).  One bf16 MFMA per product; the kernel is a stream (512 B of
// operand rows per listed vertex against 2·128·128 flops), built like gemm_atg256_h2_kernel:
//
//   * the reduction runs over the vertices, so an MFMA lane needs 8 consecutive ROWS of one
//     column.  A wave loads one whole row per instruction — 128 bf16 = 256 contiguous bytes, one
//     dword (columns 2l, 2l+1) per lane, row address from the list on the scalar unit (the row
//     gather is free) — 16 rows per step.  v_perm_b32 then packs the low halves of two rows'
//     dwords (column 2l) and the high halves (column 2l + 1) into the MFMA's k-pairs, and ONE
//     v_permlane32_swap per register moves rows 8..15 of the low half-wave's columns to the upper
//     half-wave: four 32-column fragments per step,
//         tile t, lane (i, h):  column 64·(t & 1) + 2·i + (t >> 1),  k = 8·h + (0..7)
//     — a permutation of the columns that only decides where a result element is stored;
//   * waves 0-1 stream A (alternate steps), waves 2-3 stream G; fragments go through LDS
//     (double-buffered, one barrier per two steps); wave (iw, jw) multiplies the two A tiles
//     {2iw, 2iw+1} by the two G tiles {2jw, 2jw+1}: a quarter of the 128 x 128 result;
//   * loads run kAtgBfDepth iterations ahead in a register ring; slabs of the list go to
//     workgroups, the partial products are added in slab order by atg_reduce_kernel_n
//     (deterministic, no atomics).  Lists are padded to a multiple of 16 entries with valid
//     indices; rows past n_list are multiplied as zeros.
// (Measured and NOT adopted here: the branch-free ring of gemm_atg256_h2_kernel — steps past the end
//  clamped and multiplied as zeros, so that hipcc's counted waits keep two iterations of loads in
//  flight.  The waits came out as intended (vmcnt 47..32) and the kernel got 10 % SLOWER at full
//  height, 5.40 against 4.88 ms at C5: with one dword per lane the address unit, not the wait, is
//  this kernel's limit, and the burst after a drain suits it better.)
constexpr int kAtgBfDepth = 3;
constexpr int kAtgBfBuf = 2 * 2 * 4 * kFragBytes;      // [operand 2][step 2][tile 4] fragments = 16 KiB

__global__ __launch_bounds__(256, 2) void gemm_atg128_bf16_kernel(
    const uint16_t *__restrict__ A, int64_t lda, const int32_t *__restrict__ ra,
    const uint16_t *__restrict__ G, int64_t ldg, const int32_t *__restrict__ rg, int64_t n_list,
    float *__restrict__ partial, int64_t iters_per_wg)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kAtgBfBuf];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int op = wave >> 1, st = wave & 1;              // operand this wave streams; its step parity
    const int iw = wave & 1, jw = wave >> 1;              // result quarter: A tiles 2iw.., G tiles 2jw..
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int64_t total_steps = (n_list + 15) >> 4;
    const int64_t total_iters = (total_steps + 1) >> 1;
    const int64_t i0 = (int64_t)blockIdx.x * iters_per_wg;
    const int64_t i1 = i0 + iters_per_wg < total_iters ? i0 + iters_per_wg : total_iters;
    const uint32_t *src = (const uint32_t *)(op ? G : A) + lane;      // dword = columns 2l, 2l + 1
    const int64_t ld2 = (op ? ldg : lda) >> 1;                         // row pitch in dwords
    const int32_t *rows = op ? rg : ra;

    uint32_t ring[kAtgBfDepth][16];
    auto fetch = [&](int64_t it, uint32_t (&v)[16]) {      // loads only
        const int64_t step = 2 * it + st;
        if (step < total_steps) {                          // (wave-uniform)
            const int32_t *idx = rows + step * 16;         // scalar loads
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[(int64_t)idx[k] * ld2];
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = 0u;
        }
    };
    auto publish = [&](int64_t it, uint32_t (&v)[16], unsigned char *buf) {
        const int64_t step = 2 * it + st;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (step * 16 + k >= n_list) v[k] = 0u;        // (uniform: rows past the end of the list)
        uint32_t f0[2][4], f1[2][4];                       // [even / odd column][k pair]: rows 0-7, rows 8-15
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f0[0][j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x05040100u);
            f0[1][j] = __builtin_amdgcn_perm(v[2 * j + 1], v[2 * j], 0x07060302u);
            f1[0][j] = __builtin_amdgcn_perm(v[2 * j + 9], v[2 * j + 8], 0x05040100u);
            f1[1][j] = __builtin_amdgcn_perm(v[2 * j + 9], v[2 * j + 8], 0x07060302u);
        }
        unsigned char *mine = buf + ((op * 2 + st) * 4) * kFragBytes;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // rows 8-15 of the LOW half-wave's columns <-> rows 0-7 of the HIGH half-wave's
                auto sw = __builtin_amdgcn_permlane32_swap(f0[e][j], f1[e][j], false, false);
                lo[j] = sw[0];                             // tile 2e    : columns 2i + e
                hi[j] = sw[1];                             // tile 2e + 1: columns 64 + 2i + e
            }
            *(u32x4 *)(mine + ((2 * e + 0) * 64 + lane) * 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            *(u32x4 *)(mine + ((2 * e + 1) * 64 + lane) * 16) = u32x4{hi[0], hi[1], hi[2], hi[3]};
        }
    };

#pragma unroll
    for (int d = 0; d < kAtgBfDepth; ++d)
        if (i0 + d < i1) fetch(i0 + d, ring[d]);
    for (int64_t base = i0; base < i1; base += kAtgBfDepth) {
#pragma unroll
        for (int d = 0; d < kAtgBfDepth; ++d) {
            const int64_t it = base + d;
            if (it < i1) {                                                   // (uniform over the workgroup)
                unsigned char *buf = lds + (int)((it - i0) & 1) * kAtgBfBuf;
                publish(it, ring[d], buf);
                if (it + kAtgBfDepth < i1) fetch(it + kAtgBfDepth, ring[d]);
                __syncthreads();
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 Af[2], Gf[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        // physical tile order in LDS is (even-low, even-high, odd-low, odd-high) = t with
                        // column 64·(t & 1) + 2i + (t >> 1); a wave takes t = 2iw, 2iw + 1 (one parity)
                        Af[a] = *(const u32x4 *)(buf + (((0 * 2 + s2) * 4 + 2 * iw + a) * 64 + lane) * 16);
                        Gf[a] = *(const u32x4 *)(buf + (((1 * 2 + s2) * 4 + 2 * jw + a) * 64 + lane) * 16);
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = mfma(Af[a], Gf[b], acc[a][b]);
                }
            }
        }
    }
    // D[i][j] of tile pair (ta, tb): i = (reg & 3) + 8·(reg >> 2) + 4·(lane >> 5), j = lane & 31;
    // LDS tile index t = 2e + half  ->  column 64·half + 2·idx + e   (e = t >> 1, half = t & 1)
    float *out = partial + (size_t)blockIdx.x * (128 * 128);
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ta = 2 * iw + a, tb = 2 * jw + b;
            const int col = 64 * (tb & 1) + 2 * c + (tb >> 1);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int i = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int row = 64 * (ta & 1) + 2 * i + (ta >> 1);
                out[row * 128 + col] = acc[a][b][reg];
            }
        }
}

__global__ __launch_bounds__(256) void atg_reduce_kernel_n(const float *__restrict__ partial, int n_wg,
                                                           float *__restrict__ out, int64_t ldo, int N,
                                                           int elems)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;          // one output element
    if (idx >= elems) return;
    float s = 0.f;
    for (int w = 0; w < n_wg; ++w) s += partial[(size_t)w * elems + idx];
    out[(int64_t)(idx / N) * ldo + (idx % N)] = s;
}


}   // namespace

extern "C" {

size_t gcn_gemm_xw256_workspace_bytes(void) { return (size_t)kChunks * kChunkBytes; }

int gcn_gemm_xw256_f32(const float *X, int64_t ldx, const float *W, int64_t ldw, float *Y, int64_t ldy,
                       int64_t M, void *workspace, size_t workspace_bytes, void *stream)
{
    if (M < 0 || ldx < kK || ldy < kN || ldw < kN)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32: bad sizes");
    if (M == 0) return 0;
    if (X == nullptr || W == nullptr || Y == nullptr || workspace == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32: NULL pointer");
    if (workspace_bytes < gcn_gemm_xw256_workspace_bytes())
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_gemm_xw256_f32: workspace too small");
    if ((((uintptr_t)X) | ((uintptr_t)Y) | ((uintptr_t)workspace)) % 16 != 0 || (ldx % 4) != 0 ||
        (ldy % 4) != 0)
        return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_xw256_f32: X / Y rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(split_w_kernel, dim3(kChunks * 8 * 64 / 256), dim3(256), 0, s, W, ldw,
                       (uint16_t *)workspace);
    const int64_t tiles = (M + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL(gemm_xw256_kernel, dim3((unsigned)tiles), dim3(kThreads), 0, s, X, ldx,
                       (const uint16_t *)workspace, Y, ldy, M);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_gemm_xw256_f32 launch");
    return 0;
}

size_t gcn_gemm_xw256_h2_workspace_bytes(void)
{
    return (size_t)kH2HeaderBytes + (size_t)kChunks * kH2ChunkBytes;
}

#ifdef GEMM_PROFILE_STAMPS
int gcn_debug_gemm_stamps(unsigned long long *out8, int reset)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gemm_stamps), sizeof(z));
    if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, sizeof(z));
    return (int)e;
}
int gcn_debug_gemm_step_stamps(unsigned long long *out16, int reset)
{
    unsigned long long z[16] = {0};
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gemm_step_stamps), sizeof(z));
    if (e == hipSuccess && reset) e = hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_step_stamps), z, sizeof(z));
    return (int)e;
}
#endif

size_t gcn_gemm_xw256_b3_workspace_bytes(void)
{
    return (size_t)kH2HeaderBytes + (size_t)kChunks * SchemeK<1>::ChunkBytes;
}

}   // extern "C"

// both decompositions of the 256 x 256 product (sch 0: two scaled fp16 parts, 1: three bf16 parts)
static int xw256_launch(const char *who, int sch, const float *X, int64_t ldx, const int32_t *x_rows, const float *W,
                        int64_t ldw, float *Y, int64_t ldy, int64_t M, const float *x_absmax_bound,
                        float *y_absmax, const gcn_gemm_epilogue *epi, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    (void)who;
    const float *mask_src = epi ? epi->mask_src : nullptr;
    const int64_t ld_mask = epi ? epi->ld_mask : 0;
    H2Epi ep = {};
    if (epi != nullptr) {
        if (!(epi->dropout_p >= 0.f) || epi->dropout_p >= 1.f)
            return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_h2: dropout_p must be in [0, 1)");
        if (epi->dropout_p > 0.f && !epi->relu)
            return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_h2: dropout needs relu (out > 0 encodes the mask)");
        if ((mask_src != nullptr || epi->mask_bits != nullptr) && (epi->bias != nullptr || epi->relu || epi->dropout_p > 0.f))
            return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_h2: forward epilogue and backward mask exclude each other");
        ep.bias = epi->bias;
        ep.relu = epi->relu ? 1 : 0;
        ep.drop_thresh = gcn_dropout_threshold16(epi->dropout_p);
        ep.drop_scale = gcn_dropout_scale16(ep.drop_thresh);
        ep.drop_row_base = epi->drop_row_base;
        ep.seed_lo = (uint32_t)epi->seed;
        ep.seed_hi = (uint32_t)(epi->seed >> 32);
        ep.seed_dev = epi->seed_dev;
        ep.mask_src = mask_src;
        ep.ld_mask = ld_mask;
        ep.mask_rows = (mask_src != nullptr || epi->mask_bits != nullptr) ? epi->mask_rows : nullptr;
        ep.mask_scale = epi->mask_scale;
        ep.keep_bits_out = epi->keep_bits_out;
        ep.mask_bits = epi->mask_bits;
    }
    if (M < 0 || ldx < kK || ldy < kN || ldw < kN)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_h2: bad sizes");
    if (M == 0) return 0;
    if (X == nullptr || W == nullptr || Y == nullptr || workspace == nullptr ||
        (sch == 0 && x_absmax_bound == nullptr))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_h2: NULL pointer");
    if (workspace_bytes < (sch == 0 ? gcn_gemm_xw256_h2_workspace_bytes() : gcn_gemm_xw256_b3_workspace_bytes()))
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_gemm_xw256_f32_h2: workspace too small");
    if ((((uintptr_t)X) | ((uintptr_t)Y) | ((uintptr_t)workspace)) % 16 != 0 || (ldx % 4) != 0 ||
        (ldy % 4) != 0 || (((uintptr_t)x_absmax_bound) | ((uintptr_t)y_absmax)) % 4 != 0 ||
        ((uintptr_t)mask_src) % 16 != 0 || (mask_src != nullptr && (ld_mask % 4 != 0 || ld_mask < kN)) ||
        ((uintptr_t)ep.bias) % 16 != 0)
        return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_xw256_f32_h2: X / Y rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
#ifndef GEMM_B3_S16
#define GEMM_B3_S16 1         /* 1: the three-part scheme on contiguous rows runs gemm_xw256_s16_kernel (128-row tiles, stores
                                 under the next tile's MFMAs); 0: always gemm_xw256_h2_kernel<., 1> (bit-identical to round 1) */
#endif
    // instantiation by store section (compile-time options, see the kernels): 0 plain, 2 backward mask,
    // 1 bias, 4 bias + ReLU, 5 + dropout at p = 1/2 (one-bit keep fields), 6 + dropout at another p
    const bool fwd = ep.bias != nullptr || ep.relu || ep.drop_thresh != 0u;
    const int variant = !fwd ? (ep.mask_bits != nullptr ? 3 : (ep.mask_src != nullptr ? 2 : 0))
                             : (!ep.relu ? 1 : (ep.drop_thresh == 0u ? 4 : (ep.drop_thresh == 32768u ? 5 : 6)));
    // (s16 addresses a tile's rows as 32-bit offsets from the tile origin, takes no row list, and has no
    //  instantiation for dropout at p != 1/2)
    const bool s16 = sch == 1 && GEMM_B3_S16 && x_rows == nullptr && variant != 6 && ldx < (1 << 21) && ldy < (1 << 21);
    // (the one-bit mask exists in the contiguous-row kernel's lane order only: a launch that cannot take that kernel
    //  must be given mask_src / no keep_bits_out — refuse rather than ignore)
    if ((ep.mask_bits != nullptr || ep.keep_bits_out != nullptr) &&
        !(s16 && (ep.mask_bits != nullptr ? !fwd : (ep.relu != 0)) &&
          (((uintptr_t)ep.mask_bits) | ((uintptr_t)ep.keep_bits_out)) % 8 == 0))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw256_f32_b3: keep_bits_out / mask_bits need contiguous rows, the three-part "
                                               "scheme, ReLU with dropout_p in {0, 1/2} (forward) or no forward epilogue (backward)");
    if (sch == 0)
        hipLaunchKernelGGL(split_w_h2_kernel, dim3(1), dim3(1024), 0, s, W, ldw, (unsigned char *)workspace);
    else if (s16)
        hipLaunchKernelGGL(split_w_t16_kernel, dim3(kT16Stages * 8 * 64 / 256), dim3(256), 0, s, W, ldw,
                           (uint16_t *)((unsigned char *)workspace + kH2HeaderBytes));
    else      // (the three-part image of gemm_xw256_kernel, behind the same header space)
        hipLaunchKernelGGL(split_w_kernel, dim3(kChunks * 8 * 64 / 256), dim3(256), 0, s, W, ldw,
                           (uint16_t *)((unsigned char *)workspace + kH2HeaderBytes));
    const int64_t tiles = s16 ? (M + kS16Rows - 1) / kS16Rows : (M + kTileRows - 1) / kTileRows;
    // dynamic LDS of the X-through-LDS build: the two W stages + every wave's X ring (> 64 KiB)
    const size_t dyn = s16 ? (size_t)kS16LdsBytes : (size_t)kXLdsBytes;
    if (dyn) {
        static bool raised = false;          // (idempotent; a benign race sets it twice)
        if (!raised) {
            const void *all[] = {(const void *)gemm_xw256_h2_kernel<0>, (const void *)gemm_xw256_h2_kernel<1>,
                                 (const void *)gemm_xw256_h2_kernel<2>, (const void *)gemm_xw256_h2_kernel<4>,
                                 (const void *)gemm_xw256_h2_kernel<5>, (const void *)gemm_xw256_h2_kernel<6>,
                                 (const void *)gemm_xw256_h2_kernel<0, 1>, (const void *)gemm_xw256_h2_kernel<1, 1>,
                                 (const void *)gemm_xw256_h2_kernel<2, 1>, (const void *)gemm_xw256_h2_kernel<4, 1>,
                                 (const void *)gemm_xw256_h2_kernel<5, 1>, (const void *)gemm_xw256_h2_kernel<6, 1>};
            for (const void *k : all) {
                hipError_t ae = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kXLdsBytes);
                if (ae != hipSuccess) return gcn_internal_fail_hip((int)ae, "gcn_gemm_xw256_f32_h2: LDS size");
            }
            const void *tall[] = {(const void *)gemm_xw256_s16_kernel<0>, (const void *)gemm_xw256_s16_kernel<1>,
                                  (const void *)gemm_xw256_s16_kernel<2>, (const void *)gemm_xw256_s16_kernel<3>,
                                  (const void *)gemm_xw256_s16_kernel<4>, (const void *)gemm_xw256_s16_kernel<5>,
                                  (const void *)gemm_xw256_s16_kernel<4, true>, (const void *)gemm_xw256_s16_kernel<5, true>};
            for (const void *k : tall) {
                hipError_t ae = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS16LdsBytes);
                if (ae != hipSuccess) return gcn_internal_fail_hip((int)ae, "gcn_gemm_xw256_f32_b3: LDS size");
            }
            raised = true;
        }
    }
    const unsigned grid = (fwd && !GEMM_H2_EPI_PERSIST) ? (unsigned)tiles : (unsigned)std::min<int64_t>(tiles, GEMM_H2_GRID);
#define GCN_LAUNCH_H2(V)                                                                             \
    do {                                                                                             \
        if (sch == 0)                                                                                \
            hipLaunchKernelGGL((gemm_xw256_h2_kernel<V, 0>), dim3(grid), dim3(kThreads), dyn, s, X, ldx, x_rows, \
                               (const unsigned char *)workspace, x_absmax_bound, Y, ldy, M, (uint32_t *)y_absmax, ep); \
        else if (s16)                                                                                \
            hipLaunchKernelGGL((gemm_xw256_s16_kernel<(V == 6 ? 5 : V)>), dim3(grid), dim3(kThreads), dyn, s, X, ldx, \
                               (const unsigned char *)workspace, Y, ldy, M, (uint32_t *)y_absmax, ep);     \
        else                                                                                         \
            hipLaunchKernelGGL((gemm_xw256_h2_kernel<V, 1>), dim3(grid), dim3(kThreads), dyn, s, X, ldx, x_rows, \
                               (const unsigned char *)workspace, x_absmax_bound, Y, ldy, M, (uint32_t *)y_absmax, ep); \
    } while (0)
    switch (variant) {
    case 0: GCN_LAUNCH_H2(0); break;
    case 1: GCN_LAUNCH_H2(1); break;
    case 2: GCN_LAUNCH_H2(2); break;
    case 3:          // (s16 only: checked above)
        hipLaunchKernelGGL((gemm_xw256_s16_kernel<3>), dim3(grid), dim3(kThreads), dyn, s, X, ldx,
                           (const unsigned char *)workspace, Y, ldy, M, (uint32_t *)y_absmax, ep);
        break;
    case 4:
    case 5:
        if (s16 && ep.keep_bits_out != nullptr) {        // (+ the one-bit form of the result)
            if (variant == 4)
                hipLaunchKernelGGL((gemm_xw256_s16_kernel<4, true>), dim3(grid), dim3(kThreads), dyn, s, X, ldx,
                                   (const unsigned char *)workspace, Y, ldy, M, (uint32_t *)y_absmax, ep);
            else
                hipLaunchKernelGGL((gemm_xw256_s16_kernel<5, true>), dim3(grid), dim3(kThreads), dyn, s, X, ldx,
                                   (const unsigned char *)workspace, Y, ldy, M, (uint32_t *)y_absmax, ep);
        } else if (variant == 4) {
            GCN_LAUNCH_H2(4);
        } else {
            GCN_LAUNCH_H2(5);
        }
        break;
    default: GCN_LAUNCH_H2(6); break;
    }
#undef GCN_LAUNCH_H2
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_gemm_xw256_f32_h2 launch");
    return 0;
}

extern "C" {

int gcn_gemm_xw256_f32_h2(const float *X, int64_t ldx, const int32_t *x_rows, const float *W,
                          int64_t ldw, float *Y, int64_t ldy, int64_t M, const float *x_absmax_bound,
                          float *y_absmax, const gcn_gemm_epilogue *epi, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    return xw256_launch("gcn_gemm_xw256_f32_h2", 0, X, ldx, x_rows, W, ldw, Y, ldy, M, x_absmax_bound, y_absmax,
                        epi, workspace, workspace_bytes, stream);
}

int gcn_gemm_xw256_f32_b3(const float *X, int64_t ldx, const int32_t *x_rows, const float *W,
                          int64_t ldw, float *Y, int64_t ldy, int64_t M, float *y_absmax,
                          const gcn_gemm_epilogue *epi, void *workspace, size_t workspace_bytes,
                          void *stream)
{
    return xw256_launch("gcn_gemm_xw256_f32_b3", 1, X, ldx, x_rows, W, ldw, Y, ldy, M, nullptr, y_absmax, epi,
                        workspace, workspace_bytes, stream);
}

size_t gcn_gemm_bf16_workspace_bytes(int64_t K, int64_t N)
{
    if ((K != 128 && K != 256) || (N != 128 && N != 256) || (K == 256 && N == 256)) return 0;
    return (size_t)K * (size_t)N * 2;
}

int gcn_gemm_xw_bf16(const void *X, int64_t ldx, const void *W, int64_t ldw, void *Y, int64_t ldy,
                     int64_t M, int64_t K, int64_t N, const gcn_gemm_epilogue *epi, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    H2Epi ep = {};
    if (epi != nullptr) {
        if (!(epi->dropout_p >= 0.f) || epi->dropout_p >= 1.f)
            return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: dropout_p must be in [0, 1)");
        if (epi->dropout_p > 0.f && !epi->relu)
            return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: dropout needs relu (out > 0 encodes the mask)");
        if (epi->mask_src != nullptr) {
            if (epi->bias != nullptr || epi->relu || epi->dropout_p > 0.f)
                return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: forward epilogue and backward mask exclude each other");
            if (((uintptr_t)epi->mask_src) % 16 != 0 || epi->ld_mask % 8 != 0 || epi->ld_mask < N)
                return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_xw_bf16: mask rows must be 16-byte aligned bf16 [*, N]");
            ep.mask_src = (const float *)epi->mask_src;      // (bf16 data; typed per kernel)
            ep.ld_mask = epi->ld_mask;
            ep.mask_rows = epi->mask_rows;
            ep.mask_scale = epi->mask_scale;
        }
        if (((uintptr_t)epi->bias) % 16 != 0)
            return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_xw_bf16: bias must be 16-byte aligned");
        ep.bias = epi->bias;
        ep.relu = epi->relu ? 1 : 0;
        ep.drop_thresh = gcn_dropout_threshold16(epi->dropout_p);
        ep.drop_scale = gcn_dropout_scale16(ep.drop_thresh);
        ep.drop_row_base = epi->drop_row_base;
        ep.seed_lo = (uint32_t)epi->seed;
        ep.seed_hi = (uint32_t)(epi->seed >> 32);
        ep.seed_dev = epi->seed_dev;
    }
    const bool fwd = ep.bias != nullptr || ep.relu || ep.drop_thresh != 0u;
    const size_t need = gcn_gemm_bf16_workspace_bytes(K, N);
    if (need == 0)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: (K, N) must be (128,128), (128,256) or (256,128)");
    if (M < 0 || ldx < K || ldy < N || ldw < N)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: bad sizes");
    if (M == 0) return 0;
    if (X == nullptr || W == nullptr || Y == nullptr || workspace == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: NULL pointer");
    if (workspace_bytes < need)
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_gemm_xw_bf16: workspace too small");
    if ((((uintptr_t)X) | ((uintptr_t)Y) | ((uintptr_t)workspace)) % 16 != 0 || (ldx % 8) != 0 ||
        (ldy % 8) != 0)
        return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_xw_bf16: X / Y rows must be 16-byte aligned");
    if (ldx > (1 << 24) || ldy > (1 << 24))      // (32 rows of a tile sit under one 32-bit buffer offset)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_xw_bf16: leading dimensions above 2^24 are not supported");
    hipStream_t s = (hipStream_t)stream;
    const int frags = (int)((K / 16) * (N / 32) * 64);
    hipLaunchKernelGGL(order_w_bf16_kernel, dim3((frags + 255) / 256), dim3(256), 0, s,
                       (const uint16_t *)W, ldw, (uint16_t *)workspace, (int)K, (int)N);
    const int64_t tiles = (M + 31) / 32;
    // a PERSISTENT grid: exactly the workgroups that are resident together on the 256 CUs (LDS image
    // and the instantiation's registers decide: 3 per CU at 128 x 128 … 1), asked of the runtime once
#define GCN_LAUNCH_BF16_E(KK, NN, EE)                                                               \
    do {                                                                                            \
        static int per_cu = 0;                                                                      \
        hipError_t ae = hipFuncSetAttribute((const void *)gemm_bf16_kernel<KK, NN, EE>,            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)need); \
        if (ae != hipSuccess) return gcn_internal_fail_hip((int)ae, "gcn_gemm_xw_bf16: LDS size"); \
        if (per_cu == 0) {                                                                          \
            int nb = 0;                                                                             \
            ae = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gemm_bf16_kernel<KK, NN, EE>, 256, need); \
            if (ae != hipSuccess) return gcn_internal_fail_hip((int)ae, "gcn_gemm_xw_bf16: occupancy"); \
            per_cu = nb > 0 ? nb : 1;                                                               \
        }                                                                                           \
        const unsigned grid = (unsigned)std::min<int64_t>((tiles + 3) / 4, (int64_t)256 * per_cu); \
        hipLaunchKernelGGL((gemm_bf16_kernel<KK, NN, EE>), dim3(grid), dim3(256), need, s,          \
                           (const uint16_t *)X, ldx, (const uint16_t *)workspace, (uint16_t *)Y,   \
                           ldy, M, tiles, ep);                                                      \
    } while (0)
#define GCN_LAUNCH_BF16(KK, NN)                                                                     \
    do {                                                                                            \
        if (fwd && !ep.relu) GCN_LAUNCH_BF16_E(KK, NN, 1);                                          \
        else if (fwd && ep.drop_thresh == 0u) GCN_LAUNCH_BF16_E(KK, NN, 4);                         \
        else if (fwd && ep.drop_thresh == 32768u) GCN_LAUNCH_BF16_E(KK, NN, 5);                     \
        else if (fwd) GCN_LAUNCH_BF16_E(KK, NN, 6);                                                 \
        else if (ep.mask_src == nullptr) GCN_LAUNCH_BF16_E(KK, NN, 0);                              \
        else if (ep.mask_rows == nullptr) GCN_LAUNCH_BF16_E(KK, NN, 2);                             \
        else GCN_LAUNCH_BF16_E(KK, NN, 3);                                                          \
    } while (0)
    if (K == 128 && N == 128) GCN_LAUNCH_BF16(128, 128);
    else if (K == 128 && N == 256) GCN_LAUNCH_BF16(128, 256);
    else GCN_LAUNCH_BF16(256, 128);
#undef GCN_LAUNCH_BF16
#undef GCN_LAUNCH_BF16_E
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_gemm_xw_bf16 launch");
    return 0;
}

static int64_t atg_wgs(int64_t n_list)
{
    const int64_t supers = (n_list + 31) / 32;
    return std::max<int64_t>(1, std::min<int64_t>(kAtgMaxWgs, (supers + kAtgSuperMin - 1) / kAtgSuperMin));
}

size_t gcn_gemm_atg256_workspace_bytes(int64_t n_list)
{
    if (n_list <= 0) return 256;
    return (size_t)atg_wgs(n_list) * (size_t)(kK * kN + kN) * sizeof(float);      // partial products + column sums
}

}   // extern "C"

static int atg256_launch(int sch, const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                         const int32_t *rows_g, int64_t n_list, const float *a_absmax_bound,
                         const float *g_absmax_bound, float *out, int64_t ldo, float *colsum_g, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    if (n_list < 0 || lda < kK || ldg < kN || ldo < kN)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg256_f32: bad sizes");
    if (out == nullptr) return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg256_f32: NULL output");
    if (colsum_g != nullptr && !(sch == 1 && ATG_B3_STEP16))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg256_f32: column sums exist in the three-part form only");
    hipStream_t s = (hipStream_t)stream;
    if (n_list == 0) {
        hipError_t e = hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)kN * 4, kK, s);
        if (e == hipSuccess && colsum_g != nullptr) e = hipMemsetAsync(colsum_g, 0, (size_t)kN * 4, s);
        return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_gemm_atg256_f32: memset");
    }
    if (A == nullptr || G == nullptr || rows_a == nullptr || rows_g == nullptr || workspace == nullptr ||
        (sch == 0 && (a_absmax_bound == nullptr || g_absmax_bound == nullptr)))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg256_f32: NULL pointer");
    if (workspace_bytes < gcn_gemm_atg256_workspace_bytes(n_list))
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_gemm_atg256_f32: workspace too small");
    if ((((uintptr_t)A) | ((uintptr_t)G)) % 16 != 0 || lda % 4 != 0 || ldg % 4 != 0)
        return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_atg256_f32: A / G rows must be 16-byte aligned");
    const int64_t n_wg = atg_wgs(n_list);
    const int64_t supers = (n_list + 31) / 32;
    const int64_t per = (supers + n_wg - 1) / n_wg;
    float *cs_partial = (float *)workspace + (size_t)n_wg * (kK * kN);
    {
        static bool lds_set = false;
        if (!lds_set) {
            hipError_t ae = hipFuncSetAttribute((const void *)gemm_atg256_h2_kernel<0>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, atg_lds_bytes<0>());
            if (ae == hipSuccess)
                ae = hipFuncSetAttribute((const void *)gemm_atg256_h2_kernel<1>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, atg_lds_bytes<1>());
            if (ae != hipSuccess) return gcn_internal_fail_hip((int)ae, "gcn_gemm_atg256_f32: LDS size");
            lds_set = true;
        }
    }
    if (sch == 0)
        hipLaunchKernelGGL(gemm_atg256_h2_kernel<0>, dim3((unsigned)n_wg), dim3(512), atg_lds_bytes<0>(), s, A, lda,
                           rows_a, G, ldg, rows_g, n_list, a_absmax_bound, g_absmax_bound, (float *)workspace, per,
                           (float *)nullptr);
    else
        hipLaunchKernelGGL(gemm_atg256_h2_kernel<1>, dim3((unsigned)n_wg), dim3(512), atg_lds_bytes<1>(), s, A, lda,
                           rows_a, G, ldg, rows_g, n_list, a_absmax_bound, g_absmax_bound, (float *)workspace, per,
                           colsum_g != nullptr ? cs_partial : (float *)nullptr);
    hipLaunchKernelGGL(atg_reduce_kernel, dim3(kK * kN / 256), dim3(256), 0, s, (const float *)workspace,
                       (int)n_wg, out, ldo);
    if (colsum_g != nullptr)
        hipLaunchKernelGGL(atg_colsum_reduce_kernel, dim3(1), dim3(256), 0, s, (const float *)cs_partial, (int)n_wg,
                           colsum_g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_gemm_atg256_f32 launch");
    return 0;
}

extern "C" {

int gcn_gemm_atg256_f32(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                        const int32_t *rows_g, int64_t n_list, const float *a_absmax_bound,
                        const float *g_absmax_bound, float *out, int64_t ldo, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    return atg256_launch(0, A, lda, rows_a, G, ldg, rows_g, n_list, a_absmax_bound, g_absmax_bound, out, ldo,
                         nullptr, workspace, workspace_bytes, stream);
}

int gcn_gemm_atg256_f32_b3(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                           const int32_t *rows_g, int64_t n_list, float *out, int64_t ldo, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    return atg256_launch(1, A, lda, rows_a, G, ldg, rows_g, n_list, nullptr, nullptr, out, ldo, nullptr, workspace,
                         workspace_bytes, stream);
}

int gcn_gemm_atg256_f32_b3_colsum(const float *A, int64_t lda, const int32_t *rows_a, const float *G, int64_t ldg,
                                  const int32_t *rows_g, int64_t n_list, float *out, int64_t ldo, float *colsum_g,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (colsum_g == nullptr) return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg256_f32_b3_colsum: NULL colsum_g");
    return atg256_launch(1, A, lda, rows_a, G, ldg, rows_g, n_list, nullptr, nullptr, out, ldo, colsum_g, workspace,
                         workspace_bytes, stream);
}

static int64_t atg_bf16_wgs(int64_t n_list)
{
    const int64_t iters = ((n_list + 15) / 16 + 1) / 2;
    return std::max<int64_t>(1, std::min<int64_t>(512, (iters + 7) / 8));
}

size_t gcn_gemm_atg_bf16_workspace_bytes(int64_t n_list, int64_t K, int64_t N)
{
    if (K != 128 || N != 128) return 0;
    if (n_list <= 0) return 256;
    return (size_t)atg_bf16_wgs(n_list) * (size_t)(128 * 128) * sizeof(float);
}

int gcn_gemm_atg_bf16(const void *A, int64_t lda, const int32_t *rows_a, const void *G, int64_t ldg,
                      const int32_t *rows_g, int64_t n_list, int64_t K, int64_t N, float *out,
                      int64_t ldo, void *workspace, size_t workspace_bytes, void *stream)
{
    if (K != 128 || N != 128)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg_bf16: (K, N) must be (128, 128)");
    if (n_list < 0 || lda < K || ldg < N || ldo < N || (lda & 1) || (ldg & 1))
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg_bf16: bad sizes (even leading dimensions >= K, N)");
    if (out == nullptr) return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg_bf16: NULL output");
    hipStream_t s = (hipStream_t)stream;
    if (n_list == 0) {
        hipError_t e = hipMemset2DAsync(out, (size_t)ldo * 4, 0, (size_t)N * 4, (size_t)K, s);
        return e == hipSuccess ? 0 : gcn_internal_fail_hip((int)e, "gcn_gemm_atg_bf16: memset");
    }
    if (A == nullptr || G == nullptr || rows_a == nullptr || rows_g == nullptr || workspace == nullptr)
        return gcn_internal_fail(GCN_E_BADARG, "gcn_gemm_atg_bf16: NULL pointer");
    if ((((uintptr_t)A) | ((uintptr_t)G)) % 4 != 0)
        return gcn_internal_fail(GCN_E_ALIGN, "gcn_gemm_atg_bf16: operands must be 4-byte aligned");
    if (workspace_bytes < gcn_gemm_atg_bf16_workspace_bytes(n_list, K, N))
        return gcn_internal_fail(GCN_E_WORKSPACE, "gcn_gemm_atg_bf16: workspace too small");
    const int64_t n_wg = atg_bf16_wgs(n_list);
    const int64_t iters = ((n_list + 15) / 16 + 1) / 2;
    const int64_t per = (iters + n_wg - 1) / n_wg;
    hipLaunchKernelGGL(gemm_atg128_bf16_kernel, dim3((unsigned)n_wg), dim3(256), 0, s, (const uint16_t *)A,
                       lda, rows_a, (const uint16_t *)G, ldg, rows_g, n_list, (float *)workspace, per);
    hipLaunchKernelGGL(atg_reduce_kernel_n, dim3(128 * 128 / 256), dim3(256), 0, s,
                       (const float *)workspace, (int)n_wg, out, ldo, 128, 128 * 128);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gcn_internal_fail_hip((int)e, "gcn_gemm_atg_bf16 launch");
    return 0;
}

}   // extern "C"
