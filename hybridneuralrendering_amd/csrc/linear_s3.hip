// fp32 dense layer on the bf16 matrix cores by exact operand splitting:  C[M,N] = act(A[M,K] * W[N,K]^T + bias[N] (+ R[ridx[m]])).
//
// Replaces the 256-wide nn.Linear (+LeakyReLU) layers of PointAggregator.viewmlp
// (models/aggregators/point_aggregators.py:948 block1, :972 block3) -- 78 % of the frame on the fp32 MFMA path of linear.hip.
//
// Arithmetic.  gfx950 runs v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate (157 TFLOP/s), 1/16 of the bf16 matrix rate, and has
// no TF32.  Every fp32 value x is the EXACT sum of three bf16 values, x = h + m + l (8 significand bits each, round-to-nearest:
// h = bf16(x), m = bf16(x - h), l = x - h - m, both subtractions exact), so a product is the sum of nine bf16 x bf16 products,
// each exact in fp32.  The kernel issues the six whose magnitude is >= 2^-16 |a w| (hh, hm, mh, hl, mm, lh) as
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped ones (ml, lm, ll) are <= 2^-23 |a w| together, i.e. the
// size of ONE fp32 rounding of the product.  The result therefore carries the error of an fp32 dot product (measured against
// fp64 in tests/test_linear_gpu.py beside the fp32-MFMA kernel: same error class) at 6/16 of its matrix-pipe time.
//
// Tiling (wave64).  256-thread workgroup = 4 waves, two workgroups per CU; a wave owns 32 rows x all 256 output columns
// (8 MFMA tiles = 128 accumulator registers), so the activation operand never goes through LDS: lane (row j = lane & 31,
// half h = lane >> 5) loads its own 64 contiguous bytes of row j per 32-wide K group straight into registers and splits them
// there.  The split weights are packed ONCE per checkpoint in fragment order ([k step][column tile][plane][lane][8 bf16]) and
// streamed global -> LDS by DMA (24 KiB per k step, ring of three), one ds_read_b128 per (k step, column tile, plane).
// Persistent workgroups stride the 128-row tiles.
#include <stdlib.h>

#include "hnr_common.h"

namespace hnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int S3_ROWS = 128;                 // rows per workgroup tile (4 waves x 32)
constexpr int S3_STEP_BYTES = 8 * 3 * 1024;  // one k step (16 k) of the packed weights: [8 column tiles][3 planes][64 lanes][16 B]
constexpr int S3_GROUP_BYTES = 2 * 8 * 3 * 1024;   // one 32-wide K group of the packed weights: [2 k steps][8 column tiles][3 planes][64 lanes][16 B]

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// x0,x1 -> packed (h), (m), (l) bf16 pairs; x = h + m + l exactly (round-to-nearest-even at each step)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &ph, unsigned &pm, unsigned &pl)
{
    ph = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    pm = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
    pl = cvt_pk_bf16(s0, s1);
}

__device__ __forceinline__ void s3_dma16(const char *base_uniform, unsigned byte_off, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(byte_off), "s"(base_uniform), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}

// ACT: 0 none, 1 LeakyReLU(slope).  SIDE: 1 adds R[ridx[m], :] before the activation (the per-point addend of block1.0).
// DBG: probe-only ablation bits (1 no activation loads, 2 one fragment address per k step, 4 no weight copies, 8 no stores, 16 no barriers)
//
// Two 256-thread workgroups per CU (one wave of each per SIMD): they run out of phase, so one workgroup's barrier waits,
// operand-split bursts and store tails are covered by the other's MFMAs.  Per workgroup: 128-row tiles, the weights stream
// through a ring of three 24-KiB k-step chunks (chunk n+2 is issued at the top of step n), the activations of K group g+1 are
// loaded into registers at the top of group g.
template <int ACT, int SIDE, int DBG = 0>
__global__ __launch_bounds__(256, 2) void linear_s3_kernel(const float *__restrict__ A, int lda, const char *__restrict__ W3,
                                                           const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                           int M, int K, int G, float slope, const float *__restrict__ R,
                                                           const int32_t *__restrict__ ridx, int ldr)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int n_tiles = (M + S3_ROWS - 1) / S3_ROWS;
    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const bool kmask = (K & 31) != 0;
    const int steps_per_tile = 2 * G;
    const int my_tiles = (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * steps_per_tile;                     // k steps this workgroup runs

    // weights of k step n (ring slot n % 3): 24 KiB = 6 pieces of 1 KiB per wave.  Step n of this workgroup is k step n % (2 G) of the image.
    auto issue_w = [&](int n, int slot) {
        if ((DBG & 4) && n >= 3) return;
        const char *wsrc = W3 + (size_t)(n % steps_per_tile) * S3_STEP_BYTES;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned piece = (unsigned)(wave_s * 6 + i) * 1024u;
            s3_dma16(wsrc, piece + (unsigned)lane * 16u, lds0 + (unsigned)slot * S3_STEP_BYTES + piece);
        }
    };
    // activations: lane (j, h) owns k = 32 g + 16 h + 0..15 of row j
    float4 raw[4];
    auto issue_a = [&](int t, int g) {
        if ((DBG & 1) && !(t == (int)blockIdx.x && g == 0)) return;
        int row = t * S3_ROWS + wave * 32 + j;
        if (row >= M) row = M - 1;
        const int k0 = 32 * g + 16 * h;
        const float *src = A + (size_t)row * lda + k0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            raw[q] = (k0 + 4 * q + 4 <= lda) ? *reinterpret_cast<const float4 *>(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    issue_a(tile, 0);
    issue_w(0, 0);
    if (total > 1) issue_w(1, 1);
    int n = 0, slot = 0;
    long long tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, t_start = 0, w_start = 0;
    if (DBG & 32) { t_start = t_prev = clock64(); w_start = wall_clock64(); }
#define S3_STAMP(i_) do { if (DBG & 32) { const long long t_ = clock64(); tm[i_] += t_ - t_prev; t_prev = t_; } } while (0)
    for (; tile < n_tiles; tile += gridDim.x) {
        f32x16 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        for (int g = 0; g < G; ++g) {
            u32x4 ap[2][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks, ++n) {
                // chunk n was issued two steps ago; younger in the queue: chunk n+1 (6 pieces) and, ahead of it when it was
                // issued at a ks = 0 step, the 4 activation loads of the next group
                if (n + 1 < total) { if (ks == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                S3_STAMP(0);
                if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
                S3_STAMP(1);      // chunk n has landed for every wave; slot (n+2) % 3 is no longer read
                const char *fbp = lds + lane * 16 + slot * S3_STEP_BYTES;
                u32x4 wf[2][2][3];                                   // fragments of one column-tile PAIR: [tile][plane]; the next pair is always in flight
                auto read_pair = [&](int sl, int p) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[sl][t][pl] = *reinterpret_cast<const u32x4 *>(fbp + ((((DBG & 2) ? 0 : 2 * p + t)) * 3 + pl) * 1024);
                };
                read_pair(0, 0);
                if (ks == 0) {
                    // split this group's 16 activations per lane under the latency of those reads, then fetch the next group's
                    float4 cur[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) cur[q] = raw[q];
                    if (kmask && g == G - 1) {
                        const int k0 = 32 * g + 16 * h;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (k0 + 4 * q + 0 >= K) cur[q].x = 0.f;
                            if (k0 + 4 * q + 1 >= K) cur[q].y = 0.f;
                            if (k0 + 4 * q + 2 >= K) cur[q].z = 0.f;
                            if (k0 + 4 * q + 3 >= K) cur[q].w = 0.f;
                        }
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        unsigned sh[4], sm[4], sl[4];
                        split2(cur[2 * k2].x, cur[2 * k2].y, sh[0], sm[0], sl[0]);
                        split2(cur[2 * k2].z, cur[2 * k2].w, sh[1], sm[1], sl[1]);
                        split2(cur[2 * k2 + 1].x, cur[2 * k2 + 1].y, sh[2], sm[2], sl[2]);
                        split2(cur[2 * k2 + 1].z, cur[2 * k2 + 1].w, sh[3], sm[3], sl[3]);
                        ap[k2][0] = u32x4{sh[0], sh[1], sh[2], sh[3]};
                        ap[k2][1] = u32x4{sm[0], sm[1], sm[2], sm[3]};
                        ap[k2][2] = u32x4{sl[0], sl[1], sl[2], sl[3]};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    S3_STAMP(5);
                    int t2 = tile, g2 = g + 1;
                    if (g2 == G) { g2 = 0; t2 += gridDim.x; }
                    if (t2 < n_tiles) issue_a(t2, g2);
                    S3_STAMP(6);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    int s2 = slot + 2; if (s2 >= 3) s2 -= 3;
                    if (n + 2 < total) issue_w(n + 2, s2);
                }
                __builtin_amdgcn_sched_barrier(0);
                S3_STAMP(2);
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, ap[ks][0]), Am = __builtin_bit_cast(bf16x8, ap[ks][1]), Al = __builtin_bit_cast(bf16x8, ap[ks][2]);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (p + 1 < 4) read_pair((p + 1) & 1, p + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const int c0 = 2 * p, c1 = 2 * p + 1;
#define W_(t, pl) __builtin_bit_cast(bf16x8, wf[p & 1][t][pl])
                    // smallest terms first; the two column tiles alternate so that no MFMA waits for its predecessor
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, W_(0, 0), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, W_(1, 0), acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(0, 2), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(1, 2), acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, W_(0, 1), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, W_(1, 1), acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, W_(0, 0), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, W_(1, 0), acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(0, 1), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(1, 1), acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(0, 0), acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, W_(1, 0), acc[c1], 0, 0, 0);
#undef W_
                    __builtin_amdgcn_sched_barrier(0);
                }
                S3_STAMP(3);
                if (++slot == 3) slot = 0;
            }
        }
        // epilogue.  MFMA tile c, lane column j carries output column n = 8 j + c (the weight rows are dealt to the tiles that
        // way at pack time), so for each of its 16 rows i = (r & 3) + 8 (r >> 2) + 4 h a lane owns the 8 ADJACENT columns
        // 8 j .. 8 j + 7: two 16-B stores per row, and the 32 lanes of a half write one whole 1-KiB row.
        const int row0 = tile * S3_ROWS + wave * 32;
        int my_ridx = 0;
        if (SIDE) { const int rr = row0 + j; my_ridx = ridx[rr < M ? rr : M - 1]; }
        float4 b0 = *reinterpret_cast<const float4 *>(bias + 8 * j), b1 = *reinterpret_cast<const float4 *>(bias + 8 * j + 4);
        if (DBG & 32) { asm volatile("" : "+v"(b0.x), "+v"(b1.x)); S3_STAMP(7); }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = row0 + i;
            float4 v0 = make_float4(acc[0][r] + b0.x, acc[1][r] + b0.y, acc[2][r] + b0.z, acc[3][r] + b0.w);
            float4 v1 = make_float4(acc[4][r] + b1.x, acc[5][r] + b1.y, acc[6][r] + b1.z, acc[7][r] + b1.w);
            if (SIDE) {
                const int src = __shfl(my_ridx, i, 64);
                const float4 r0 = *reinterpret_cast<const float4 *>(R + (size_t)src * ldr + 8 * j);
                const float4 r1 = *reinterpret_cast<const float4 *>(R + (size_t)src * ldr + 8 * j + 4);
                v0.x += r0.x; v0.y += r0.y; v0.z += r0.z; v0.w += r0.w;
                v1.x += r1.x; v1.y += r1.y; v1.z += r1.z; v1.w += r1.w;
            }
            if (ACT) {
                v0.x = v0.x > 0.f ? v0.x : v0.x * slope; v0.y = v0.y > 0.f ? v0.y : v0.y * slope;
                v0.z = v0.z > 0.f ? v0.z : v0.z * slope; v0.w = v0.w > 0.f ? v0.w : v0.w * slope;
                v1.x = v1.x > 0.f ? v1.x : v1.x * slope; v1.y = v1.y > 0.f ? v1.y : v1.y * slope;
                v1.z = v1.z > 0.f ? v1.z : v1.z * slope; v1.w = v1.w > 0.f ? v1.w : v1.w * slope;
            }
            if (row < M && (!(DBG & 8) || v0.x == 123.456f)) {
                float *dst = C + (size_t)row * ldc + 8 * j;
                *reinterpret_cast<float4 *>(dst) = v0;
                *reinterpret_cast<float4 *>(dst + 4) = v1;
            }
        }
        S3_STAMP(4);
    }
    if ((DBG & 32) && blockIdx.x == 0 && lane == 0) {
        long long *o = reinterpret_cast<long long *>(const_cast<float *>(R)) + wave * 16;
        for (int i = 0; i < 8; ++i) o[i] = tm[i];
        o[8] = clock64() - t_start; o[9] = wall_clock64() - w_start; o[10] = total;
    }
#undef S3_STAMP
}

// W[N,K] fp32 -> split bf16 fragments; bias -> padded fp32[256]
__global__ void pack_s3_kernel(const float *__restrict__ W, const float *__restrict__ bias, int N, int K, int G,
                               unsigned short *__restrict__ W3, float *__restrict__ bias_p)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one (g, ks, c, lane, e) per thread
    const int64_t total = (int64_t)G * 2 * 8 * 64 * 8;
    if (i < 256) bias_p[i] = (i < N && bias) ? bias[i] : 0.f;
    if (i >= total) return;
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63), c = (int)((i >> 9) & 7), ks = (int)((i >> 12) & 1), g = (int)(i >> 13);
    const int n = 8 * (l & 31) + c, k = 32 * g + 16 * (l >> 5) + 8 * ks + e;     // tile c, lane column j <-> output column 8 j + c
    const float x = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    auto rne = [](float v) -> unsigned {                 // fp32 -> bf16 bits, round to nearest even (finite inputs)
        unsigned u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    const unsigned hb = rne(x);
    const float r1 = x - __uint_as_float(hb << 16);
    const unsigned mb = rne(r1);
    const float r2 = r1 - __uint_as_float(mb << 16);
    const unsigned lb = rne(r2);
    const size_t frag = ((size_t)((g * 2 + ks) * 8 + c) * 3) * 512 + (size_t)l * 8 + e;     // in bf16 elements; planes 512 apart
    W3[frag] = (unsigned short)hb;
    W3[frag + 512] = (unsigned short)mb;
    W3[frag + 1024] = (unsigned short)lb;
}

}  // namespace hnr

using namespace hnr;

extern "C" int64_t hnr_linear_s3_packed_bytes(int N, int K)
{
    if (N <= 0 || N > 256 || K <= 0) return -1;
    return (int64_t)((K + 31) / 32) * S3_GROUP_BYTES;
}

extern "C" int hnr_linear_s3_pack(const float *d_W, const float *d_bias, int N, int K, void *d_W3, float *d_bias_p, void *stream)
{
    if (!d_W || !d_W3 || !d_bias_p || N <= 0 || N > 256 || K <= 0) { set_error("hnr_linear_s3_pack: bad argument (N must be <= 256)"); return HNR_ERR_BADARG; }
    const int G = (K + 31) / 32;
    const int64_t total = (int64_t)G * 2 * 8 * 64 * 8;
    pack_s3_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(d_W, d_bias, N, K, G, (unsigned short *)d_W3, d_bias_p);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_linear_s3(const float *d_A, int lda, const void *d_W3, const float *d_bias_p, const float *d_R,
                             const int32_t *d_ridx, int ldr, float *d_C, int ldc, int M, int N, int K, int act, float slope,
                             void *stream)
{
    if (M < 0 || N != 256 || K <= 0 || lda < K || (lda & 3) || ldc < N || (act != 0 && act != 1) || (d_R && (!d_ridx || ldr < N || (ldr & 3) || ((uintptr_t)d_R & 15))) || (ldc & 3) || ((uintptr_t)d_C & 15)) {
        set_error("hnr_linear_s3: bad sizes (M=%d N=%d K=%d lda=%d ldc=%d act=%d; N must be 256, lda / ldc / ldr multiples of 4, 16-B aligned)", M, N, K, lda, ldc, act);
        return HNR_ERR_BADARG;
    }
    if (M == 0) return HNR_OK;
    if (!d_A || !d_W3 || !d_bias_p || !d_C || ((uintptr_t)d_A & 15)) { set_error("hnr_linear_s3: NULL or unaligned pointer"); return HNR_ERR_BADARG; }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int n_tiles = cdiv(M, S3_ROWS), G = (K + 31) / 32;
    const int grid = n_tiles < 2 * n_cu ? n_tiles : 2 * n_cu;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 3 * S3_STEP_BYTES;
    const char *w3 = (const char *)d_W3;
#define HNR_S3_LAUNCH(ACT_, SIDE_) linear_s3_kernel<ACT_, SIDE_><<<grid, 256, lds, st>>>(d_A, lda, w3, d_bias_p, d_C, ldc, M, K, G, slope, d_R, d_ridx, ldr)
#ifdef HNR_LINEAR_PROBE
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("HNR_S3_DBG"); dbg = e ? atoi(e) : 0; }
#define HNR_S3_ABL(X_) case X_: linear_s3_kernel<1, 0, X_><<<grid, 256, lds, st>>>(d_A, lda, w3, d_bias_p, d_C, ldc, M, K, G, slope, d_R, d_ridx, ldr); break;
    if (dbg == 32 && act && !d_R) {
        static long long *d_dbg = nullptr;
        if (!d_dbg && hipMalloc(&d_dbg, 4 * 16 * sizeof(long long)) != hipSuccess) return HNR_ERR_HIP;
        linear_s3_kernel<1, 0, 32><<<grid, 256, lds, st>>>(d_A, lda, w3, d_bias_p, d_C, ldc, M, K, G, slope, reinterpret_cast<const float *>(d_dbg), d_ridx, ldr);
        long long hh[64];
        if (hipMemcpy(hh, d_dbg, sizeof(hh), hipMemcpyDeviceToHost) != hipSuccess) return HNR_ERR_HIP;
        int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
        for (int w = 0; w < 4; ++w) {
            const long long *o = hh + 16 * w;
            const double ns = (double)o[10], nt = ns / (2 * G);
            fprintf(stderr, "[s3 dbg32] wave %d: %lld k steps, %lld cycles at %.3f GHz; per k step: vmcnt wait %.0f, barrier %.0f, first reads + split (per group) %.0f, activation loads (per group) %.0f, weight copies %.0f, 48 MFMAs %.0f; per tile: bias wait %.0f, stores %.0f\n", w,
                    o[10], o[8], (double)o[8] / ((double)o[9] / (wall_khz * 1e3)) / 1e9, o[0] / ns, o[1] / ns, o[5] / (ns / 2), o[6] / (ns / 2), o[2] / ns, o[3] / ns, o[7] / nt, o[4] / nt);
        }
    } else if (dbg && act && !d_R) {
        switch (dbg) { HNR_S3_ABL(1) HNR_S3_ABL(2) HNR_S3_ABL(4) HNR_S3_ABL(8) HNR_S3_ABL(16) HNR_S3_ABL(3) HNR_S3_ABL(11) HNR_S3_ABL(15) HNR_S3_ABL(31) default: break; }
    } else
#endif
    if (d_R) { if (act) HNR_S3_LAUNCH(1, 1); else HNR_S3_LAUNCH(0, 1); }
    else { if (act) HNR_S3_LAUNCH(1, 0); else HNR_S3_LAUNCH(0, 0); }
#undef HNR_S3_LAUNCH
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
