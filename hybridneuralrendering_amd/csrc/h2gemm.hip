// Dense layers of the TRAINING step on the 16-bit matrix pipe, fp32 in / fp32 out in HBM, "f16x2" arithmetic (hnr_h2.h: every operand split
// into two fp16 terms under an exact power-of-two scale, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation).  What torch
// autograd derives from the nn.Linear (+ LeakyReLU) layers of PointAggregator.viewmlp (models/aggregators/point_aggregators.py:948 block1,
// :972 block3, :1037 color_feature_branch, :1199 aux_merge_weight_block, :1292 color_mixup_block):
//
//   h2lin_kernel    C[M,N] = epi(A[M,K] W^T): forward layers and INPUT gradients (dZ_prev = (dZ W) * LeakyReLU'(Y_prev) with W^T packed as
//                   the weight).  Rows are scaled per row (the reduction runs over a row), weights per layer, like csrc/chain.hip.
//   h2wgrad_kernel  dW[N,K] = dZ[M,N]^T X[M,K], db[N] = column sums of dZ: WEIGHT gradients.  The reduction runs over the ROW index here,
//                   so both operands reach the MFMA transposed: fp32 rows -> fp16 planes, row-major in LDS -> ds_read_b64_tr_b16 (gfx950's
//                   transposing LDS read) -> fragments with 8 consecutive rows per lane.  A reduction over rows rules out per-row scales:
//                   each operand is scaled by ONE power of two from its maximum |value| (left on the device by the kernel that produced it);
//                   a value v then carries an absolute error <= 2^-40 max|.| (fp16 subnormal quantum of the low plane), which over the
//                   3e5-row sums of a training batch stays below the fp32 accumulation error of the products themselves
//                   (tests/test_h2gemm_gpu.py measures both against fp64).  Deterministic: fixed-order partials + a fixed-order reduction.
//
// Every row count is read on the device (*d_m), so the training step needs no host synchronisation (the reference's torch autograd
// sizes every gradient tensor from host-side shapes).
#include <stdlib.h>
#include <type_traits>

#include "chain_defs.h"

namespace hnr {

constexpr int HL_META_FLOATS = 256 + 4;            // bias[256], descale, max|W| bits, pad
constexpr int HL_DESC = 256, HL_WMAX = 257;

// ------------------------------------------------------------------------------------------------------------------------ packing (batched)
// job j: W element (n, k) = W[n * rs + k * cs] (cs = 1, rs = ld: the nn.Linear weight itself; rs = 1, cs = ld: its transpose, for input
// gradients), N <= 256 output columns, K <= 288 inputs, image = [k step][column tile 8][plane 2][64 lanes][16 B] + meta.
struct H2PackJob { const float *W; long long rs, cs; int N, K; const float *bias; char *out; };
constexpr int H2_MAX_JOBS = 16;
struct H2PackArgs { H2PackJob job[H2_MAX_JOBS]; };

__global__ void h2_meta_zero_kernel(H2PackArgs a)
{
    const int j = threadIdx.x;
    if (j < H2_MAX_JOBS && a.job[j].W) reinterpret_cast<unsigned *>(a.job[j].out + (size_t)((a.job[j].K + 15) / 16) * CH_WSTEP)[HL_WMAX] = 0u;
}

__global__ void h2_wmax_kernel(H2PackArgs a)
{
    const H2PackJob &jb = a.job[blockIdx.y];
    if (!jb.W) return;
    const int S = (jb.K + 15) / 16;
    unsigned *wmax = reinterpret_cast<unsigned *>(jb.out + (size_t)S * CH_WSTEP) + HL_WMAX;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.N * jb.K; i += gridDim.x * blockDim.x) {
        const int n = i / jb.K, k = i - n * jb.K;
        m = fmaxf(m, fabsf(jb.W[(long long)n * jb.rs + (long long)k * jb.cs]));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(wmax, __float_as_uint(m));
}

__global__ void h2_pack_kernel(H2PackArgs a)
{
    const H2PackJob &jb = a.job[blockIdx.y];
    if (!jb.W) return;
    const int S = (jb.K + 15) / 16;
    float *meta = reinterpret_cast<float *>(jb.out + (size_t)S * CH_WSTEP);
    const unsigned maxbits = reinterpret_cast<const unsigned *>(meta)[HL_WMAX];
    int ex = (int)((maxbits >> 23) & 0xffu);
    ex = ex < 110 ? 110 : (ex > 160 ? 160 : ex);
    const int sw = CH_W_EXP + 126 - ex;
    const float scale = pow2f(sw);
    const int total = S * 8 * 64 * 8;                                          // (s, ct, lane, e)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 7, ln = (i >> 3) & 63, ct = (i >> 9) & 7, s = i >> 12;
        const int ii = ln & 31, hh = ln >> 5;
        const int n = 32 * ct + 16 * ((ii >> 2) & 1) + (ii & 3) + 4 * (ii >> 3), k = 16 * s + 8 * hh + e;
        const float x = (n < jb.N && k < jb.K) ? __fmul_rn(jb.W[(long long)n * jb.rs + (long long)k * jb.cs], scale) : 0.f;
        const _Float16 hv = (_Float16)x;
        const _Float16 mv = (_Float16)__fsub_rn(x, (float)hv);
        _Float16 *dst = reinterpret_cast<_Float16 *>(jb.out + (size_t)s * CH_WSTEP + (ct * 2) * 1024 + ln * 16) + e;
        dst[0] = hv;
        dst[512] = mv;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) meta[i] = (jb.bias && i < jb.N) ? jb.bias[i] : 0.f;
        if (threadIdx.x == 0) meta[HL_DESC] = pow2f(-sw);
    }
}

// ------------------------------------------------------------------------------------------------------------------------ h2lin
struct H2LinArgs {
    const float *A; int lda;
    const long long *d_m; long long M_cap;     // rows = min(M_cap, *d_m) (d_m may be NULL)
    int n_seg; long long seg_stride;           // n_seg > 1: the rows are n_seg SEGMENTS of min(*d_m, seg_stride) rows each, segment v starting at physical row v * seg_stride
    const char *wimg;
    int N, K;
    int mode;                                  // 0: C = act(A W^T + bias); 1: C = (A W^T) * (side > 0 ? 1 : slope)
    int act;                                   // mode 0: LeakyReLU after the bias
    float slope;
    const float *side; int lds_;               // mode 1: stored forward activation [M, lds_]
    const uint32_t *side_bits;                 // mode 1, optional: its signs instead -- one word per (32-row tile, wave, lane) of the chain kernels' layout, bit 31 - i = (value i of the
                                               // lane's 32 columns col0 + 32 c + 0..15 is > 0); 1/32 of the bytes and one load per row tile and lane instead of eight
    float *C; int ldc;
    unsigned *absmax;                          // optional: max |C| (bit pattern, atomicMax)
    int spread;                                // narrow layers: the (row tile, column tile) pairs dealt to all four waves (HNR_H2LIN_SPREAD=0: wave w = columns 64 w .. only)
};

// 64-row tiles, two workgroups per CU (one's row loads / split / epilogue under the other's MFMAs); wave w owns output columns 64 w .. + 63.
template <int S>
__global__ __launch_bounds__(256, 2) void h2lin_kernel(H2LinArgs a)
{
    constexpr int RT = 2, SLOT = RT * 2048 + 32, ROWS = 32 * RT;             // (+ 32 B between the k steps' slots: the prologue's plane stores of one row spread over the banks)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    long long M = a.M_cap, n_unit = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    if (a.n_seg > 1) { n_unit = M < a.seg_stride ? M : a.seg_stride; M = n_unit * a.n_seg; }
    // logical row -> physical row (segments: at most 7 compares, no integer division)
    auto phys = [&](long long m) -> long long {
        if (a.n_seg <= 1) return m;
        int q = 0;
        for (int v = 1; v < a.n_seg && v < 8; ++v) q += (m >= (long long)v * n_unit) ? 1 : 0;
        return (long long)q * a.seg_stride + (m - (long long)q * n_unit);
    };
    const int n_tiles = (int)((M + ROWS - 1) / ROWS);
    const float *meta = reinterpret_cast<const float *>(a.wimg + (size_t)S * CH_WSTEP);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, S * CH_WSTEP, 0x00020000);
    float *rowinv = reinterpret_cast<float *>(lds + S * SLOT);                 // [ROWS]
    const float dw = meta[HL_DESC];
    float gmax = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row_base = (long long)tile * ROWS;
        // ---- prologue: fp32 rows -> per-row power-of-two scale -> fp16 (h, m) planes in MFMA fragment order (as csrc/mlp.hip)
        {
            constexpr int COLS = 16 * S, LPR = COLS > 128 ? 64 : (COLS > 64 ? 32 : 16), RPI = 64 / LPR, NB = (COLS + 4 * LPR - 1) / (4 * LPR);
            int lane_t = lane;                                                 // laundered per tile: hoisted out of the tile loop, the per-lane source / destination
            asm volatile("" : "+v"(lane_t));                                   // offsets below become dozens of live registers (spills)
            const int lr = lane_t % LPR, sub = lane_t / LPR;
            constexpr int RW = 8 * RT;                                         // rows of this wave
            constexpr int BATCH = RW / RPI;
            float4 v[BATCH][NB];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int rl = RW * wave + b * RPI + sub;
                long long row = row_base + rl;
                if (row >= M) row = M - 1;
                const float *src = a.A + (size_t)phys(row) * a.lda;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = 4 * (nb * LPR + lr);
                    v[b][nb] = *reinterpret_cast<const float4 *>(src + (c + 4 <= a.lda ? c : 0));
                }
            }
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int rl = RW * wave + b * RPI + sub;
                float m = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = 4 * (nb * LPR + lr);
                    float *t = reinterpret_cast<float *>(&v[b][nb]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { t[e] = (c + e < a.K) ? t[e] : 0.f; m = fmaxf(m, fabsf(t[e])); }
                }
                m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xf, 0xf, false));
                m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xf, 0xf, false));
                m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x141, 0xf, 0xf, false));
                m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x140, 0xf, 0xf, false));
                if (LPR >= 32) m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x142, 0xa, 0xf, false));
                if (LPR >= 64) m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x143, 0xc, 0xf, false));
                if (LPR == 64) m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
                else if (LPR == 32) { const float m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31)), m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63)); m = sub ? m1 : m0; }
                const int k = row_scale_exp(m);
                const float sc = pow2f(k);
                if (lr == 0) rowinv[rl] = pow2f(-k);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = 4 * (nb * LPR + lr);
                    if (c < COLS) {
                        unsigned ph0, pm0, ph1, pm1;
                        split2h(__fmul_rn(v[b][nb].x, sc), __fmul_rn(v[b][nb].y, sc), ph0, pm0);
                        split2h(__fmul_rn(v[b][nb].z, sc), __fmul_rn(v[b][nb].w, sc), ph1, pm1);
                        char *dst = lds + (c >> 4) * SLOT + ((rl >> 5) * 2) * 1024 + ((((c >> 3) & 1) * 32 + (rl & 31)) * 16) + (c & 7) * 2;
                        *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
                        *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(pm0, pm1);
                    }
                }
            }
        }
        __syncthreads();
        // Which (row tile, column tile) pairs a wave multiplies.  Wide layers (N > 128): wave w = the 64 columns 64 w .. of both row tiles.  Narrower ones
        // left waves idle through the MFMA loop AND the epilogue (N <= 64: three of four) -- the narrow layers' input gradients of a training step are
        // ~15 launches of 20 - 65 us on the main queue, 2.5 - 3x their bytes' time --: N <= 128: wave w = column tile w of both row tiles; N <= 64:
        // wave w = column tile w & 1 of row tile w >> 1.  Same arithmetic per element.
        auto compute = [&](auto rtw_c, auto nct_c, int rt0, int ct0) __attribute__((always_inline)) {
            constexpr int RTW = decltype(rtw_c)::value, NCT = decltype(nct_c)::value;
            const int colb = 32 * ct0 + 16 * h;                                // this lane's columns of the wave's column tile c: colb + 32 c + 0..15
            const unsigned wo = (unsigned)ct0 * 2048u + (unsigned)lane * 16u;
            float inv[RTW];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) inv[rt] = __fmul_rn(rowinv[32 * (rt0 + rt) + j], dw);
            f32x16 acc[RTW][NCT];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][c][r] = 0.f;
            h2_mfma_layer<RTW, NCT, S, 0, CH_WSTEP, SLOT>(wsrd, 0, wo, lds + rt0 * 2048, lane, acc, []() {});
            // mode 1: the rows' stored activations, one row tile ahead of its use (the weight fragments' registers are free by now)
            float4 sd[2][NCT][4];
            unsigned sbits[2] = {0u, 0u};
            const bool use_bits = a.side_bits != nullptr;                      // (N = 256 only: the wave owns the chain kernels' 64 columns)
            auto load_side = [&](int rt) {
                if (use_bits) { sbits[rt & 1] = a.side_bits[(((row_base + 32 * (rt0 + rt)) >> 5) * 4 + wave) * 64 + lane]; return; }
                long long row = row_base + 32 * (rt0 + rt) + j;
                if (row >= M) row = M - 1;
                const float *srow = a.side + (size_t)phys(row) * a.lds_ + colb;
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sd[rt & 1][c][q] = (colb + 32 * c + 4 * q + 4 <= a.lds_) ? *reinterpret_cast<const float4 *>(srow + 32 * c + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
            };
            if (a.mode == 1) load_side(0);
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                const long long row = row_base + 32 * (rt0 + rt) + j;
                if (a.mode == 1 && rt + 1 < RTW) load_side(rt + 1);
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    float o[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float bq[4] = {0.f, 0.f, 0.f, 0.f};
                        if (a.mode == 0) { const float4 b4 = *reinterpret_cast<const float4 *>(meta + colb + 32 * c + 4 * q); bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w; }
                        const float *sq = reinterpret_cast<const float *>(&sd[rt & 1][c][q]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaf(acc[rt][c][4 * q + e], inv[rt], bq[e]);
                            if (a.mode == 0) { if (a.act) v = fmaxf(v, __fmul_rn(v, a.slope)); }
                            else if (use_bits) v = __fmul_rn(v, ((sbits[rt & 1] >> (31 - (16 * c + 4 * q + e))) & 1u) ? 1.f : a.slope);
                            else v = __fmul_rn(v, sq[e] > 0.f ? 1.f : a.slope);
                            o[4 * q + e] = v;
                            gmax = fmaxf(gmax, fabsf(v));
                        }
                    }
                    if (row < M) {
                        float *dst = a.C + (size_t)phys(row) * a.ldc + colb + 32 * c;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (colb + 32 * c + 4 * q + 4 <= a.ldc && colb + 32 * c + 4 * q < ((a.N + 3) & ~3))
                                *reinterpret_cast<float4 *>(dst + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                    }
                }
            }
        };
        if (a.N > 128 || !a.spread) { if (64 * wave < a.N) compute(std::integral_constant<int, RT>{}, std::integral_constant<int, 2>{}, 0, 2 * wave); }
        else if (a.N > 64) { if (32 * wave < a.N) compute(std::integral_constant<int, RT>{}, std::integral_constant<int, 1>{}, 0, wave); }
        else if (32 * (wave & 1) < a.N) compute(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, wave >> 1, wave & 1);
        __syncthreads();                                                       // the planes are rewritten by the next tile's prologue
    }
    if (a.absmax) {
        // rows past M repeat row M - 1 (values that exist anyway); padded columns are exact zeros.  One atomic per workgroup.
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        float *s_m = rowinv;                                                   // (free after the last tile's barrier)
        if (lane == 0) s_m[wave] = gmax;
        __syncthreads();
        if (tid == 0) { gmax = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])); if (gmax > 0.f) atomicMax(a.absmax, __float_as_uint(gmax)); }
    }
}

// ------------------------------------------------------------------------------------------------------------------------ absmax
__global__ __launch_bounds__(256) void h2_absmax_kernel(const float *__restrict__ A, int lda, const long long *__restrict__ d_m, long long M_cap, int N,
                                                        unsigned *__restrict__ out, int n_seg, long long seg_stride)
{
    long long M = M_cap, n_unit = M_cap;
    if (d_m) { const long long c = *d_m; if (c < M) M = c; }
    if (n_seg > 1) { n_unit = M < seg_stride ? M : seg_stride; M = n_unit * n_seg; }
    // 64 / LPR rows per wave pass, LPR lanes x float4 per row pass
    const int n4 = (N + 3) >> 2, LPR = n4 >= 64 ? 64 : (n4 > 16 ? 32 : (n4 > 8 ? 16 : 8)), RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, lr = lane % LPR, sub = lane / LPR;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    float m = 0.f;
    for (long long r0 = wave * RPW; r0 < M; r0 += n_waves * RPW) {
        long long row = r0 + sub;
        if (row >= M) continue;
        if (n_seg > 1) { int q = 0; for (int v = 1; v < n_seg && v < 8; ++v) q += (row >= (long long)v * n_unit) ? 1 : 0; row = (long long)q * seg_stride + (row - (long long)q * n_unit); }
        const float *p = A + (size_t)row * lda;
        for (int c4 = lr; c4 < n4; c4 += LPR) {
            const int c = 4 * c4;
            if (c + 4 <= lda) {
                const float4 v = *reinterpret_cast<const float4 *>(p + c);
                m = fmaxf(m, fabsf(v.x));
                if (c + 1 < N) m = fmaxf(m, fabsf(v.y));
                if (c + 2 < N) m = fmaxf(m, fabsf(v.z));
                if (c + 3 < N) m = fmaxf(m, fabsf(v.w));
            } else {
                for (int e = 0; e < 4; ++e) if (c + e < N) m = fmaxf(m, fabsf(p[c + e]));
            }
        }
    }
    // one atomic per workgroup (thousands of same-address atomics cost ~12 ns each)
    __shared__ float s_m[4];
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])); if (m > 0.f) atomicMax(out, __float_as_uint(m)); }
}

// ------------------------------------------------------------------------------------------------------------------------ h2wgrad
struct H2WgradArgs {
    const float *dZ; int ldz;                  // [M, ldz], N columns used
    const float *X; int ldx;                   // [M, ldx], K columns used
    const long long *d_m; long long M_cap;
    int n_seg; long long seg_stride;           // as in H2LinArgs
    int N, K;
    const unsigned *zmax, *xmax;               // bit patterns of max |dZ|, max |X| (device)
    float *partial;                            // [gridDim.x][NP][KP] fixed-order partials; column K = db (the staged X rows carry a 1 there)
    int dbg;                                   // probe builds only (HNR_WG_DBG): 1 no MFMAs, 2 no conversion, 4 no global loads
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 hl_fp16x4_t;

// One transposed fragment (32 x 16, 8 consecutive ROWS per lane) of a row-major fp16 plane in LDS: two ds_read_b64_tr_b16, each handing a
// 16-lane group a [4 rows][16 columns] block column-major (lane i of the group supplies the address of row i >> 2, columns 4 (i & 3) .. + 3,
// and receives column i of the four rows).  plane: LDS base of the 32-row tile; rs: row stride in bytes; col0: first column of the 32.
__device__ __forceinline__ f16x8 h2_tr_frag(const char *plane, int rs, int col0, int kstep, int lane)
{
    const int i = lane & 15, q = lane >> 4;
    const char *p = plane + (size_t)(16 * kstep + 8 * (q >> 1) + (i >> 2)) * rs + (col0 + 16 * (q & 1) + 4 * (i & 3)) * 2;
#ifdef HNR_H2_NO_TR                                                            // reference form of the same read (16-bit gathers): debugging only
    const int n = col0 + (lane & 31), m0 = 16 * kstep + 8 * (lane >> 5);
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const _Float16 *>(plane + (size_t)(m0 + e) * rs + n * 2);
    (void)p;
    return r;
#else
    typedef __attribute__((address_space(3))) hl_fp16x4_t *lds_p;
    const unsigned pa = (unsigned)(uintptr_t)p;                               // an LDS pointer's low 32 bits are its LDS address
    const hl_fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(uintptr_t)(pa));
    const hl_fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_p)(uintptr_t)(pa + 4u * (unsigned)rs));
    f16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
#endif
}

// NT = 32-column tiles of dZ (N), KT = tiles of X (K + 1 columns: the bias column) covered by the workgroup; 8 waves = WN = NT (along N) x WK (along K).
// Pipeline over 16-row blocks (one MFMA k step): three LDS stages of fp16 planes, two register sets of fp32 rows.  In iteration i a wave issues
// the loads of block i + 2, runs the MFMAs of block i from its stage and -- between them -- converts block i + 1 (loaded one iteration ago) into
// the next stage: the conversion VALU work and the LDS writes ride under the wave's own MFMAs, the HBM latency under a whole iteration.
// The loop body is STRAIGHT-LINE code: every predicate is folded into an address or a select (a load or an MFMA inside a branch makes the
// compiler's wait-count pass wait for vmcnt(0), i.e. for the loads it has just issued: no lookahead, 1.9 TB/s).
// BIASV (K = 256: no spare column in 8 tiles, and a ninth tile costs 16 more accumulator registers per wave than the file has): the bias gradient
// is summed on the side in fp32 from the dZ rows as they pass through the staging registers.
// DEPTH = 1 (NT = 8, KT = 9: the largest accumulator set leaves no room for a second register set): one set, loaded at the top of the iteration
// for block i + 1 and converted in the iteration's last rounds.
template <int NT, int KT, int WK, int BIASV = 0, int DEPTH = 2>
__global__ __launch_bounds__(512, 1) void h2wgrad_kernel(H2WgradArgs a)
{
    constexpr int WN = NT;
    static_assert(WN * WK == 8, "8 waves");
    constexpr int KTW = (KT + WK - 1) / WK;
    constexpr int WZ = 32 * NT, WX = 32 * KT;                                  // columns staged per row
    constexpr int RSZ = ((WZ * 2 - 64 + 255) & ~255) + 64, RSX = ((WX * 2 - 64 + 255) & ~255) + 64;      // row strides in bytes: 64 (mod 256), so the 4 rows x 32 B of a transposed read hit distinct banks
    constexpr int RB = 16;                                                     // rows per block
    constexpr int PZ = RB * RSZ, PX = RB * RSX, STAGE = 2 * PZ + 2 * PX;
    constexpr int Z4 = WZ / 4, X4 = WX / 4, NLD = (RB * (Z4 + X4) + 511) / 512;      // float4 loads per thread per block
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave % WN, wk = wave / WN;
    long long M = a.M_cap, n_unit = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    if (a.n_seg > 1) { n_unit = M < a.seg_stride ? M : a.seg_stride; M = n_unit * a.n_seg; }
    const long long n_blocks = (M + RB - 1) / RB;
    if (n_blocks == 0 && blockIdx.x > 0) return;
    const int kz = row_scale_exp(__uint_as_float(*a.zmax)), kx = row_scale_exp(__uint_as_float(*a.xmax));
    const float sz = pow2f(kz), sx = pow2f(kx);

    f32x16 acc[KTW];
#pragma unroll
    for (int u = 0; u < KTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    // staging slots of this thread: slot it = float4 (row, columns c .. c + 3) of the dZ tile or of the X tile of a block; (row, c, tile) do not
    // depend on the block.  Per slot: the element offset inside a block's rows, the LDS offset of its planes, the masks of its four values.
    float4 stg[DEPTH][NLD];
    constexpr int NZS = BIASV ? (RB * Z4) / 512 : 0;                           // BIASV: the first NZS slots of every thread are dZ slots
    static_assert(!BIASV || (RB * Z4) % 512 == 0, "BIASV needs whole dZ slots");
    float bsum[NZS > 0 ? NZS : 1][4];
#pragma unroll
    for (int i = 0; i < (NZS > 0 ? NZS : 1); ++i) bsum[i][0] = bsum[i][1] = bsum[i][2] = bsum[i][3] = 0.f;
    int s_rc[NLD], s_lds[NLD];                                                 // row << 16 | first column read; LDS offset of the high plane
    unsigned s_keep[NLD];                                                      // bit e: value e is a real column (< N or < K); bit 4: the X tile; bit 8 + e: value e is the bias column
    const long long seg_extra = a.n_seg > 1 ? a.seg_stride - n_unit : 0;       // physical row = m + (segment of m) * seg_extra
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int idx = (tid + 512 * it < RB * (Z4 + X4)) ? tid + 512 * it : tid + 512 * (it - 1);      // no slot left: this thread repeats its previous one
        const bool has = true;
        const bool isx = idx >= RB * Z4;
        const int id2 = isx ? idx - RB * Z4 : idx, per = isx ? X4 : Z4;
        const int row = id2 / per, c = 4 * (id2 - row * per);
        const int lim = isx ? a.K : a.N, ld = isx ? a.ldx : a.ldz;
        const bool inrow = c + 4 <= ld;
        s_rc[it] = (row << 16) | (inrow ? c : 0);
        unsigned keep = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) keep |= (has && inrow && c + e < lim) ? (1u << e) : 0u;
        if (isx) { keep |= 16u; if (!BIASV && has && (a.K >> 2) == (c >> 2)) keep |= 256u << (a.K & 3); }
        s_keep[it] = keep;
        s_lds[it] = (isx ? 2 * PZ : 0) + row * (isx ? RSX : RSZ) + c * 2;
    }
    auto load_slot = [&](long long blk, int it, float4 &v) {
        const bool isx = (s_keep[it] & 16u) != 0;
        long long m = blk * RB + (s_rc[it] >> 16);
        m = m < M ? m : (M > 0 ? M - 1 : 0);                                    // rows past the end re-read the last row (masked when stored); M = 0: row 0 of the (capacity-sized) operands
        int q = 0;
#pragma unroll
        for (int sv = 1; sv < 8; ++sv) q += (sv < a.n_seg && m >= (long long)sv * n_unit) ? 1 : 0;
        const long long pm = m + (long long)q * seg_extra;
        const float *src = (isx ? a.X : a.dZ) + (size_t)pm * (isx ? a.ldx : a.ldz) + (s_rc[it] & 0xffff);
        v = *reinterpret_cast<const float4 *>(src);
    };
    auto store_slot = [&](char *base, long long blk, int it, const float4 &v) {
        const unsigned keep = (blk * RB + (s_rc[it] >> 16) < M) ? s_keep[it] : (s_keep[it] & ~15u);
        const float sc = (keep & 16u) ? sx : sz;
        if (BIASV && it < NZS) {
            bsum[it < NZS ? it : 0][0] += (keep & 1u) ? v.x : 0.f; bsum[it < NZS ? it : 0][1] += (keep & 2u) ? v.y : 0.f;
            bsum[it < NZS ? it : 0][2] += (keep & 4u) ? v.z : 0.f; bsum[it < NZS ? it : 0][3] += (keep & 8u) ? v.w : 0.f;
        }
        unsigned ph0, pm0, ph1, pm1;
        split2h((keep & 1u) ? __fmul_rn(v.x, sc) : 0.f, (keep & 2u) ? __fmul_rn(v.y, sc) : 0.f, ph0, pm0);
        split2h((keep & 4u) ? __fmul_rn(v.z, sc) : 0.f, (keep & 8u) ? __fmul_rn(v.w, sc) : 0.f, ph1, pm1);
        // column K of the staged X rows = 1 (fp16 1.0 in the high plane, unscaled): dW[:, K] becomes the column sums of dZ = db
        ph0 = (keep & 0x100u) ? ((ph0 & 0xffff0000u) | 0x3c00u) : (keep & 0x200u) ? ((ph0 & 0x0000ffffu) | 0x3c000000u) : ph0;
        ph1 = (keep & 0x400u) ? ((ph1 & 0xffff0000u) | 0x3c00u) : (keep & 0x800u) ? ((ph1 & 0x0000ffffu) | 0x3c000000u) : ph1;
        *reinterpret_cast<uint2 *>(base + s_lds[it]) = make_uint2(ph0, ph1);
        *reinterpret_cast<uint2 *>(base + s_lds[it] + ((keep & 16u) ? PX : PZ)) = make_uint2(pm0, pm1);
    };
    const long long step = gridDim.x;
    // one iteration: MFMAs over `cur` (block i) with the conversion of register set SET (block i + 1) into `nxt` spread between them; the loads of
    // block i + 2 go into the other set first.  Blocks past the end: the loads re-read the last rows, the conversion stores zeros -- harmless work
    // in the (at most two) tail iterations instead of branches in every iteration.
    auto body = [&](auto set_c, long long blk, const char *cur, char *nxt) {
        constexpr int SET = decltype(set_c)::value;
        const long long b1 = blk + step, b2 = blk + 2 * step;
#pragma unroll
        for (int it = 0; it < NLD; ++it) load_slot(DEPTH == 2 ? b2 : b1, it, stg[DEPTH == 2 ? (SET ^ 1) : 0][it]);
        const char *zb = cur, *xb = cur + 2 * PZ;
        const f16x8 zh = h2_tr_frag(zb, RSZ, 32 * wn, 0, lane), zm = h2_tr_frag(zb + PZ, RSZ, 32 * wn, 0, lane);
#pragma unroll
        for (int u0 = 0; u0 < KTW; u0 += 2) {
            // two X tiles per round: six MFMAs on two accumulators, alternating (a dependent MFMA waits for its predecessor's result); a wave whose
            // tile index runs past KT recomputes tile KT - 1 (its partial is not written)
            const int kt0 = wk * KTW + u0 < KT ? wk * KTW + u0 : KT - 1, kt1 = wk * KTW + u0 + 1 < KT ? wk * KTW + u0 + 1 : KT - 1;
            const f16x8 xh0 = h2_tr_frag(xb, RSX, 32 * kt0, 0, lane), xm0 = h2_tr_frag(xb + PX, RSX, 32 * kt0, 0, lane);
            if (u0 + 1 < KTW) {
                const f16x8 xh1 = h2_tr_frag(xb, RSX, 32 * kt1, 0, lane), xm1 = h2_tr_frag(xb + PX, RSX, 32 * kt1, 0, lane);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh0, acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh1, acc[u0 + 1], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm0, acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm1, acc[u0 + 1], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh0, acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh1, acc[u0 + 1], 0, 0, 0);
            } else {
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh0, acc[u0], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm0, acc[u0], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh0, acc[u0], 0, 0, 0);
            }
            // this round's share of the next block's conversion
            constexpr int R0 = (KTW + 1) / 2;
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                if (DEPTH == 2) { if (it * R0 / NLD == u0 / 2) store_slot(nxt, b1, it, stg[SET][it]); }
                else if (u0 + 2 >= KTW) store_slot(nxt, b1, it, stg[0][it]);      // one set: everything behind the last round (the loads had the whole iteration)
            }
        }
    };

    long long blk = blockIdx.x;
#pragma unroll
    for (int it = 0; it < NLD; ++it) load_slot(blk, it, stg[0][it]);
#pragma unroll
    for (int it = 0; it < NLD; ++it) store_slot(lds, blk, it, stg[0][it]);
    if (DEPTH == 2) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) load_slot(blk + step, it, stg[0][it]);
    }
    __syncthreads();
    int st_i = 0;                                                               // stage of the running block
    for (; blk < n_blocks; blk += 2 * step) {
        int nx = st_i == 2 ? 0 : st_i + 1;
        body(std::integral_constant<int, 0>{}, blk, lds + st_i * STAGE, lds + nx * STAGE);
        st_i = nx;
        __syncthreads();
        // (an odd block count: this second half then multiplies a stage of zeros -- the conversion of a block past the end)
        nx = st_i == 2 ? 0 : st_i + 1;
        body(std::integral_constant<int, 1>{}, blk + step, lds + st_i * STAGE, lds + nx * STAGE);
        st_i = nx;
        __syncthreads();
    }
    // ---- this workgroup's partial: [NP = 32 NT][KP], true units (scales removed); accumulator (reg r, lane): row n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    constexpr int NP = 32 * NT, KP = 32 * KT, LDP = KP + (BIASV ? 32 : 0);
    float *out = a.partial + (size_t)blockIdx.x * NP * LDP;
    const float dsz = pow2f(-kz), dsx = pow2f(-kx);
    if (BIASV) {
        // column sums of dZ: a thread's slot it covers row (tid + 512 it) / Z4 and columns 4 ((tid + 512 it) % Z4) .. + 3, the same columns in every block
        float *sb = reinterpret_cast<float *>(lds);                            // [RB rows][WZ columns] (the stages are free: the loop's last barrier is behind)
#pragma unroll
        for (int it = 0; it < NZS; ++it) {
            const int idx = tid + 512 * it, row = idx / Z4, c = 4 * (idx - row * Z4);
            *reinterpret_cast<float4 *>(sb + row * WZ + c) = make_float4(bsum[it][0], bsum[it][1], bsum[it][2], bsum[it][3]);
        }
        __syncthreads();
        if (tid < WZ) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RB; ++r) t += sb[r * WZ + tid];
            out[(size_t)tid * LDP + a.K] = t;                                   // (K = KP here)
        }
    }
#pragma unroll
    for (int u = 0; u < KTW; ++u) {
        const int kt = wk * KTW + u;
        if (kt >= KT) continue;
        const int kc = 32 * kt + (lane & 31);
        const float d2 = kc == a.K ? 1.0f : dsx;                               // the bias column was staged unscaled
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(size_t)n * LDP + kc] = __fmul_rn(__fmul_rn(acc[u][r], dsz), d2);
        }
    }
}

// The 256-wide weight gradient (NT = 8, KT = 9) with the fp32 rows staged through LDS by DMA.  h2wgrad_kernel<8, 9, 1, 0, 1> has room for ONE register
// set of staged rows beside its 144 accumulator registers, so a block's loads have one iteration of MFMAs (864 cycles) to arrive and the kernel ran at
// the latency of its loads: 3.2 us per 16-row block, 2.3 - 2.5 TB/s (32 KiB in flight per CU).  Here the rows of block i + 2 go global -> LDS (raw fp32,
// `global_load_lds_dwordx4`: no registers) while block i is multiplied, and block i + 1 is converted to its fp16 planes AFTER the MFMAs of the iteration:
// every block has nearly two iterations to land, two blocks (68 KiB) are in flight per CU.  Same staging arithmetic, the same MFMA order per
// accumulator and the same blocks per workgroup as h2wgrad_kernel: bit-identical partials.
__device__ __forceinline__ void h2_dma_row(const float *src, int voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void h2_dma_row8(const float *src, int voff, unsigned lds_dst)      // lanes 0..7 only (the 32 columns past the 256th)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xff\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds_dst) : "memory", "m0");
}
__global__ __launch_bounds__(512, 1) void h2wgrad_dma_kernel(H2WgradArgs a)
{
    constexpr int NT = 8, KT = 9, KTW = KT;
    constexpr int WZ = 32 * NT, WX = 32 * KT;
    constexpr int RSZ = ((WZ * 2 - 64 + 255) & ~255) + 64, RSX = ((WX * 2 - 64 + 255) & ~255) + 64;
    constexpr int RB = 16;
    constexpr int PZ = RB * RSZ, PX = RB * RSX, STAGE = 2 * PZ + 2 * PX;
    constexpr int RAWZ = RB * WZ * 4, RAWX = RB * WX * 4, RAW = RAWZ + RAWX;   // raw fp32 rows of a block: [16][256] dZ, then [16][288] X
    constexpr int Z4 = WZ / 4, X4 = WX / 4, NLD = (RB * (Z4 + X4) + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave;
    long long M = a.M_cap, n_unit = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    if (a.n_seg > 1) { n_unit = M < a.seg_stride ? M : a.seg_stride; M = n_unit * a.n_seg; }
    const long long n_blocks = (M + RB - 1) / RB;
    if (n_blocks == 0 && blockIdx.x > 0) return;
    const int kz = row_scale_exp(__uint_as_float(*a.zmax)), kx = row_scale_exp(__uint_as_float(*a.xmax));
    const float sz = pow2f(kz), sx = pow2f(kx);
    f32x16 acc[KTW];
#pragma unroll
    for (int u = 0; u < KTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    // conversion slots of this thread (as in h2wgrad_kernel): float4 (row, columns c .. c + 3) of the dZ or the X tile
    int s_row[NLD], s_raw[NLD], s_lds[NLD];
    unsigned s_keep[NLD];
    const long long seg_extra = a.n_seg > 1 ? a.seg_stride - n_unit : 0;
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int idx = (tid + 512 * it < RB * (Z4 + X4)) ? tid + 512 * it : tid + 512 * (it - 1);
        const bool isx = idx >= RB * Z4;
        const int id2 = isx ? idx - RB * Z4 : idx, per = isx ? X4 : Z4;
        const int row = id2 / per, c = 4 * (id2 - row * per);
        const int lim = isx ? a.K : a.N, ld = isx ? a.ldx : a.ldz;
        const bool inrow = c + 4 <= ld;
        s_row[it] = row;
        unsigned keep = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) keep |= (inrow && c + e < lim) ? (1u << e) : 0u;
        if (isx) { keep |= 16u; if ((a.K >> 2) == (c >> 2)) keep |= 256u << (a.K & 3); }
        s_keep[it] = keep;
        s_lds[it] = (isx ? 2 * PZ : 0) + row * (isx ? RSX : RSZ) + c * 2;
        s_raw[it] = isx ? RAWZ + (row * WX + c) * 4 : (row * WZ + c) * 4;
    }
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>(lds);
    // this wave's rows of a block: 2 w, 2 w + 1 of dZ and of X.  A lane past the row's allocated width (ld) re-reads the row's first columns
    // (its values are masked by `keep`); rows past M re-read the last row (masked when converted)
    const int vz = (4 * lane + 4 <= a.ldz) ? lane * 16 : 0, vx = (4 * lane + 4 <= a.ldx) ? lane * 16 : 0;
    const int vx8 = (lane < 8 && 256 + 4 * lane + 4 <= a.ldx) ? (256 + 4 * lane) * 4 : 0;
    auto dma_block = [&](long long blk, unsigned raw) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            long long m = blk * RB + row;
            m = m < M ? m : (M > 0 ? M - 1 : 0);
            int q = 0;
#pragma unroll
            for (int sv = 1; sv < 8; ++sv) q += (sv < a.n_seg && m >= (long long)sv * n_unit) ? 1 : 0;
            const long long pm = m + (long long)q * seg_extra;
            h2_dma_row(a.dZ + (size_t)pm * a.ldz, vz, raw + (unsigned)(row * WZ * 4));
            h2_dma_row(a.X + (size_t)pm * a.ldx, vx, raw + (unsigned)(RAWZ + row * WX * 4));
            h2_dma_row8(a.X + (size_t)pm * a.ldx, vx8, raw + (unsigned)(RAWZ + row * WX * 4 + 1024));
        }
    };
    auto convert_one = [&](const float4 v, char *base, long long blk, int it) __attribute__((always_inline)) {
        const unsigned keep = (blk * RB + s_row[it] < M) ? s_keep[it] : (s_keep[it] & ~15u);
        const float sc = (keep & 16u) ? sx : sz;
        unsigned ph0, pm0, ph1, pm1;
        split2h((keep & 1u) ? __fmul_rn(v.x, sc) : 0.f, (keep & 2u) ? __fmul_rn(v.y, sc) : 0.f, ph0, pm0);
        split2h((keep & 4u) ? __fmul_rn(v.z, sc) : 0.f, (keep & 8u) ? __fmul_rn(v.w, sc) : 0.f, ph1, pm1);
        ph0 = (keep & 0x100u) ? ((ph0 & 0xffff0000u) | 0x3c00u) : (keep & 0x200u) ? ((ph0 & 0x0000ffffu) | 0x3c000000u) : ph0;
        ph1 = (keep & 0x400u) ? ((ph1 & 0xffff0000u) | 0x3c00u) : (keep & 0x800u) ? ((ph1 & 0x0000ffffu) | 0x3c000000u) : ph1;
        *reinterpret_cast<uint2 *>(base + s_lds[it]) = make_uint2(ph0, ph1);
        *reinterpret_cast<uint2 *>(base + s_lds[it] + ((keep & 16u) ? PX : PZ)) = make_uint2(pm0, pm1);
    };
    auto convert = [&](const char *raw, char *base, long long blk) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) convert_one(*reinterpret_cast<const float4 *>(raw + s_raw[it]), base, blk, it);
    };
    const long long step = gridDim.x;
    long long blk = blockIdx.x;
    auto planes = [&](int p_) { return lds + p_ * STAGE; };
    auto raw_p = [&](int p_) { return lds + 2 * STAGE + p_ * RAW; };
    auto raw_l = [&](int p_) { return lds_base + (unsigned)(2 * STAGE + p_ * RAW); };
    dma_block(blk, raw_l(0));
    dma_block(blk + step, raw_l(1));
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    convert(raw_p(0), planes(0), blk);
    // One barrier per block: behind it block i's planes are complete, block i + 1's raw rows have landed, and nobody still reads what this
    // iteration overwrites (raw(par): converted in iteration i - 1; planes(par ^ 1): multiplied in iteration i - 1).  Block i's MFMAs and block
    // i + 1's conversion run in ONE phase, a conversion slot behind every pair of column tiles.  (Measured the same as two phases with a barrier
    // between them, and as loads three blocks ahead through a ring of three raw blocks: profiles/r05_wgrad_ablation.txt -- the loop's terms add up.)
    int par = 0;
    for (; blk < n_blocks; blk += step, par ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's rows of block i + 1 have landed
        __syncthreads();
#ifdef HNR_WG_DBG
        if (!(a.dbg & 4))
#endif
        dma_block(blk + 2 * step, raw_l(par));
        const char *zb = planes(par), *xb = planes(par) + 2 * PZ;
        const char *rawn = raw_p(par ^ 1);
        char *basen = planes(par ^ 1);
        const long long blkn = blk + step;
        const f16x8 zh = h2_tr_frag(zb, RSZ, 32 * wn, 0, lane), zm = h2_tr_frag(zb + PZ, RSZ, 32 * wn, 0, lane);
        f16x8 xh[2][2], xm[2][2];
        xh[0][0] = h2_tr_frag(xb, RSX, 0, 0, lane); xm[0][0] = h2_tr_frag(xb + PX, RSX, 0, 0, lane);
        xh[0][1] = h2_tr_frag(xb, RSX, 32, 0, lane); xm[0][1] = h2_tr_frag(xb + PX, RSX, 32, 0, lane);
#pragma unroll
        for (int u0 = 0; u0 < KTW; u0 += 2) {
            const int cb = (u0 >> 1) & 1, it = u0 >> 1;
            if (u0 + 2 < KTW) {                                                 // the next pair's fragments: in flight under this pair's MFMAs
                xh[cb ^ 1][0] = h2_tr_frag(xb, RSX, 32 * (u0 + 2), 0, lane); xm[cb ^ 1][0] = h2_tr_frag(xb + PX, RSX, 32 * (u0 + 2), 0, lane);
                if (u0 + 3 < KTW) { xh[cb ^ 1][1] = h2_tr_frag(xb, RSX, 32 * (u0 + 3), 0, lane); xm[cb ^ 1][1] = h2_tr_frag(xb + PX, RSX, 32 * (u0 + 3), 0, lane); }
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < NLD) v = *reinterpret_cast<const float4 *>(rawn + s_raw[it]);
#ifdef HNR_WG_DBG
            if (!(a.dbg & 1))
#endif
            if (u0 + 1 < KTW) {
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][0], acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][1], acc[u0 + 1], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][0], acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][1], acc[u0 + 1], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][0], acc[u0], 0, 0, 0);
                acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][1], acc[u0 + 1], 0, 0, 0);
            } else {
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][0], acc[u0], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][0], acc[u0], 0, 0, 0);
                acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][0], acc[u0], 0, 0, 0);
            }
#ifdef HNR_WG_DBG
            if (!(a.dbg & 2))
#endif
            if (it < NLD) convert_one(v, basen, blkn, it);
        }
        static_assert(NLD <= (KTW + 1) / 2, "a conversion slot behind every pair of column tiles");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // no DMA into this workgroup's LDS may outlive it
    constexpr int NP = 32 * NT, KP = 32 * KT, LDP = KP;
    float *out = a.partial + (size_t)blockIdx.x * NP * LDP;
    const float dsz = pow2f(-kz), dsx = pow2f(-kx);
#pragma unroll
    for (int u = 0; u < KTW; ++u) {
        const int kc = 32 * u + (lane & 31);
        const float d2 = kc == a.K ? 1.0f : dsx;                               // the bias column was staged unscaled
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(size_t)n * LDP + kc] = __fmul_rn(__fmul_rn(acc[u][r], dsz), d2);
        }
    }
}

// The same kernel for its shape in the training step -- N = 256 output columns, K = 256 .. 287 inputs, plain rows (n_seg = 1): the three 256-wide layers
// of the per-neighbour chain, 306 k rows each.  h2wgrad_dma_kernel spends ~370 instructions per 16-row block and wave for 27 MFMAs: segment arithmetic in
// 64-bit compares for every DMA'd row, a mask select per converted value, exec-masked fix-ups for the bias column in every conversion slot -- 14
// instructions per MFMA where the matrix pipe hides ~6.  Here
//   * the 512 real columns of a block (256 of dZ + 256 of X) are exactly four conversion slots per thread, a thread keeping its operand, its four
//     columns and its scale: LDS read, four products, two splits, one paired LDS write -- no masks (rows past the end: the dZ scale is 0);
//   * the ninth X tile (columns 256 .. 287: the inputs past the 256th, then the bias column of ones, then zeros) is constant for K = 256 -- written once
//     into both plane stages -- and one extra slot of the first 128 threads otherwise (NINTH);
//   * a DMA'd row costs scalar arithmetic only.
// Same staging values, same MFMA order per accumulator, same blocks per workgroup: partials bit-identical to h2wgrad_dma_kernel's.
template <bool NINTH>
__global__ __launch_bounds__(512, 1) void h2wgrad_dma256_kernel(H2WgradArgs a)
{
    constexpr int NT = 8, KT = 9, KTW = KT;
    constexpr int WZ = 32 * NT, WX = 32 * KT;
    constexpr int RSZ = ((WZ * 2 - 64 + 255) & ~255) + 64, RSX = ((WX * 2 - 64 + 255) & ~255) + 64;
    constexpr int RB = 16;
    constexpr int PZ = RB * RSZ, PX = RB * RSX, STAGE = 2 * PZ + 2 * PX;
    constexpr int RAWZ = RB * WZ * 4, RAWX = RB * WX * 4, RAW = RAWZ + RAWX;   // raw fp32 rows of a block: [16][256] dZ, then [16][288] X
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave;
    long long M = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    const long long n_blocks = (M + RB - 1) / RB;
    if (n_blocks == 0 && blockIdx.x > 0) return;
    const int kz = row_scale_exp(__uint_as_float(*a.zmax)), kx = row_scale_exp(__uint_as_float(*a.xmax));
    const float sz = pow2f(kz), sx = pow2f(kx);
    f32x16 acc[KTW];
#pragma unroll
    for (int u = 0; u < KTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    // main conversion slots of this thread: operand (tid & 64 ? X : dZ), columns 4 (tid & 63) .. + 3, rows (tid >> 7) + 4 it, it = 0..3
    const bool isx = (tid & 64) != 0;
    const int c0 = 4 * (tid & 63), r0 = tid >> 7;
    const int raw0 = isx ? RAWZ + (r0 * WX + c0) * 4 : (r0 * WZ + c0) * 4, raw_step = 4 * (isx ? WX : WZ) * 4;
    const int pl0 = (isx ? 2 * PZ : 0) + r0 * (isx ? RSX : RSZ) + c0 * 2, pl_step = 4 * (isx ? RSX : RSZ), pl_m = isx ? PX : PZ;
    const float sc_main = isx ? sx : sz;
    // the ninth X tile's slot (threads 0 .. 127): row tid >> 3, columns 256 + 4 (tid & 7) ..
    const int r9 = tid >> 3, c9 = 256 + 4 * (tid & 7);
    unsigned keep9 = 0;
    if (NINTH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) keep9 |= (c9 + 4 <= a.ldx && c9 + e < a.K) ? (1u << e) : 0u;
        if ((a.K >> 2) == (c9 >> 2)) keep9 |= 256u << (a.K & 3);
    }
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>(lds);
    const int vx8 = (lane < 8 && 256 + 4 * lane + 4 <= a.ldx) ? (256 + 4 * lane) * 4 : 0;
    // (32-bit block / row numbers: M < 2^31 rows; a DMA'd row costs a scalar min, one 64-bit product and two adds per operand)
    const int M32 = (int)M, last_row = M32 > 0 ? M32 - 1 : 0;
    auto dma_block = [&](int blk, unsigned raw) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            int m = blk * RB + row;                                             // (uniform: scalar arithmetic)
            m = m < last_row ? m : last_row;                                    // rows past the end re-read the last row (their dZ is scaled by 0)
            h2_dma_row(a.dZ + (size_t)(unsigned)m * (unsigned)a.ldz, lane * 16, raw + (unsigned)(row * WZ * 4));
            h2_dma_row(a.X + (size_t)(unsigned)m * (unsigned)a.ldx, lane * 16, raw + (unsigned)(RAWZ + row * WX * 4));
            if (NINTH) h2_dma_row8(a.X + (size_t)(unsigned)m * (unsigned)a.ldx, vx8, raw + (unsigned)(RAWZ + row * WX * 4 + 1024));
        }
    };
    auto convert_main = [&](const float4 v, char *base, int blk, int it) __attribute__((always_inline)) {
        const float sc = (isx || blk * RB + r0 + 4 * it < M32) ? sc_main : 0.f;
        unsigned ph0, pm0, ph1, pm1;
        split2h(__fmul_rn(v.x, sc), __fmul_rn(v.y, sc), ph0, pm0);
        split2h(__fmul_rn(v.z, sc), __fmul_rn(v.w, sc), ph1, pm1);
        char *dst = base + pl0 + it * pl_step;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
        *reinterpret_cast<uint2 *>(dst + pl_m) = make_uint2(pm0, pm1);
    };
    auto convert_ninth = [&](const float4 v, char *base) __attribute__((always_inline)) {      // (as h2wgrad_dma_kernel's slot; X rows past the end meet dZ = 0)
        const unsigned keep = keep9;
        unsigned ph0, pm0, ph1, pm1;
        split2h((keep & 1u) ? __fmul_rn(v.x, sx) : 0.f, (keep & 2u) ? __fmul_rn(v.y, sx) : 0.f, ph0, pm0);
        split2h((keep & 4u) ? __fmul_rn(v.z, sx) : 0.f, (keep & 8u) ? __fmul_rn(v.w, sx) : 0.f, ph1, pm1);
        ph0 = (keep & 0x100u) ? ((ph0 & 0xffff0000u) | 0x3c00u) : (keep & 0x200u) ? ((ph0 & 0x0000ffffu) | 0x3c000000u) : ph0;
        ph1 = (keep & 0x400u) ? ((ph1 & 0xffff0000u) | 0x3c00u) : (keep & 0x800u) ? ((ph1 & 0x0000ffffu) | 0x3c000000u) : ph1;
        char *dst = base + 2 * PZ + r9 * RSX + c9 * 2;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
        *reinterpret_cast<uint2 *>(dst + PX) = make_uint2(pm0, pm1);
    };
    const int step = gridDim.x, n_blk = (int)n_blocks;
    int blk = blockIdx.x;
    auto planes = [&](int p_) { return lds + p_ * STAGE; };
    auto raw_p = [&](int p_) { return lds + 2 * STAGE + p_ * RAW; };
    auto raw_l = [&](int p_) { return lds_base + (unsigned)(2 * STAGE + p_ * RAW); };
    if (!NINTH && tid < 128) {
        // K = 256: the ninth tile of every block = [1 (the bias column: fp16 1.0 in the high plane), 0, 0, ...]
#pragma unroll
        for (int p_ = 0; p_ < 2; ++p_) {
            char *dst = planes(p_) + 2 * PZ + r9 * RSX + c9 * 2;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(c9 == 256 ? 0x3c00u : 0u, 0u);
            *reinterpret_cast<uint2 *>(dst + PX) = make_uint2(0u, 0u);
        }
    }
    dma_block(blk, raw_l(0));
    dma_block(blk + step, raw_l(1));
    if (NINTH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) convert_main(*reinterpret_cast<const float4 *>(raw_p(0) + raw0 + it * raw_step), planes(0), blk, it);
    if (NINTH && tid < 128) convert_ninth(*reinterpret_cast<const float4 *>(raw_p(0) + RAWZ + (r9 * WX + c9) * 4), planes(0));
    // The loop body is FLAT: 27 MFMAs with everything else dealt out between them, a few instructions behind each (fragment reads of the next pair of X
    // tiles, a piece of the next block's conversion, the DMA of the block after), fenced so that the compiler keeps the order.  Two waves share a SIMD's
    // matrix pipe: six MFMAs back to back in both meant twelve serialised, and a wave waiting to issue an MFMA holds the SIMD's VALU port -- nothing else
    // of either wave ran meanwhile (the kernel's terms added up: profiles/r05_wgrad_ablation.txt); with an MFMA every ~8 instructions the other wave's
    // VALU work runs in the gaps.
    int par = 0;
    for (; blk < n_blk; blk += step, par ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's rows of block i + 1 have landed
        __syncthreads();
        const char *zb = planes(par), *xb = planes(par) + 2 * PZ;
        const char *rawn = raw_p(par ^ 1);
        char *basen = planes(par ^ 1);
        const int blkn = blk + step;
        const f16x8 zh = h2_tr_frag(zb, RSZ, 32 * wn, 0, lane), zm = h2_tr_frag(zb + PZ, RSZ, 32 * wn, 0, lane);
        f16x8 xh[2][2], xm[2][2];
        xh[0][0] = h2_tr_frag(xb, RSX, 0, 0, lane); xh[0][1] = h2_tr_frag(xb, RSX, 32, 0, lane);
        xm[0][0] = h2_tr_frag(xb + PX, RSX, 0, 0, lane); xm[0][1] = h2_tr_frag(xb + PX, RSX, 32, 0, lane);
        dma_block(blk + 2 * step, raw_l(par));
        __builtin_amdgcn_sched_barrier(0);
        float4 v = *reinterpret_cast<const float4 *>(rawn + raw0);
        float sc = 0.f;
        unsigned ph0 = 0, pm0 = 0, ph1 = 0, pm1 = 0;
#pragma unroll
        for (int u0 = 0; u0 < KTW; u0 += 2) {
            const int cb = (u0 >> 1) & 1, it = u0 >> 1;
            const bool pair = u0 + 1 < KTW;
            // conversion slot `it` of the next block in six pieces (it = 4: the ninth tile's slot / nothing)
            auto piece = [&](int pc) __attribute__((always_inline)) {
                if (it < 4) {
                    if (pc == 0) { sc = (isx || blkn * RB + r0 + 4 * it < M32) ? sc_main : 0.f; v.x = __fmul_rn(v.x, sc); v.y = __fmul_rn(v.y, sc); }
                    else if (pc == 1) { v.z = __fmul_rn(v.z, sc); v.w = __fmul_rn(v.w, sc); }
                    else if (pc == 2) split2h(v.x, v.y, ph0, pm0);
                    else if (pc == 3) split2h(v.z, v.w, ph1, pm1);
                    else if (pc == 4) {
                        char *dst = basen + pl0 + it * pl_step;
                        *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
                        *reinterpret_cast<uint2 *>(dst + pl_m) = make_uint2(pm0, pm1);
                    } else if (it + 1 < 4) v = *reinterpret_cast<const float4 *>(rawn + raw0 + (it + 1) * raw_step);
                    else if (NINTH) v = *reinterpret_cast<const float4 *>(rawn + RAWZ + ((r9 & 15) * WX + c9) * 4);
                } else if (NINTH && pc == 0 && tid < 128) convert_ninth(v, basen);
            };
            // the next pair's fragments (two transposing reads each), one behind each of the first four MFMAs
            auto frag = [&](int pc) __attribute__((always_inline)) {
                if (u0 + 2 >= KTW) return;
                if (pc == 0) xh[cb ^ 1][0] = h2_tr_frag(xb, RSX, 32 * (u0 + 2), 0, lane);
                else if (pc == 1) xm[cb ^ 1][0] = h2_tr_frag(xb + PX, RSX, 32 * (u0 + 2), 0, lane);
                else if (u0 + 3 < KTW && pc == 2) xh[cb ^ 1][1] = h2_tr_frag(xb, RSX, 32 * (u0 + 3), 0, lane);
                else if (u0 + 3 < KTW && pc == 3) xm[cb ^ 1][1] = h2_tr_frag(xb + PX, RSX, 32 * (u0 + 3), 0, lane);
            };
#define HNR_WG_SLOT(pc_, stmt_) do { stmt_; frag(pc_); piece(pc_); __builtin_amdgcn_sched_barrier(0); } while (0)
            if (pair) {
                HNR_WG_SLOT(0, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][0], acc[u0], 0, 0, 0));
                HNR_WG_SLOT(1, acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][1], acc[u0 + 1], 0, 0, 0));
                HNR_WG_SLOT(2, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][0], acc[u0], 0, 0, 0));
                HNR_WG_SLOT(3, acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][1], acc[u0 + 1], 0, 0, 0));
                HNR_WG_SLOT(4, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][0], acc[u0], 0, 0, 0));
                HNR_WG_SLOT(5, acc[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][1], acc[u0 + 1], 0, 0, 0));
            } else {
                HNR_WG_SLOT(0, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zm, xh[cb][0], acc[u0], 0, 0, 0));
                HNR_WG_SLOT(1, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xm[cb][0], acc[u0], 0, 0, 0));
                HNR_WG_SLOT(2, acc[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh, xh[cb][0], acc[u0], 0, 0, 0));
            }
#undef HNR_WG_SLOT
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // no DMA into this workgroup's LDS may outlive it
    constexpr int NP = 32 * NT, KP = 32 * KT, LDP = KP;
    float *out = a.partial + (size_t)blockIdx.x * NP * LDP;
    const float dsz = pow2f(-kz), dsx = pow2f(-kx);
#pragma unroll
    for (int u = 0; u < KTW; ++u) {
        const int kc = 32 * u + (lane & 31);
        const float d2 = kc == a.K ? 1.0f : dsx;                               // the bias column was staged unscaled
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(size_t)n * LDP + kc] = __fmul_rn(__fmul_rn(acc[u][r], dsz), d2);
        }
    }
}

// dW[n, k] (+)= sum over the workgroups that had rows, in a FIXED order; db[n] likewise (column KP of the partials).  64 outputs per block
// (coalesced along k), the partials dealt to the block's four waves (wave w takes partials w, w + 4, ...: eight interleaved running sums each),
// the four results added in wave order -- a single chain of `used` dependent loads per output was 13 us of latency per launch, fifteen times
// per training step.
__global__ __launch_bounds__(256) void h2wgrad_reduce_kernel(const float *__restrict__ partial, int n_wg, const long long *__restrict__ d_m, long long M_cap,
                                                             int NP, int LDP, int N, int K, float *__restrict__ dW, int lddw, float *__restrict__ db, int accumulate,
                                                             int n_seg, long long seg_stride, int rows_per_block = 16)
{
    __shared__ float s_p[4][64];
    long long M = M_cap;
    if (d_m) { const long long c = *d_m; if (c < M) M = c; }
    if (n_seg > 1) M = (M < seg_stride ? M : seg_stride) * n_seg;
    const long long n_blocks = (M + rows_per_block - 1) / rows_per_block;
    const int used = (int)(n_blocks < n_wg ? n_blocks : n_wg);
    const int lx = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lx;
    const int n = t / (K + 1), k = t - n * (K + 1);
    const bool live = n < N;
    const float *p = partial + (size_t)(live ? n : 0) * LDP + (live ? k : 0);      // (column K of the partials = the column sums of dZ)
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t gs = (size_t)NP * LDP;
    int g = wy;
    for (; g + 28 < used; g += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] += p[(size_t)(g + 4 * i) * gs];
    }
    for (int i = 0; g + 4 * i < used; ++i) s8[i] += p[(size_t)(g + 4 * i) * gs];
    s_p[wy][lx] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    __syncthreads();
    if (wy != 0 || !live) return;
    const float s = (s_p[0][lx] + s_p[1][lx]) + (s_p[2][lx] + s_p[3][lx]);
    const bool isb = k == K;
    if (isb) { if (db) db[n] = accumulate ? db[n] + s : s; }
    else dW[(size_t)n * lddw + k] = accumulate ? dW[(size_t)n * lddw + k] + s : s;
}

}  // namespace hnr

using namespace hnr;

static int h2_num_cus() { return device_num_cus(); }

extern "C" int64_t hnr_h2lin_packed_bytes(int K)
{
    if (K <= 0 || K > 288) return -1;
    return (int64_t)((K + 15) / 16) * CH_WSTEP + HL_META_FLOATS * 4;
}

// Batched: n_jobs <= 16 weight matrices packed by two launches.  W element (n, k) of job j = d_W[j][n * rs[j] + k * cs[j]].
extern "C" int hnr_h2lin_pack(int n_jobs, const float *const *d_W, const int64_t *rs, const int64_t *cs, const int *N, const int *K,
                              const float *const *d_bias, void *const *d_packed, void *stream)
{
    if (n_jobs <= 0 || n_jobs > H2_MAX_JOBS || !d_W || !rs || !cs || !N || !K || !d_packed) { set_error("hnr_h2lin_pack: bad argument (1..%d jobs)", H2_MAX_JOBS); return HNR_ERR_BADARG; }
    H2PackArgs a;
    hipStream_t st = (hipStream_t)stream;
    for (int j = 0; j < H2_MAX_JOBS; ++j) {
        if (j >= n_jobs) { a.job[j].W = nullptr; continue; }
        if (!d_W[j] || !d_packed[j] || N[j] <= 0 || N[j] > 256 || K[j] <= 0 || K[j] > 288 || ((uintptr_t)d_packed[j] & 15)) {
            set_error("hnr_h2lin_pack: job %d: N=%d (1..256) K=%d (1..288) or NULL / unaligned pointer", j, N[j], K[j]); return HNR_ERR_BADARG;
        }
        a.job[j].W = d_W[j]; a.job[j].rs = rs[j]; a.job[j].cs = cs[j]; a.job[j].N = N[j]; a.job[j].K = K[j];
        a.job[j].bias = d_bias ? d_bias[j] : nullptr; a.job[j].out = (char *)d_packed[j];
    }
    h2_meta_zero_kernel<<<1, 64, 0, st>>>(a);                                           // the max |W| words of all jobs (one launch instead of a memset per job)
    h2_wmax_kernel<<<dim3(16, n_jobs), 256, 0, st>>>(a);
    h2_pack_kernel<<<dim3(32, n_jobs), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

namespace hnr {
int h2lin_launch(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, const void *d_packed, int N, int K, int mode,
                 int act, float slope, const float *d_side, int ld_side, const uint32_t *d_side_bits, float *d_C, int ldc, uint32_t *d_absmax, void *stream);
int launch_h2lin_ws(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, float slope, const uint32_t *d_side_bits, float *d_C, int ldc,
                    uint32_t *d_absmax, void *stream);                          // csrc/h2lin_ws.hip
// dX = (dZ W) * LeakyReLU'(forward activation) with the activation's signs as the chain kernels' bit words (csrc/chain_ws.hip, training form)
int h2lin_dgrad_bits(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, int N, int K, float slope, const uint32_t *d_side_bits,
                     float *d_C, int ldc, uint32_t *d_absmax, void *stream)
{
    if (!d_side_bits || N != 256 || (M_cap & 31)) { set_error("h2lin_dgrad_bits: needs the bit words, N = 256 and whole 32-row tiles"); return HNR_ERR_BADARG; }
    // K = 256: the weight-stationary kernel (csrc/h2lin_ws.hip; bit-identical results).  HNR_H2LIN_WS=0: the streaming kernel below (A/B timing)
    static int use_ws = -1;
    if (use_ws < 0) { const char *e = getenv("HNR_H2LIN_WS"); use_ws = e ? atoi(e) : 1; }
    if (use_ws && K == 256 && d_dZ && d_packed && d_C && ldz >= 256 && !(ldz & 3) && ldc >= 256 && !(ldc & 3) && !((uintptr_t)d_dZ & 15) && !((uintptr_t)d_C & 15) &&
        !((uintptr_t)d_packed & 15) && slope > 0.f && slope < 1.f && (long long)M_cap * ldc * 4 < 0x7fffffffLL)
        return launch_h2lin_ws(d_dZ, ldz, M_cap, d_m, d_packed, slope, d_side_bits, d_C, ldc, d_absmax, stream);
    return h2lin_launch(d_dZ, ldz, M_cap, d_m, 1, 0, d_packed, N, K, 1, 0, slope, nullptr, 0, d_side_bits, d_C, ldc, d_absmax, stream);
}
}  // namespace hnr

extern "C" int hnr_h2lin_dgrad_bits(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, int N, int K, float slope,
                                    const uint32_t *d_side_bits, float *d_C, int ldc, uint32_t *d_absmax, void *stream)
{
    return hnr::h2lin_dgrad_bits(d_dZ, ldz, M_cap, d_m, d_packed, N, K, slope, d_side_bits, d_C, ldc, d_absmax, stream);
}

extern "C" int hnr_h2lin(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, const void *d_packed, int N, int K, int mode,
                         int act, float slope, const float *d_side, int ld_side, float *d_C, int ldc, uint32_t *d_absmax, void *stream)
{
    return hnr::h2lin_launch(d_A, lda, M_cap, d_m, n_seg, seg_stride, d_packed, N, K, mode, act, slope, d_side, ld_side, nullptr, d_C, ldc, d_absmax, stream);
}

int hnr::h2lin_launch(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, const void *d_packed, int N, int K, int mode,
                      int act, float slope, const float *d_side, int ld_side, const uint32_t *d_side_bits, float *d_C, int ldc, uint32_t *d_absmax, void *stream)
{
    if (n_seg < 1 || n_seg > 8 || (n_seg > 1 && seg_stride <= 0)) { set_error("hnr_h2lin: n_seg must be 1..8 (got %d) with a positive seg_stride", n_seg); return HNR_ERR_BADARG; }
    if (M_cap < 0 || N <= 0 || N > 256 || K <= 0 || K > 288 || lda < K || (lda & 3) || ldc < N || (ldc & 3) || (mode != 0 && mode != 1) ||
        (mode == 1 && !d_side_bits && (!d_side || ld_side < N || (ld_side & 3) || ((uintptr_t)d_side & 15))) || !(slope > 0.f && slope < 1.f)) {
        set_error("hnr_h2lin: bad sizes (N=%d K=%d lda=%d ldc=%d mode=%d ld_side=%d slope=%g)", N, K, lda, ldc, mode, ld_side, (double)slope);
        return HNR_ERR_BADARG;
    }
    if (M_cap == 0) return HNR_OK;
    if (!d_A || !d_packed || !d_C || ((uintptr_t)d_A & 15) || ((uintptr_t)d_C & 15) || ((uintptr_t)d_packed & 15)) { set_error("hnr_h2lin: NULL / unaligned pointer"); return HNR_ERR_BADARG; }
    H2LinArgs a;
    a.A = d_A; a.lda = lda; a.d_m = reinterpret_cast<const long long *>(d_m); a.M_cap = M_cap; a.n_seg = n_seg; a.seg_stride = seg_stride;
    a.wimg = (const char *)d_packed; a.N = N; a.K = K;
    a.mode = mode; a.act = act; a.slope = slope; a.side = d_side; a.lds_ = ld_side; a.side_bits = d_side_bits; a.C = d_C; a.ldc = ldc; a.absmax = d_absmax;
    static int spread = -1;
    if (spread < 0) { const char *e = getenv("HNR_H2LIN_SPREAD"); spread = e ? atoi(e) : 1; }
    a.spread = spread;
    const int S = (K + 15) / 16;
    const int64_t tiles = (M_cap * n_seg + 63) / 64;
    const int wgs = 2 * h2_num_cus(), grid = (int)(tiles < wgs ? tiles : wgs);
    hipStream_t st = (hipStream_t)stream;
#define HNR_H2LIN_CASE(S_)                                                                                                              \
    if (S == S_) {                                                                                                                      \
        constexpr int ldsb = S_ * (4096 + 32) + 64 * 4;                                                                                        \
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2lin_kernel<S_>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        h2lin_kernel<S_><<<grid, 256, ldsb, st>>>(a);                                                                                   \
        HNR_LAUNCH_CHECK();                                                                                                             \
        return HNR_OK;                                                                                                                  \
    }
    HNR_H2LIN_CASE(3) HNR_H2LIN_CASE(4) HNR_H2LIN_CASE(8) HNR_H2LIN_CASE(14) HNR_H2LIN_CASE(16)
#undef HNR_H2LIN_CASE
    set_error("hnr_h2lin: no kernel for K = %d (%d k steps); built: 3, 4, 8, 14, 16 k steps of 16", K, S);
    return HNR_ERR_BADARG;
}

extern "C" int hnr_absmax(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, int N, uint32_t *d_out, void *stream)
{
    if (M_cap < 0 || N <= 0 || lda < N || !d_out || n_seg < 1 || n_seg > 8 || (n_seg > 1 && seg_stride <= 0)) { set_error("hnr_absmax: bad argument"); return HNR_ERR_BADARG; }
    if (M_cap == 0) return HNR_OK;
    if (!d_A) { set_error("hnr_absmax: NULL pointer"); return HNR_ERR_BADARG; }
    const int64_t work = M_cap * n_seg * ((N + 3) / 4);
    const int64_t blocks = (work + 1023) / 1024;                                      // ~4 float4 per thread
    h2_absmax_kernel<<<(int)(blocks < 512 ? (blocks < 1 ? 1 : blocks) : 512), 256, 0, (hipStream_t)stream>>>(d_A, lda, reinterpret_cast<const long long *>(d_m), M_cap, N, d_out, n_seg, seg_stride);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// tile configurations of the weight-gradient kernel: NT = 32-column tiles of dZ (N <= 64 / 128 / 256), KT = tiles of the K + 1 staged columns of X
// (the bias column included), from the built list
static void h2wgrad_cfg(int N, int K, int *NT, int *KT)
{
    const int kt = K / 32 + 1;
    if (N <= 64) { *NT = 2; *KT = kt <= 2 ? 2 : (kt <= 3 ? 3 : (kt <= 5 ? 5 : 9)); }
    else if (N <= 128) { *NT = 4; *KT = kt <= 5 ? 5 : 9; }
    else { *NT = 8; *KT = kt <= 2 ? 2 : (kt <= 8 ? 8 : 9); }
}

extern "C" int64_t hnr_h2wgrad_scratch_bytes(int N, int K)
{
    if (N <= 0 || N > 256 || K <= 0 || K > 287) return -1;
    int NT, KT;
    h2wgrad_cfg(N, K, &NT, &KT);
    return (int64_t)h2_num_cus() * (32 * NT) * (32 * KT + 32) * 4;
}

extern "C" int hnr_h2wgrad(const float *d_dZ, int ldz, const float *d_X, int ldx, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, int N, int K,
                           const uint32_t *d_absmax_z, const uint32_t *d_absmax_x, float *d_dW, int lddw, float *d_db, int accumulate,
                           void *d_scratch, void *stream)
{
    if (n_seg < 1 || n_seg > 8 || (n_seg > 1 && seg_stride <= 0)) { set_error("hnr_h2wgrad: n_seg must be 1..8 (got %d) with a positive seg_stride", n_seg); return HNR_ERR_BADARG; }
    if (M_cap < 0 || N <= 0 || N > 256 || K <= 0 || K > 287 || ldz < N || (ldz & 3) || ldx < K || (ldx & 3) || lddw < K) {
        set_error("hnr_h2wgrad: bad sizes (N=%d K=%d ldz=%d ldx=%d lddw=%d)", N, K, ldz, ldx, lddw); return HNR_ERR_BADARG;
    }
    if (!d_dZ || !d_X || !d_absmax_z || !d_absmax_x || !d_dW || !d_scratch || ((uintptr_t)d_dZ & 15) || ((uintptr_t)d_X & 15)) { set_error("hnr_h2wgrad: NULL / unaligned pointer"); return HNR_ERR_BADARG; }
    int NT, KT;
    h2wgrad_cfg(N, K, &NT, &KT);
    H2WgradArgs a;
    a.dZ = d_dZ; a.ldz = ldz; a.X = d_X; a.ldx = ldx; a.d_m = reinterpret_cast<const long long *>(d_m); a.M_cap = M_cap; a.n_seg = n_seg; a.seg_stride = seg_stride;
    a.N = N; a.K = K;
    a.zmax = d_absmax_z; a.xmax = d_absmax_x; a.partial = (float *)d_scratch; a.dbg = 0;
#ifdef HNR_WG_DBG
    { const char *e = getenv("HNR_WG_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    const int64_t blocks = (M_cap * n_seg + 15) / 16;
    const int n_cu = h2_num_cus();
    int grid = (int)((blocks + 15) / 16 < n_cu ? (blocks + 15) / 16 : n_cu);           // at least 8 row blocks per workgroup: every workgroup writes (and the reduction reads) a whole partial
    if (grid < 1) grid = 1;
    hipStream_t st = (hipStream_t)stream;
#define HNR_H2WG_CASE(NT_, KT_)                                                                                                         \
    if (NT == NT_ && KT == KT_ && !biasv && !(NT_ == 8 && KT_ == 9)) {                                                                                                       \
        constexpr int rsz = ((32 * NT_ * 2 - 64 + 255) & ~255) + 64, rsx = ((32 * KT_ * 2 - 64 + 255) & ~255) + 64;                     \
        constexpr int ldsb = 3 * (2 * 16 * rsz + 2 * 16 * rsx);                                                                    \
        static PerDeviceOnce once_;                                                                                                     \
        if (once_.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_kernel<NT_, KT_, 8 / NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        h2wgrad_kernel<NT_, KT_, 8 / NT_><<<grid, 512, ldsb, st>>>(a);                                                                  \
    }
    const bool biasv = false;
    if (NT == 8 && KT == 9) {
        constexpr int rsz = ((32 * 8 * 2 - 64 + 255) & ~255) + 64, rsx = ((32 * 9 * 2 - 64 + 255) & ~255) + 64, ldsb = 3 * (2 * 16 * rsz + 2 * 16 * rsx);
        constexpr int lds_dma = 2 * (2 * 16 * rsz + 2 * 16 * rsx) + 2 * 16 * (256 + 288) * 4;     // two plane stages + two raw blocks
        static int use_dma = -1;
        if (use_dma < 0) { const char *e = getenv("HNR_WGRAD_DMA"); use_dma = e ? atoi(e) : 1; }   // 0: the register-staged kernel (A/B timing)
        static PerDeviceOnce once89;
        if (once89.first()) {
            HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_kernel<8, 9, 1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
            HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
            HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_dma256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
            HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_dma256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
        }
        // N = 256, plain rows: the kernel specialised for the training step's 256-wide layers (HNR_WGRAD_DMA=2: the general DMA kernel instead, A/B timing)
        const bool fast = use_dma == 1 && N == 256 && n_seg == 1 && ldz >= 256 && ldx >= 256 && M_cap < 0x7ffff000LL;
        if (fast && K == 256) h2wgrad_dma256_kernel<false><<<grid, 512, lds_dma, st>>>(a);
        else if (fast) h2wgrad_dma256_kernel<true><<<grid, 512, lds_dma, st>>>(a);
        else if (use_dma) h2wgrad_dma_kernel<<<grid, 512, lds_dma, st>>>(a);
        else h2wgrad_kernel<8, 9, 1, 0, 1><<<grid, 512, ldsb, st>>>(a);
    }
    HNR_H2WG_CASE(8, 9) HNR_H2WG_CASE(8, 8) HNR_H2WG_CASE(8, 2) HNR_H2WG_CASE(4, 9) HNR_H2WG_CASE(4, 5) HNR_H2WG_CASE(2, 9) HNR_H2WG_CASE(2, 5) HNR_H2WG_CASE(2, 3) HNR_H2WG_CASE(2, 2)
#undef HNR_H2WG_CASE
    HNR_LAUNCH_CHECK();
    const int NP = 32 * NT, LDP = 32 * KT + (biasv ? 32 : 0);
    const int total = N * (K + 1);
    h2wgrad_reduce_kernel<<<(total + 63) / 64, 256, 0, st>>>((const float *)d_scratch, grid, reinterpret_cast<const long long *>(d_m), M_cap, NP, LDP, N, K,
                                                              d_dW, lddw, d_db, accumulate, n_seg, seg_stride);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
