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
    float *C; int ldc;
    unsigned *absmax;                          // optional: max |C| (bit pattern, atomicMax)
};

// 64-row tiles, two workgroups per CU (one's row loads / split / epilogue under the other's MFMAs); wave w owns output columns 64 w .. + 63.
template <int S>
__global__ __launch_bounds__(256, 2) void h2lin_kernel(H2LinArgs a)
{
    constexpr int RT = 2, SLOT = RT * 2048, ROWS = 32 * RT;
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
    const int col0 = 64 * wave + 16 * h;
    const unsigned woff = (unsigned)(2 * wave) * 2048u + (unsigned)lane * 16u;
    const bool active = 64 * wave < a.N;
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
        if (active) {
            float inv[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) inv[rt] = __fmul_rn(rowinv[32 * rt + j], dw);
            f32x16 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][c][r] = 0.f;
            h2_mfma_layer<RT, 2, S, 0, CH_WSTEP, SLOT>(wsrd, 0, woff, lds, lane, acc, []() {});
            // mode 1: the rows' stored activations, one row tile ahead of its use (the weight fragments' registers are free by now)
            float4 sd[2][2][4];
            auto load_side = [&](int rt) {
                long long row = row_base + 32 * rt + j;
                if (row >= M) row = M - 1;
                const float *srow = a.side + (size_t)phys(row) * a.lds_ + col0;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sd[rt & 1][c][q] = (col0 + 32 * c + 4 * q + 4 <= a.lds_) ? *reinterpret_cast<const float4 *>(srow + 32 * c + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
            };
            if (a.mode == 1) load_side(0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const long long row = row_base + 32 * rt + j;
                if (a.mode == 1 && rt + 1 < RT) load_side(rt + 1);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float o[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float bq[4] = {0.f, 0.f, 0.f, 0.f};
                        if (a.mode == 0) { const float4 b4 = *reinterpret_cast<const float4 *>(meta + col0 + 32 * c + 4 * q); bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w; }
                        const float *sq = reinterpret_cast<const float *>(&sd[rt & 1][c][q]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaf(acc[rt][c][4 * q + e], inv[rt], bq[e]);
                            if (a.mode == 0) { if (a.act) v = fmaxf(v, __fmul_rn(v, a.slope)); }
                            else v = __fmul_rn(v, sq[e] > 0.f ? 1.f : a.slope);
                            o[4 * q + e] = v;
                            gmax = fmaxf(gmax, fabsf(v));
                        }
                    }
                    if (row < M) {
                        float *dst = a.C + (size_t)phys(row) * a.ldc + col0 + 32 * c;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col0 + 32 * c + 4 * q + 4 <= a.ldc && col0 + 32 * c + 4 * q < ((a.N + 3) & ~3))
                                *reinterpret_cast<float4 *>(dst + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                    }
                }
            }
        }
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

// NT = 32-column tiles of dZ (N), KT = tiles of X (K) covered by the workgroup; 8 waves = WN (along N) x WK (along K).  32-row blocks,
// double-buffered LDS planes: the next block's rows are in flight (registers) while the MFMAs run over the current one.
template <int NT, int KT, int WN, int WK>
__global__ __launch_bounds__(512, 1) void h2wgrad_kernel(H2WgradArgs a)
{
    static_assert(WN * WK == 8, "8 waves");
    constexpr int NTW = (NT + WN - 1) / WN, KTW = (KT + WK - 1) / WK;
    constexpr int WZ = 32 * NT, WX = 32 * KT;                                  // columns staged per row
    constexpr int RSZ = ((WZ * 2 - 64 + 255) & ~255) + 64, RSX = ((WX * 2 - 64 + 255) & ~255) + 64;      // row strides in bytes: 64 (mod 256), so the 4 rows x 32 B of a transposed read hit distinct banks
    constexpr int PZ = 32 * RSZ, PX = 32 * RSX, STAGE = 2 * PZ + 2 * PX;
    constexpr int Z4 = WZ / 4, X4 = WX / 4, NLD = (32 * (Z4 + X4) + 511) / 512;      // float4 loads per thread per block
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave % WN, wk = wave / WN;
    long long M = a.M_cap, n_unit = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    if (a.n_seg > 1) { n_unit = M < a.seg_stride ? M : a.seg_stride; M = n_unit * a.n_seg; }
    const long long n_blocks = (M + 31) / 32;
    const int kz = row_scale_exp(__uint_as_float(*a.zmax)), kx = row_scale_exp(__uint_as_float(*a.xmax));
    const float sz = pow2f(kz), sx = pow2f(kx);

    f32x16 acc[NTW][KTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int u = 0; u < KTW; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const int kt_used = a.K / 32 + 1;                                          // tiles up to the one that holds column K (the bias column)

    float4 stg[NLD];
    auto load_block = [&](long long blk) {
        int tid_t = tid;                                                       // laundered: the per-slot (row, column) arithmetic is cheap, hoisted it is 2 NLD live registers
        asm volatile("" : "+v"(tid_t));
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid_t + 512 * it;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < 32 * (Z4 + X4)) {
                const bool isx = idx >= 32 * Z4;
                const int id2 = isx ? idx - 32 * Z4 : idx, per = isx ? X4 : Z4;
                const int row = id2 / per, c = 4 * (id2 - row * per);
                const long long m = blk * 32 + row;
                const int lim = isx ? a.K : a.N, ld = isx ? a.ldx : a.ldz;
                if (m < M && c < lim && c + 4 <= ld) {                          // (columns past lim are staged as zeros: their outputs are never read)
                    long long pm = m;
                    if (a.n_seg > 1) { int q = 0; for (int sv = 1; sv < a.n_seg && sv < 8; ++sv) q += (m >= (long long)sv * n_unit) ? 1 : 0; pm = (long long)q * a.seg_stride + (m - (long long)q * n_unit); }
                    v = *reinterpret_cast<const float4 *>((isx ? a.X : a.dZ) + (size_t)pm * ld + c);
                    if (c + 1 >= lim) v.y = 0.f;
                    if (c + 2 >= lim) v.z = 0.f;
                    if (c + 3 >= lim) v.w = 0.f;
                }
            }
            stg[it] = v;
        }
    };
    auto store_block = [&](int buf) {
        char *base = lds + buf * STAGE;
        int tid_t = tid;
        asm volatile("" : "+v"(tid_t));
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = tid_t + 512 * it;
            if (idx < 32 * (Z4 + X4)) {
                const bool isx = idx >= 32 * Z4;
                const int id2 = isx ? idx - 32 * Z4 : idx, per = isx ? X4 : Z4;
                const int row = id2 / per, c = 4 * (id2 - row * per);
                const float sc = isx ? sx : sz;
                unsigned ph0, pm0, ph1, pm1;
                split2h(__fmul_rn(stg[it].x, sc), __fmul_rn(stg[it].y, sc), ph0, pm0);
                split2h(__fmul_rn(stg[it].z, sc), __fmul_rn(stg[it].w, sc), ph1, pm1);
                if (isx && (a.K >> 2) == (c >> 2)) {
                    // column K of the staged X rows = 1 (fp16 1.0 in the high plane, unscaled): dW[:, K] becomes the column sums of dZ = db
                    const int e = a.K & 3;
                    if (e < 2) { ph0 = e == 0 ? ((ph0 & 0xffff0000u) | 0x3c00u) : ((ph0 & 0x0000ffffu) | 0x3c000000u); pm0 = e == 0 ? (pm0 & 0xffff0000u) : (pm0 & 0x0000ffffu); }
                    else { ph1 = e == 2 ? ((ph1 & 0xffff0000u) | 0x3c00u) : ((ph1 & 0x0000ffffu) | 0x3c000000u); pm1 = e == 2 ? (pm1 & 0xffff0000u) : (pm1 & 0x0000ffffu); }
                }
                char *dst = base + (isx ? 2 * PZ : 0) + (size_t)row * (isx ? RSX : RSZ) + c * 2;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
                *reinterpret_cast<uint2 *>(dst + (isx ? PX : PZ)) = make_uint2(pm0, pm1);
            }
        }
    };

    long long blk = blockIdx.x;
    int buf = 0;
    if (blk < n_blocks) { load_block(blk); store_block(0); }
    __syncthreads();
    for (; blk < n_blocks; blk += gridDim.x, buf ^= 1) {
        const long long nxt = blk + gridDim.x;
        if (nxt < n_blocks) load_block(nxt);
        const char *zb = lds + buf * STAGE, *xb = zb + 2 * PZ;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 zf[NTW][2];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int nt = wn * NTW + t;
                if (nt < NT) { zf[t][0] = h2_tr_frag(zb, RSZ, 32 * nt, ks, lane); zf[t][1] = h2_tr_frag(zb + PZ, RSZ, 32 * nt, ks, lane); }
            }
#pragma unroll
            for (int u = 0; u < KTW; ++u) {
                const int kt = wk * KTW + u;
                if (kt < KT && kt < kt_used) {
                    const f16x8 xh = h2_tr_frag(xb, RSX, 32 * kt, ks, lane), xm = h2_tr_frag(xb + PX, RSX, 32 * kt, ks, lane);
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        if (wn * NTW + t < NT) {
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[t][1], xh, acc[t][u], 0, 0, 0);
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[t][0], xm, acc[t][u], 0, 0, 0);
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf[t][0], xh, acc[t][u], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (nxt < n_blocks) store_block(buf ^ 1);
        __syncthreads();
    }
    // ---- this workgroup's partial: [NP = 32 NT][KP], true units (scales removed); accumulator (reg r, lane): row n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    constexpr int NP = 32 * NT, KP = 32 * KT, LDP = KP;
    float *out = a.partial + (size_t)blockIdx.x * NP * LDP;
    const float dsz = pow2f(-kz), dsx = pow2f(-kx);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int nt = wn * NTW + t;
        if (nt >= NT) continue;
#pragma unroll
        for (int u = 0; u < KTW; ++u) {
            const int kt = wk * KTW + u;
            if (kt >= KT || kt >= kt_used) continue;
            const int kc = 32 * kt + (lane & 31);
            const float d2 = kc == a.K ? 1.0f : dsx;                           // the bias column was staged unscaled
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(size_t)n * LDP + kc] = __fmul_rn(__fmul_rn(acc[t][u][r], dsz), d2);
            }
        }
    }
}

// dW[n, k] (+)= sum over the workgroups that had rows, in index order; db[n] likewise (column KP of the partials)
__global__ __launch_bounds__(256) void h2wgrad_reduce_kernel(const float *__restrict__ partial, int n_wg, const long long *__restrict__ d_m, long long M_cap,
                                                             int NP, int LDP, int N, int K, float *__restrict__ dW, int lddw, float *__restrict__ db, int accumulate,
                                                             int n_seg, long long seg_stride)
{
    long long M = M_cap;
    if (d_m) { const long long c = *d_m; if (c < M) M = c; }
    if (n_seg > 1) M = (M < seg_stride ? M : seg_stride) * n_seg;
    const long long n_blocks = (M + 31) / 32;
    const int used = (int)(n_blocks < n_wg ? n_blocks : n_wg);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = t / (K + 1), k = t - n * (K + 1);
    if (n >= N) return;
    const bool isb = k == K;
    if (isb && !db) return;
    const float *p = partial + (size_t)n * LDP + k;                                 // (column K of the partials = the column sums of dZ)
    // fixed order: eight interleaved running sums (their loads are independent: a single dependent chain of `used` loads is latency-bound), then a fixed tree
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t gs = (size_t)NP * LDP;
    int g = 0;
    for (; g + 8 <= used; g += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] += p[(size_t)(g + i) * gs];
    }
    for (int i = 0; g + i < used; ++i) s8[i] += p[(size_t)(g + i) * gs];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    if (isb) db[n] = accumulate ? db[n] + s : s;
    else dW[(size_t)n * lddw + k] = accumulate ? dW[(size_t)n * lddw + k] + s : s;
}

}  // namespace hnr

using namespace hnr;

static int h2_num_cus()
{
    int dev = 0;
    static int n_cu[64] = {0};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        hipDeviceProp_t prop;
        n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}

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

extern "C" int hnr_h2lin(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, const void *d_packed, int N, int K, int mode,
                         int act, float slope, const float *d_side, int ld_side, float *d_C, int ldc, uint32_t *d_absmax, void *stream)
{
    if (n_seg < 1 || n_seg > 8 || (n_seg > 1 && seg_stride <= 0)) { set_error("hnr_h2lin: n_seg must be 1..8 (got %d) with a positive seg_stride", n_seg); return HNR_ERR_BADARG; }
    if (M_cap < 0 || N <= 0 || N > 256 || K <= 0 || K > 288 || lda < K || (lda & 3) || ldc < N || (ldc & 3) || (mode != 0 && mode != 1) ||
        (mode == 1 && (!d_side || ld_side < N || (ld_side & 3) || ((uintptr_t)d_side & 15))) || !(slope > 0.f && slope < 1.f)) {
        set_error("hnr_h2lin: bad sizes (N=%d K=%d lda=%d ldc=%d mode=%d ld_side=%d slope=%g)", N, K, lda, ldc, mode, ld_side, (double)slope);
        return HNR_ERR_BADARG;
    }
    if (M_cap == 0) return HNR_OK;
    if (!d_A || !d_packed || !d_C || ((uintptr_t)d_A & 15) || ((uintptr_t)d_C & 15) || ((uintptr_t)d_packed & 15)) { set_error("hnr_h2lin: NULL / unaligned pointer"); return HNR_ERR_BADARG; }
    H2LinArgs a;
    a.A = d_A; a.lda = lda; a.d_m = reinterpret_cast<const long long *>(d_m); a.M_cap = M_cap; a.n_seg = n_seg; a.seg_stride = seg_stride;
    a.wimg = (const char *)d_packed; a.N = N; a.K = K;
    a.mode = mode; a.act = act; a.slope = slope; a.side = d_side; a.lds_ = ld_side; a.C = d_C; a.ldc = ldc; a.absmax = d_absmax;
    const int S = (K + 15) / 16;
    const int64_t tiles = (M_cap * n_seg + 63) / 64;
    const int wgs = 2 * h2_num_cus(), grid = (int)(tiles < wgs ? tiles : wgs);
    hipStream_t st = (hipStream_t)stream;
#define HNR_H2LIN_CASE(S_)                                                                                                              \
    if (S == S_) {                                                                                                                      \
        constexpr int ldsb = S_ * 4096 + 64 * 4;                                                                                        \
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

// tile configurations of the weight-gradient kernel (K + 1 columns of X are staged: the bias column): (N <= 256, K <= 287), (N <= 128, K <= 287),
// (N <= 64, K <= 159)
static void h2wgrad_cfg(int N, int K, int *NT, int *KT)
{
    if (N <= 64 && K <= 159) { *NT = 2; *KT = 5; }
    else if (N <= 128) { *NT = 4; *KT = 9; }
    else { *NT = 8; *KT = 9; }
}

extern "C" int64_t hnr_h2wgrad_scratch_bytes(int N, int K)
{
    if (N <= 0 || N > 256 || K <= 0 || K > 287) return -1;
    int NT, KT;
    h2wgrad_cfg(N, K, &NT, &KT);
    return (int64_t)h2_num_cus() * (32 * NT) * (32 * KT) * 4;
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
    a.zmax = d_absmax_z; a.xmax = d_absmax_x; a.partial = (float *)d_scratch;
    const int64_t blocks = (M_cap * n_seg + 31) / 32;
    const int n_cu = h2_num_cus();
    int grid = (int)((blocks + 7) / 8 < n_cu ? (blocks + 7) / 8 : n_cu);           // at least 8 row blocks per workgroup: every workgroup writes (and the reduction reads) a whole partial
    if (grid < 1) grid = 1;
    hipStream_t st = (hipStream_t)stream;
#define HNR_H2WG_CASE(NT_, KT_, WN_, WK_)                                                                                               \
    if (NT == NT_ && KT == KT_) {                                                                                                       \
        constexpr int rsz = ((32 * NT_ * 2 - 64 + 255) & ~255) + 64, rsx = ((32 * KT_ * 2 - 64 + 255) & ~255) + 64;                               \
        constexpr int ldsb = 2 * (2 * 32 * rsz + 2 * 32 * rsx);                                                                         \
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2wgrad_kernel<NT_, KT_, WN_, WK_>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        h2wgrad_kernel<NT_, KT_, WN_, WK_><<<grid, 512, ldsb, st>>>(a);                                                                 \
    }
    HNR_H2WG_CASE(8, 9, 8, 1) HNR_H2WG_CASE(4, 9, 4, 2) HNR_H2WG_CASE(2, 5, 2, 4)
#undef HNR_H2WG_CASE
    HNR_LAUNCH_CHECK();
    const int NP = 32 * NT, LDP = 32 * KT;
    const int total = N * (K + 1);
    h2wgrad_reduce_kernel<<<(total + 255) / 256, 256, 0, st>>>((const float *)d_scratch, grid, reinterpret_cast<const long long *>(d_m), M_cap, NP, LDP, N, K,
                                                              d_dW, lddw, d_db, accumulate, n_seg, seg_stride);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
