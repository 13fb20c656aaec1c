// fp32 dense layer on the matrix cores:  C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]).
//
// Replaces the nn.Linear (+LeakyReLU) layers inside PointAggregator.viewmlp
// (models/aggregators/point_aggregators.py:948 block1, :972 block3, :1037 color_feature_branch,
// :1199 aux_merge_weight_block, :1292 color_mixup_block), which the reference runs as eager
// cuBLAS GEMMs on boolean-mask-gathered rows.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain
// (no TF32/xf32 exists on gfx950), so results differ from the CPU oracle only by summation order.
//
// Tiling (wave64): 256-thread workgroup = 2x2 waves; each wave owns TM x TN MFMA tiles of 32x32.
//   <2,2>: 128x128 block tile (N >= 128)      <2,1>: 128x64 block tile (N <= 64)
// BK = 32.  A and W tiles are staged through LDS in row-major [row][BK+1] (the +1 pad makes the
// MFMA fragment reads -- 32 lanes x 32 rows, same k -- conflict-free); the next tile's global loads are
// issued into registers before the current tile's MFMAs (register double buffering).
// Weights are pre-packed once per checkpoint into a zero-padded [N_pad][K_pad] image, so the W loads
// are unguarded 16-B loads; A rows are guarded (zero-filled) for m >= M and k >= K.
#include "hnr_common.h"

namespace hnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 1;

template <int TM, int TN, int ACT>   // ACT: 0 none, 1 LeakyReLU(slope)
__global__ __launch_bounds__(256) void linear_f32_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Wp, int K_pad,
                                                         const float *__restrict__ bias_p, float *__restrict__ C, int ldc,
                                                         int M, int N, int K, float slope)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_F4 = BM * BK / 4 / 256;     // float4 per thread for the A tile
    constexpr int W_F4 = BN * BK / 4 / 256;
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_F4], rw[W_F4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, c4 = (idx & 7) << 2;
            const int gm = m0 + row, gk = k0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M && gk < K) {
                v = *reinterpret_cast<const float4 *>(A + (size_t)gm * lda + gk);
                if (gk + 3 >= K) {
                    if (gk + 1 >= K) v.y = 0.f;
                    if (gk + 2 >= K) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, c4 = (idx & 7) << 2;
            rw[i] = *reinterpret_cast<const float4 *>(Wp + (size_t)(n0 + row) * K_pad + k0 + c4);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256;
            float *d = As + (idx >> 3) * LDS_LD + ((idx & 7) << 2);
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const int idx = tid + i * 256;
            float *d = Ws + (idx >> 3) * LDS_LD + ((idx & 7) << 2);
            d[0] = rw[i].x; d[1] = rw[i].y; d[2] = rw[i].z; d[3] = rw[i].w;
        }
    };

    const int nk = K_pad / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    const float *a_base = As + (wr * 32 * TM + (lane & 31)) * LDS_LD + (lane >> 5);
    const float *w_base = Ws + (wc * 32 * TN + (lane & 31)) * LDS_LD + (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_base[i * 32 * LDS_LD + kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = w_base[j * 32 * LDS_LD + kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = n0 + wc * 32 * TN + j * 32 + (lane & 31);
        const float bv = bias_p[gn];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wr * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bv;
                if (ACT == 1) v = v > 0.f ? v : v * slope;
                if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v;
            }
        }
    }
}

__global__ void pack_linear_kernel(const float *__restrict__ W, const float *__restrict__ bias, int N, int K,
                                   int N_pad, int K_pad, float *__restrict__ Wp, float *__restrict__ bias_p)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)N_pad * K_pad) {
        const int n = (int)(i / K_pad), k = (int)(i % K_pad);
        Wp[i] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
    if (i < N_pad) bias_p[i] = (i < N && bias) ? bias[i] : 0.f;
}

}  // namespace hnr

using namespace hnr;

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int hnr_linear_packed_dims(int N, int K, int *N_pad, int *K_pad)
{
    if (N <= 0 || K <= 0 || !N_pad || !K_pad) return HNR_ERR_BADARG;
    *N_pad = round_up(N, N >= 128 ? 128 : 64);
    *K_pad = round_up(K, BK);
    return HNR_OK;
}

extern "C" int hnr_linear_pack(const float *d_W, const float *d_bias, int N, int K, float *d_Wp, float *d_bias_p, void *stream)
{
    int Np, Kp;
    if (!d_W || !d_Wp || !d_bias_p || hnr_linear_packed_dims(N, K, &Np, &Kp) != HNR_OK) {
        set_error("hnr_linear_pack: bad argument"); return HNR_ERR_BADARG;
    }
    const int64_t n = (int64_t)Np * Kp;
    pack_linear_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_W, d_bias, N, K, Np, Kp, d_Wp, d_bias_p);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_linear_f32(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, float *d_C, int ldc,
                              int M, int N, int K, int act, float slope, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || lda < K || (lda & 3) || ldc < N || (act != 0 && act != 1)) {
        set_error("hnr_linear_f32: bad sizes (M=%d N=%d K=%d lda=%d ldc=%d act=%d; lda must be a multiple of 4 and >= K)", M, N, K, lda, ldc, act);
        return HNR_ERR_BADARG;
    }
    if (M == 0) return HNR_OK;
    if (!d_A || !d_Wp || !d_bias_p || !d_C || ((uintptr_t)d_A & 15)) { set_error("hnr_linear_f32: NULL or unaligned pointer"); return HNR_ERR_BADARG; }
    int Np, Kp;
    hnr_linear_packed_dims(N, K, &Np, &Kp);
    hipStream_t st = (hipStream_t)stream;
    if (N >= 128) {
        dim3 grid(cdiv(M, 128), Np / 128);
        if (act) linear_f32_kernel<2, 2, 1><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else linear_f32_kernel<2, 2, 0><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
    } else {
        dim3 grid(cdiv(M, 128), Np / 64);
        if (act) linear_f32_kernel<2, 1, 1><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else linear_f32_kernel<2, 1, 0><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
    }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
