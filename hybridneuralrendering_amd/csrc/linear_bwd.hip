// Weight gradient of a dense layer on the matrix cores:  dW[N,K] = dZ[M,N]^T * X[M,K],  db[N] = sum_m dZ[m,:].
//
// Backward of every nn.Linear of PointAggregator.viewmlp (models/aggregators/point_aggregators.py:948, :972, :1037,
// :1199, :1292), which the reference gets from torch autograd (addmm backward = one cuBLAS GEMM per layer).
//
// The contraction runs over the ROW index m, which is the slow index of both operands in memory, and that is exactly
// the operand layout of v_mfma_f32_32x32x2_f32: lane l supplies A[i = l % 32][k = l / 32] and B[k = l / 32][j = l % 32].
// With A[i][k] = dZ[m0 + k][n(i)] and B[k][j] = X[m0 + k][c(j)] lanes 0-31 read row m0 and lanes 32-63 row m0+1 -- plain
// coalesced global loads, no LDS, no transpose.  One lane loads 16 B (4 consecutive columns), and since the assignment
// of output columns to MFMA tiles is free, tile t of a wave takes the columns {4 i + t}: one dwordx4 load per operand
// feeds a 4 x 4 grid of MFMA tiles = a 128 x 128 block of dW per wave (256 accumulator registers; one wave per SIMD).
// Loads run U row pairs ahead of the MFMAs in registers.
//
// Grid = (row partitions, N/128, K/128), one wave each; partition p writes its 128 x 128 partial sums, a second kernel adds
// the partitions in a fixed order (deterministic, no atomics) and applies `accumulate`.
#include "hnr_common.h"

namespace hnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_U = 4;          // row pairs in flight

__global__ __launch_bounds__(64) void linear_wgrad_kernel(const float *__restrict__ dZ, int ldz, const float *__restrict__ X, int ldx,
                                                          int M, int rows_per_part, int Np, int Kp,
                                                          float *__restrict__ partial, float *__restrict__ bias_partial)
{
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int p = blockIdx.x, n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
    const int m_begin = p * rows_per_part;
    const int m_end = m_begin + rows_per_part < M ? m_begin + rows_per_part : M;
    // columns past the row stride are never loaded (ld is a multiple of 4); columns in [N, ld) may hold anything: a column only
    // ever contributes to its own output column, and those are not read back
    const bool a_ok = n0 + 4 * c < ldz, b_ok = k0 + 4 * c < ldx;
    const float *pa = dZ + n0 + 4 * c, *pb = X + k0 + 4 * c;

    f32x16 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](float4 (&a)[WG_U], float4 (&b)[WG_U], int m) {
#pragma unroll
        for (int q = 0; q < WG_U; ++q) {
            const int row = m + 2 * q + h;
            const bool ok = row < m_end;
            a[q] = (ok && a_ok) ? *reinterpret_cast<const float4 *>(pa + (size_t)row * ldz) : zero4;
            b[q] = (ok && b_ok) ? *reinterpret_cast<const float4 *>(pb + (size_t)row * ldx) : zero4;
        }
    };
    float4 a[WG_U], b[WG_U];
    load(a, b, m_begin);
#pragma unroll 1
    for (int m = m_begin; m < m_end; m += 2 * WG_U) {
        float4 an[WG_U], bn[WG_U];
        load(an, bn, m + 2 * WG_U);                 // rows >= m_end come back as zeros
#pragma unroll
        for (int q = 0; q < WG_U; ++q) {
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
            const float bv[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bs[t] += av[t];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < WG_U; ++q) { a[q] = an[q]; b[q] = bn[q]; }
    }
    // D layout of the 32x32 MFMA: column j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    // output element (n, k) = (n0 + 4 i + t, k0 + 4 j + u): the 4 u values of a lane are one 16-B store
    float *out = partial + (size_t)p * Np * Kp;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int n = n0 + 4 * i + t;
            *reinterpret_cast<float4 *>(out + (size_t)n * Kp + k0 + 4 * c) = make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
        }
    if (blockIdx.z == 0 && bias_partial) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bs[t] += __shfl_xor(bs[t], 32);
        if (h == 0) *reinterpret_cast<float4 *>(bias_partial + (size_t)p * Np + n0 + 4 * c) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    }
}

// 8 lanes per output element, each adding every 8th partition, then an xor-shuffle tree: fixed order, deterministic
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bias_partial,
                                                                  int P, int Np, int Kp, int N, int K, float *__restrict__ dW, int lddw,
                                                                  float *__restrict__ db, int accumulate)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = t >> 3, sub = t & 7;
    float s = 0.f;
    float *o = nullptr;
    if (idx < N * K) {
        const int n = idx / K, k = idx - n * K;
        const float *src = partial + (size_t)n * Kp + k;
        for (int p = sub; p < P; p += 8) s += src[(size_t)p * Np * Kp];
        o = dW + (size_t)n * lddw + k;
    } else if (db && idx < N * K + N) {
        const int n = idx - N * K;
        for (int p = sub; p < P; p += 8) s += bias_partial[(size_t)p * Np + n];
        o = db + n;
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (o && sub == 0) *o = accumulate ? *o + s : s;
}

static void wgrad_plan(int M, int N, int K, int *P, int *rows, int *Np, int *Kp)
{
    *Np = (N + 127) / 128 * 128;
    *Kp = (K + 127) / 128 * 128;
    const int tiles = (*Np / 128) * (*Kp / 128);
    int p = 1024 / tiles;                           // one wave per SIMD on 256 CUs
    if (p < 1) p = 1;
    int r = (M + p - 1) / p;
    if (r < 256) r = 256;                           // do not slice thinner than 256 rows
    r = (r + 7) / 8 * 8;
    *rows = r;
    *P = M > 0 ? (M + r - 1) / r : 1;
}

}  // namespace hnr

using namespace hnr;

extern "C" int64_t hnr_linear_wgrad_scratch_elems(int M, int N, int K)
{
    if (M < 0 || N <= 0 || K <= 0) return 0;
    int P, rows, Np, Kp;
    wgrad_plan(M, N, K, &P, &rows, &Np, &Kp);
    return (int64_t)P * Np * Kp + (int64_t)P * Np;
}

extern "C" int hnr_linear_f32_wgrad(const float *d_dZ, int ldz, const float *d_X, int ldx, int M, int N, int K, float *d_dW, int lddw,
                                    float *d_db, int accumulate, float *d_scratch, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || ldz < N || ldx < K || (ldz & 3) || (ldx & 3) || lddw < K) {
        set_error("hnr_linear_f32_wgrad: bad sizes (M=%d N=%d K=%d ldz=%d ldx=%d lddw=%d; ldz/ldx must be multiples of 4)", M, N, K, ldz, ldx, lddw);
        return HNR_ERR_BADARG;
    }
    if (!d_dW || !d_scratch || (M > 0 && (!d_dZ || !d_X || ((uintptr_t)d_dZ & 15) || ((uintptr_t)d_X & 15)))) {
        set_error("hnr_linear_f32_wgrad: NULL or unaligned pointer"); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    int P, rows, Np, Kp;
    wgrad_plan(M, N, K, &P, &rows, &Np, &Kp);
    float *partial = d_scratch, *bias_partial = d_scratch + (size_t)P * Np * Kp;
    dim3 grid(P, Np / 128, Kp / 128);
    linear_wgrad_kernel<<<grid, 64, 0, st>>>(d_dZ, ldz, d_X, ldx, M, rows, Np, Kp, partial, d_db ? bias_partial : nullptr);
    linear_wgrad_reduce_kernel<<<cdiv(((int64_t)N * K + N) * 8, 256), 256, 0, st>>>(partial, bias_partial, P, Np, Kp, N, K, d_dW, lddw, d_db, accumulate);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
