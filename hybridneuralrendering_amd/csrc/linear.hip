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
// BK = 32.  Every MFMA fragment is one ds_read_b128 (see the K-assignment note below); the A tile is staged through
// registers one tile ahead, the W tile goes global -> LDS by DMA (global_load_lds_dwordx4).
// Weights are pre-packed once per checkpoint into a zero-padded [N_pad][K_pad] image, so the W loads
// are unguarded 16-B loads; A rows are guarded (zero-filled) for m >= M and k >= K.
#include <stdlib.h>

#include "hnr_common.h"

namespace hnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;     // A rows: 36 floats = 144 B (16-B aligned; 9*row mod 16 spreads ds_read_b128 over all banks)

// K-assignment of the 32x32x2 MFMA inside a 32-wide K tile: the instruction takes k = 0 from lanes 0-31 and k = 1 from
// lanes 32-63; any pairing of k values works as long as A and W agree, so in quarter q (8 k values) lane-half h reads
// the 16-B chunk 2q+h of its row ONCE (ds_read_b128) and feeds its 4 floats to 4 consecutive MFMAs.
//   A tile: staged through registers (rows clamped, K tail masked), stored as [row][36] in natural k order;
//   W tile: copied global -> LDS by the DMA path (global_load_lds_dwordx4, no VGPRs, no ds_write); the LDS image must be
//           lane-linear, so rows are [32] un-padded and the 16-B chunks of a row are XOR-swizzled by (row & 7) through the
//           per-lane SOURCE address; the fragment read applies the same XOR (2-way conflict instead of 8-way).
template <int TM, int TN, int ACT, int DBG = 0, int WN = 2>   // ACT: 0 none, 1 LeakyReLU(slope); DBG: ablation switches (probe only)
__global__ __launch_bounds__(128 * WN, 2) void linear_f32_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Wp, int K_pad,
                                                         const float *__restrict__ bias_p, float *__restrict__ C, int ldc,
                                                         int M, int N, int K, float slope)
{
    // workgroup = 2 x WN waves; each wave owns TM x TN MFMA tiles of 32x32
    constexpr int NT = 128 * WN;                // threads
    constexpr int NW = NT / 64;                 // waves
    constexpr int RPP = NT / 8;                 // tile rows covered by one pass of the A loader (8 float4 per 32-float row)
    constexpr int BM = 64 * TM, BN = 32 * TN * WN;
    constexpr int A_F4 = BM * BK / 4 / NT;      // float4 per thread for the A tile
    constexpr int W_DMA = (BN / 8) / NW;        // 1-KiB DMA pieces (8 rows x 128 B) per wave for the W tile
    static_assert((BN / 8) % NW == 0, "W tile must split evenly over the waves");
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDS_LD];     // double-buffered: one barrier per K tile
    __shared__ __attribute__((aligned(16))) float Ws[2 * BN * BK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int n0 = blockIdx.y * BN;
    const int n_mtiles = (M + BM - 1) / BM;
    const int nk = (K + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A staging registers.  Loads are UNCONDITIONAL (row and column clamped into the buffer) and masked afterwards with
    // selects, so no branch consumes a loaded value early: the loads stay in flight across the tile's MFMAs.
    float4 ra[A_F4];
    const int c4 = (tid & 7) << 2;
    const int trow = tid >> 3;                  // row of this thread's first float4 inside a tile (+RPP per extra float4)
    auto load_a = [&](int mt, int k0) {
        const int gk = k0 + c4;
        const int gk_safe = gk + 4 <= lda ? gk : lda - 4;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int gm = mt * BM + trow + RPP * i;
            ra[i] = *reinterpret_cast<const float4 *>(A + (size_t)(gm < M ? gm : M - 1) * lda + gk_safe);
        }
    };
    auto mask_k = [&](int k0) {                 // only the last K tile can hold k >= K (rows >= M are never stored)
        const int gk = k0 + c4;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            ra[i].x = gk < K ? ra[i].x : 0.f;
            ra[i].y = gk + 1 < K ? ra[i].y : 0.f;
            ra[i].z = gk + 2 < K ? ra[i].z : 0.f;
            ra[i].w = gk + 3 < K ? ra[i].w : 0.f;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i)
            *reinterpret_cast<float4 *>(As + buf * BM * LDS_LD + (trow + RPP * i) * LDS_LD + c4) = ra[i];
    };
    // W tile kt -> LDS buffer `buf`: piece p = wave + NW*i covers rows 8p..8p+7; lane l lands at byte 16*l of the piece,
    // i.e. (row 8p + l/8, chunk position l%8), and fetches the global chunk (l%8) ^ (l/8) of that row.
    const int w_src = ((lane >> 3) * K_pad) + (((lane & 7) ^ (lane >> 3)) << 2);
    auto dma_w = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) {
            const int piece = wave + NW * i;
            const float *src = Wp + (size_t)(n0 + 8 * piece) * K_pad + kt * BK + w_src;
            float *dst = Ws + buf * BN * BK + piece * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    const int frag_a = (wr * 32 * TM + (lane & 31)) * LDS_LD + 4 * (lane >> 5);
    const int frag_w_row = (wc * 32 * TN + (lane & 31)) * BK;
    const int frag_w_h = lane >> 5, frag_w_x = lane & 7;
    auto frag_load = [&](float4 (&a)[TM], float4 (&b)[TN], const float *ab, const float *wb, int qt) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(ab + i * 32 * LDS_LD + qt * 8);
        const int pos = ((2 * qt + frag_w_h) ^ frag_w_x) << 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(wb + j * 32 * BK + pos);
    };
    auto mfma16 = [&](const float4 (&a)[TM], const float4 (&b)[TN]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bv = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_v[j] = bias_p[n0 + wc * 32 * TN + j * 32 + (lane & 31)];
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    auto epilogue = [&](int mt) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wc * 32 * TN + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = mt * BM + wr * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] + bias_v[j];
                    if (ACT == 1) v = v > 0.f ? v : v * slope;
                    if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v;
                    acc[i][j][r] = 0.f;
                }
            }
        }
    };

    // PERSISTENT over M tiles: the (M tile, K tile) pairs of this workgroup form one flat software pipeline, so the first
    // K tile of the next M tile is already in flight while the current one finishes and its results are stored.  With
    // K = 256..284 an M tile is only 8-9 K tiles long; a prologue/epilogue bubble per M tile would cost ~25 %.
    // Pipeline per K tile t:  [regs hold A(t+1), loaded during t-1]  ds_write A(t+1) + DMA W(t+1) -> other LDS buffer ; issue
    // global loads A(t+2) -> regs ; MFMAs of tile t ; ONE barrier.  Stores, DMA and loads sit in the shadow of the MFMAs.
    int mt = blockIdx.x;
    if (mt >= n_mtiles) return;
    auto compute_full = [&](int cur) {                        // 4 software-pipelined quarters of 4 MFMA steps
        const float *ab = As + cur * BM * LDS_LD + frag_a;
        const float *wb = Ws + cur * BN * BK + frag_w_row;
        float4 a0[TM], b0[TN], a1[TM], b1[TN];
        frag_load(a0, b0, ab, wb, 0);
        frag_load(a1, b1, ab, wb, 1);
        mfma16(a0, b0);
        frag_load(a0, b0, ab, wb, 2);
        mfma16(a1, b1);
        frag_load(a1, b1, ab, wb, 3);
        mfma16(a0, b0);
        mfma16(a1, b1);
    };
    auto compute_partial = [&](int cur, int kvalid) {         // ceil(k_valid / 8) quarters
        const float *ab = As + cur * BM * LDS_LD + frag_a;
        const float *wb = Ws + cur * BN * BK + frag_w_row;
        const int nquart = (kvalid + 7) >> 3;
#pragma unroll 1
        for (int qt = 0; qt < nquart; ++qt) {
            float4 a0[TM], b0[TN];
            frag_load(a0, b0, ab, wb, qt);
            mfma16(a0, b0);
        }
    };
    // flat position of "the tile after (m, k)"
    auto next_tile = [&](int &m, int &k) { if (++k == nk) { k = 0; m += (int)gridDim.x; } };
    const int kvalid_last = K - (nk - 1) * BK;

    int pm = mt, pk = 0;                                       // (M tile, K tile) of the A data currently in the staging registers
    load_a(pm, 0);
    dma_w(0, 0);
    if (nk == 1) mask_k(0);
    store_a(0);
    next_tile(pm, pk);
    if (pm < n_mtiles) load_a(pm, pk * BK);                    // tile 1 -> regs
    __syncthreads();                                           // (drains the DMA: vmcnt(0) precedes the barrier)
    int it = 0;
    auto stage_next = [&](int cur) {
        if (pm < n_mtiles) {                                      // regs hold A of the next tile (pm, pk): park it in the other buffer
            if (pk + 1 == nk) mask_k(pk * BK);
            if (DBG < 2) { store_a(cur ^ 1); dma_w(cur ^ 1, pk); }   // that buffer was last read one iteration ago (barrier passed)
            next_tile(pm, pk);
            if (pm < n_mtiles && DBG == 0) load_a(pm, pk * BK);   // tile after next -> regs, in flight during the MFMAs
        }
    };
#pragma unroll 1
    for (; mt < n_mtiles; mt += gridDim.x) {
#pragma unroll 1
        for (int kt = 0; kt + 1 < nk; ++kt, ++it) {              // full tiles
            const int cur = it & 1;
            stage_next(cur);
            // keep the stores/prefetch ABOVE the MFMAs: without this fence hipcc sinks the global loads below the last MFMA
            // and waits for them at once (vmcnt(0)), exposing the whole HBM/L2 latency every tile
            __builtin_amdgcn_sched_barrier(0);
            compute_full(cur);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
        {                                                         // last K tile of this M tile, then its results
            const int cur = it & 1;
            stage_next(cur);
            __builtin_amdgcn_sched_barrier(0);
            if (kvalid_last >= 25) compute_full(cur);
            else compute_partial(cur, kvalid_last);
            __builtin_amdgcn_sched_barrier(0);
            epilogue(mt);
            __syncthreads();
            ++it;
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
    // persistent workgroups: 2 per CU fit (registers + 2 x LDS double buffers); they stride over the M tiles
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int n_mtiles = cdiv(M, 128);
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("HNR_LINEAR_DBG"); dbg = e ? atoi(e) : 0; }
    if (N > 128 && Np % 256 == 0 && dbg != 3) {
        // 128 x 256 block tile, 8 waves (2 x 4): the A tile is fetched ONCE for all 256 output columns.  (With two 128-column
        // workgroups per M tile the PMC counters show A coming from HBM twice: 56.7 GB fetched per 24.3 GB of A.)
        const int ny = Np / 256;
        int gx = n_cu / ny;                                       // one 512-thread workgroup per CU (110 KB of LDS)
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (act) linear_f32_kernel<2, 2, 1, 0, 4><<<grid, 512, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else linear_f32_kernel<2, 2, 0, 0, 4><<<grid, 512, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
    } else if (N >= 128) {
        const int ny = Np / 128;
        int gx = (2 * n_cu) / ny;
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (dbg == 1) linear_f32_kernel<2, 2, 1, 1><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else if (dbg == 2) linear_f32_kernel<2, 2, 1, 2><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else if (act) linear_f32_kernel<2, 2, 1><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else linear_f32_kernel<2, 2, 0><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
    } else {
        const int ny = Np / 64;
        int gx = (2 * n_cu) / ny;
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (act) linear_f32_kernel<2, 1, 1><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
        else linear_f32_kernel<2, 1, 0><<<grid, 256, 0, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope);
    }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
