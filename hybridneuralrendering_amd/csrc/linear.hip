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

// K-assignment of the 32x32x2 MFMA inside a 32-wide K tile: the instruction takes k = 0 from lanes 0-31 and k = 1 from
// lanes 32-63; any pairing of k values works as long as A and W agree, so in quarter q (8 k values) lane-half h reads
// the 16-B chunk 2q+h of its row ONCE (ds_read_b128) and feeds its 4 floats to 4 consecutive MFMAs.
//
// Both tiles are copied global -> LDS by the DMA path (global_load_lds_dwordx4: no VGPRs, no ds_write).  The LDS image must
// be lane-linear, so rows are [32] un-padded and the 16-B chunks of a row are XOR-swizzled by (row & 7) through the per-lane
// SOURCE address; the fragment read applies the same XOR (2-way conflict instead of 8-way).  W (L2-resident) runs one
// tile ahead in 2 LDS buffers; A (streamed from HBM once) runs TWO tiles ahead in 3 LDS buffers, kept in flight across the
// barrier by a counted `s_waitcnt vmcnt(N)` + raw `s_barrier` (a plain __syncthreads() would drain the DMA queue).
// A partial last K tile (K % 32 != 0) is copied like the others; its k >= K values are zeroed in the fragment registers.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to the 1 KiB at LDS byte address `lds_base`
// (wave-uniform, goes through M0).  Written in inline asm ON PURPOSE: hipcc (ROCm 7.2) tracks a builtin LDS-DMA as a pending
// LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read, which drains the copy queue every K tile; through asm
// the copies are invisible to that pass and are ordered by the counted waits + barriers of the pipeline below.
// (M0 is otherwise unused in this kernel: gfx950 LDS instructions do not read it.)
__device__ __forceinline__ void lds_dma16(const float *src, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}

// SIDE = 1 compiles the side-operand epilogue in (its index / value arrays cost registers: with it in the plain kernel the
// 128x256 variant spilled 68 B per lane and the big layers ran 4 % slower).
//
// PAIR = 1 (TN == 2 only) deals the output columns of a wave's 64-column strip to its two MFMA tiles as {2c + j} instead of
// {32j + c}: the W rows are permuted through the DMA source address (free), and lane c then owns the ADJACENT columns 2c, 2c+1
// of every row it holds, so the epilogue issues one 8-B store (and one 8-B side-operand load) where it issued two 4-B ones.
// The store tail of a tile is bound by the number of store INSTRUCTIONS, not by bytes.
template <int TM, int TN, int ACT, int DBG = 0, int WN = 2, int SIDE = 0, int PAIR = 0>   // ACT: 0 none, 1 LeakyReLU(slope); DBG: ablation switches (probe only)
__global__ __launch_bounds__(128 * WN, 2) void linear_f32_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Wp, int K_pad,
                                                         const float *__restrict__ bias_p, float *__restrict__ C, int ldc,
                                                         int M, int N, int K, float slope,
                                                         const float *R, const int32_t *__restrict__ ridx, int ldr,
                                                         int r_cols, int r_mode)
{
    // workgroup = 2 x WN waves; each wave owns TM x TN MFMA tiles of 32x32
    constexpr int NT = 128 * WN;                // threads
    constexpr int NW = NT / 64;                 // waves
    constexpr int BM = 64 * TM, BN = 32 * TN * WN;
    constexpr int W_DMA = (BN / 8) / NW;        // 1-KiB DMA pieces (8 rows x 128 B) per wave for the W tile
    constexpr int A_DMA = (BM / 8) / NW;        // ... and for the A tile
    static_assert((BN / 8) % NW == 0 && (BM / 8) % NW == 0, "tiles must split evenly over the waves");
    // dynamic LDS, one object (sized by the launcher: (3*BM + 2*BN) * BK floats)
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    float *const As = lds_all;                                             // A: two tiles ahead (3 buffers)
    float *const Ws = lds_all + 3 * BM * BK;                               // W: one tile ahead (2 buffers)
    const unsigned lds_a0 = (unsigned)(uintptr_t)As, lds_w0 = (unsigned)(uintptr_t)Ws;   // LDS byte addresses

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

    // DMA: piece p = wave + NW*i covers rows 8p..8p+7 of a tile; lane l lands at byte 16*l of the piece, i.e. (row 8p + l/8,
    // chunk position l%8), and fetches the global chunk (l%8) ^ (l/8) of that row.
    const int swz = ((lane & 7) ^ (lane >> 3)) << 2;
    // rows past M are clamped to the last row (never stored); chunks past lda (partial last K tile) are clamped into the
    // row -- their values are zeroed in the fragment registers by compute_partial, they never enter a product
    auto dma_a = [&](int buf, int mt, int kt) {
        int kk = kt * BK + swz;
        kk = kk + 4 <= lda ? kk : lda - 4;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) {
            const int piece = wave + NW * i;
            const int gm = mt * BM + 8 * piece + (lane >> 3);
            const float *src = A + (size_t)(gm < M ? gm : M - 1) * lda + kk;
            lds_dma16(src, lds_a0 + (unsigned)((buf * BM * BK + piece * 256) * 4));
        }
    };
    auto dma_w = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) {
            const int piece = wave + NW * i;
            int rl = 8 * piece + (lane >> 3);                         // LDS row = (strip, tile j, lane c)
            if (PAIR) rl = (rl & ~63) + 2 * (rl & 31) + ((rl >> 5) & 1);   // ... holds the W row of output column strip + 2c + j
            const float *src = Wp + (size_t)(n0 + rl) * K_pad + kt * BK + swz;
            lds_dma16(src, lds_w0 + (unsigned)((buf * BN * BK + piece * 256) * 4));
        }
    };
    const int frag_a = (wr * 32 * TM + (lane & 31)) * BK;
    const int frag_w = (wc * 32 * TN + (lane & 31)) * BK;
    const int frag_h = lane >> 5, frag_x = lane & 7;
    auto frag_load = [&](float4 (&a)[TM], float4 (&b)[TN], const float *ab, const float *wb, int qt) {
        const int pos = ((2 * qt + frag_h) ^ frag_x) << 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(ab + i * 32 * BK + pos);
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
    static_assert(!PAIR || TN == 2, "PAIR needs two MFMA tiles per wave along N");
    auto col_of = [&](int j) { return n0 + wc * 32 * TN + (PAIR ? 2 * (lane & 31) + j : j * 32 + (lane & 31)); };
    float bias_v[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_v[j] = bias_p[col_of(j)];
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    auto epilogue = [&](int mt) {
        const int row0 = mt * BM + wr * 32 * TM + 4 * (lane >> 5);
        if (SIDE && R != nullptr) {
            // side operand R (row stride ldr), applied to the columns < r_cols:
            //   r_mode 0: gathered addend, v = act(acc + bias + R[ridx[m], n])   (ridx == NULL: row m itself, e.g. R == C for "+=")
            //   r_mode 1: LeakyReLU derivative of a stored activation, v = (acc + bias) * (R[m, n] > 0 ? 1 : slope)  (backward)
            // This lane's 16*TM row indices are fetched first, then all side values of one column block at once (the loads
            // are independent, so they overlap instead of forming index -> value -> store chains).
            if (PAIR) {
                // launcher guarantees: r_cols, ldr, ldc, N even; R and C 8-B aligned
                const int gn = col_of(0);
                const bool use = gn < r_cols;
                const int gn_safe = use ? gn : r_cols - 2;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int rid[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        const int gmc = gm < M ? gm : M - 1;
                        rid[r] = ridx ? ridx[gmc] : gmc;
                    }
#pragma unroll
                    for (int rb = 0; rb < 16; rb += 8) {
                        float2 add[8];                             // 8 independent 8-B loads in flight
#pragma unroll
                        for (int r = 0; r < 8; ++r) add[r] = *reinterpret_cast<const float2 *>(R + (size_t)rid[rb + r] * ldr + gn_safe);
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const int rr = rb + r;
                            const int gm = row0 + i * 32 + (rr & 3) + 8 * (rr >> 2);
                            float v0 = acc[i][0][rr] + bias_v[0], v1 = acc[i][1][rr] + bias_v[1];
                            if (r_mode == 0) {
                                v0 += use ? add[r].x : 0.f;
                                v1 += use ? add[r].y : 0.f;
                                if (ACT == 1) { v0 = v0 > 0.f ? v0 : v0 * slope; v1 = v1 > 0.f ? v1 : v1 * slope; }
                            } else {
                                v0 *= (use && !(add[r].x > 0.f)) ? slope : 1.f;
                                v1 *= (use && !(add[r].y > 0.f)) ? slope : 1.f;
                            }
                            if (gm < M && gn < N) *reinterpret_cast<float2 *>(C + (size_t)gm * ldc + gn) = make_float2(v0, v1);
                            acc[i][0][rr] = 0.f;
                            acc[i][1][rr] = 0.f;
                        }
                    }
                }
                return;
            }
            int rid[TM][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    const int gmc = gm < M ? gm : M - 1;
                    rid[i][r] = ridx ? ridx[gmc] : gmc;
                }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = col_of(j);
                const bool use = gn < r_cols;
                const int gn_safe = use ? gn : r_cols - 1;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float add[16];                               // 16 independent loads in flight per (i, j) block
#pragma unroll
                    for (int r = 0; r < 16; ++r) add[r] = R[(size_t)rid[i][r] * ldr + gn_safe];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        float v = acc[i][j][r] + bias_v[j];
                        if (r_mode == 0) {
                            v += use ? add[r] : 0.f;
                            if (ACT == 1) v = v > 0.f ? v : v * slope;
                        } else {
                            v *= (use && !(add[r] > 0.f)) ? slope : 1.f;
                        }
                        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = v;
                        acc[i][j][r] = 0.f;
                    }
                }
            }
            return;
        }
        if (PAIR) {
            const int gn = col_of(0);                              // even; N and ldc even, C 8-B aligned (launcher)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    float v0 = acc[i][0][r] + bias_v[0], v1 = acc[i][1][r] + bias_v[1];
                    if (ACT == 1) { v0 = v0 > 0.f ? v0 : v0 * slope; v1 = v1 > 0.f ? v1 : v1 * slope; }
                    if ((DBG != 4 || v0 == -1.2345e30f) && gm < M && gn < N) *reinterpret_cast<float2 *>(C + (size_t)gm * ldc + gn) = make_float2(v0, v1);
                    acc[i][0][r] = 0.f;
                    acc[i][1][r] = 0.f;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = col_of(j);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bias_v[j];
                    if (ACT == 1) v = v > 0.f ? v : v * slope;
                    if ((DBG != 4 || v == -1.2345e30f) && gm < M && gn < N) C[(size_t)gm * ldc + gn] = v;
                    acc[i][j][r] = 0.f;
                }
            }
        }
    };
    auto compute_full = [&](int abuf, int wbuf) {              // 4 software-pipelined quarters of 4 MFMA steps
        const float *ab = As + abuf * BM * BK + frag_a;
        const float *wb = Ws + wbuf * BN * BK + frag_w;
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
    auto compute_partial = [&](int abuf, int wbuf, int kvalid) {   // ceil(k_valid / 8) quarters; A values with k >= K zeroed
        const float *ab = As + abuf * BM * BK + frag_a;
        const float *wb = Ws + wbuf * BN * BK + frag_w;
        const int nquart = (kvalid + 7) >> 3;
#pragma unroll 1
        for (int qt = 0; qt < nquart; ++qt) {
            float4 a0[TM], b0[TN];
            frag_load(a0, b0, ab, wb, qt);
            const int kb = 8 * qt + 4 * frag_h;                    // this lane's chunk holds k = kb .. kb+3 of the tile
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i].x = kb < kvalid ? a0[i].x : 0.f;
                a0[i].y = kb + 1 < kvalid ? a0[i].y : 0.f;
                a0[i].z = kb + 2 < kvalid ? a0[i].z : 0.f;
                a0[i].w = kb + 3 < kvalid ? a0[i].w : 0.f;
            }
            mfma16(a0, b0);
        }
    };

    // PERSISTENT over M tiles: the (M tile, K tile) pairs of this workgroup form one flat software pipeline f = 0, 1, 2, ...
    //   iteration f:  [partial A(f+1) parked from registers, if any]  DMA W(f+1) ; DMA A(f+2) (or its register loads) ;
    //                 MFMAs of tile f ; [epilogue] ; s_waitcnt vmcnt(#ops issued for A(f+2)) ; s_barrier
    // so that A(f+1) and W(f+1) have landed for everyone while A(f+2) stays in flight across the barrier.
    const int mt0 = blockIdx.x;
    if (mt0 >= n_mtiles) return;
    const int kvalid_last = K - (nk - 1) * BK;
    auto next_tile = [&](int &m, int &k) { if (++k == nk) { k = 0; m += (int)gridDim.x; } };
    auto barrier_keep_a = [&]() {                               // the A_DMA youngest copies (tile f+2) stay in flight
        wait_vmcnt<A_DMA>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto barrier_drain = [&]() {
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    int m1 = mt0, k1 = 0;                                      // cursor: tile f+1
    next_tile(m1, k1);
    int m2 = m1, k2 = k1;                                      // cursor: tile f+2
    if (m2 < n_mtiles) next_tile(m2, k2);
    // prologue: W(0), A(0), A(1); everything lands before the first barrier
    dma_w(0, 0);
    dma_a(0, mt0, 0);
    if (m1 < n_mtiles) dma_a(1, m1, k1);
    barrier_drain();

    int f = 0;
    auto stage = [&]() -> bool {                                  // issue the copies for tiles f+1 (W) and f+2 (A), in that order
        bool a_issued = false;
        if (DBG == 0 || DBG == 4) {
            if (m1 < n_mtiles) dma_w((f + 1) & 1, k1);                // buffer last read in iteration f-1 (barrier passed)
            if (m2 < n_mtiles) { dma_a((f + 2) % 3, m2, k2); a_issued = true; }   // buffer (f+2)%3 = (f-1)%3: same argument
        }
        return a_issued;
    };
    auto advance = [&]() {
        m1 = m2; k1 = k2;
        if (m2 < n_mtiles) next_tile(m2, k2);
        ++f;
    };
#pragma unroll 1
    for (int mt = mt0; mt < n_mtiles; mt += gridDim.x) {
#pragma unroll 1
        for (int kt = 0; kt + 1 < nk; ++kt) {                     // full tiles
            const bool a_issued = stage();
            __builtin_amdgcn_sched_barrier(0);                    // keep the DMA issue ABOVE the MFMAs
            compute_full(f % 3, f & 1);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            if (a_issued) barrier_keep_a(); else barrier_drain();
        }
        {                                                         // last K tile of this M tile, then its results
            stage();
            __builtin_amdgcn_sched_barrier(0);
            if (kvalid_last == BK) compute_full(f % 3, f & 1);
            else compute_partial(f % 3, f & 1, kvalid_last);     // also zeroes the k >= K values of the copied tile
            __builtin_amdgcn_sched_barrier(0);
            epilogue(mt);                                         // its stores are younger than the DMA: drain here
            advance();
            barrier_drain();
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

static int linear_launch(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, float *d_C, int ldc, int M, int N, int K,
                         int act, float slope, const float *d_R, const int32_t *d_ridx, int ldr, int r_cols, int r_mode, void *stream);

extern "C" int hnr_linear_f32(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, float *d_C, int ldc,
                              int M, int N, int K, int act, float slope, void *stream)
{
    return linear_launch(d_A, lda, d_Wp, d_bias_p, d_C, ldc, M, N, K, act, slope, nullptr, nullptr, 0, 0, 0, stream);
}

extern "C" int hnr_linear_f32_gather_add(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, const float *d_R,
                                         const int32_t *d_ridx, int ldr, float *d_C, int ldc, int M, int N, int K, int act,
                                         float slope, void *stream)
{
    if (M > 0 && (!d_R || !d_ridx || ldr < N)) { set_error("hnr_linear_f32_gather_add: bad addend arguments"); return HNR_ERR_BADARG; }
    return linear_launch(d_A, lda, d_Wp, d_bias_p, d_C, ldc, M, N, K, act, slope, d_R, d_ridx, ldr, N, 0, stream);
}

extern "C" int hnr_linear_f32_side(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, const float *d_R,
                                   const int32_t *d_ridx, int ldr, int r_cols, int r_mode, float *d_C, int ldc, int M, int N, int K,
                                   int act, float slope, void *stream)
{
    if (M > 0 && (!d_R || r_cols <= 0 || r_cols > N || ldr < r_cols || (r_mode != 0 && r_mode != 1) || (r_mode == 1 && (act != 0 || d_ridx)))) {
        set_error("hnr_linear_f32_side: bad side-operand arguments"); return HNR_ERR_BADARG;
    }
    return linear_launch(d_A, lda, d_Wp, d_bias_p, d_C, ldc, M, N, K, act, slope, d_R, d_ridx, ldr, r_cols, r_mode, stream);
}

#define HNR_LINEAR_ARGS d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope, d_R, d_ridx, ldr, r_cols, r_mode
#define HNR_LINEAR_LAUNCH(TM_, TN_, ACT_, DBG_, WN_, PAIR_, THREADS_, LDS_)                                                   \
    do {                                                                                                                     \
        if (d_R) linear_f32_kernel<TM_, TN_, ACT_, DBG_, WN_, 1, PAIR_><<<grid, THREADS_, LDS_, st>>>(HNR_LINEAR_ARGS);      \
        else linear_f32_kernel<TM_, TN_, ACT_, DBG_, WN_, 0, PAIR_><<<grid, THREADS_, LDS_, st>>>(HNR_LINEAR_ARGS);          \
    } while (0)

static int linear_launch(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, float *d_C, int ldc, int M, int N, int K,
                         int act, float slope, const float *d_R, const int32_t *d_ridx, int ldr, int r_cols, int r_mode, void *stream)
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
    // column pairing (8-B epilogue accesses) needs even strides / column counts and 8-B aligned bases
    static int pair_env = -1;
    if (pair_env < 0) { const char *e = getenv("HNR_LINEAR_PAIR"); pair_env = e ? atoi(e) : 1; }
    const bool pair = pair_env && N >= 128 && !(N & 1) && !(ldc & 1) && !((uintptr_t)d_C & 7) &&
                      (!d_R || (!(ldr & 1) && !(r_cols & 1) && !((uintptr_t)d_R & 7)));
    if (N > 128 && Np % 256 == 0 && dbg != 3) {
        // 128 x 256 block tile, 8 waves (2 x 4): the A tile is fetched ONCE for all 256 output columns.  (With two 128-column
        // workgroups per M tile the PMC counters show A coming from HBM twice: 56.7 GB fetched per 24.3 GB of A.)
        const int ny = Np / 256;
        int gx = n_cu / ny;                                       // one 512-thread workgroup per CU (110 KB of LDS)
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (dbg == 4) { if (pair) HNR_LINEAR_LAUNCH(2, 2, 1, 4, 4, 1, 512, 114688); else HNR_LINEAR_LAUNCH(2, 2, 1, 4, 4, 0, 512, 114688); }
        else if (pair) { if (act) HNR_LINEAR_LAUNCH(2, 2, 1, 0, 4, 1, 512, 114688); else HNR_LINEAR_LAUNCH(2, 2, 0, 0, 4, 1, 512, 114688); }
        else if (act) HNR_LINEAR_LAUNCH(2, 2, 1, 0, 4, 0, 512, 114688);
        else HNR_LINEAR_LAUNCH(2, 2, 0, 0, 4, 0, 512, 114688);
    } else if (N >= 128) {
        const int ny = Np / 128;
        int gx = (2 * n_cu) / ny;
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (dbg == 1) HNR_LINEAR_LAUNCH(2, 2, 1, 1, 2, 0, 256, 81920);
        else if (dbg == 2) HNR_LINEAR_LAUNCH(2, 2, 1, 2, 2, 0, 256, 81920);
        else if (pair) { if (act) HNR_LINEAR_LAUNCH(2, 2, 1, 0, 2, 1, 256, 81920); else HNR_LINEAR_LAUNCH(2, 2, 0, 0, 2, 1, 256, 81920); }
        else if (act) HNR_LINEAR_LAUNCH(2, 2, 1, 0, 2, 0, 256, 81920);
        else HNR_LINEAR_LAUNCH(2, 2, 0, 0, 2, 0, 256, 81920);
    } else {
        const int ny = Np / 64;
        int gx = (2 * n_cu) / ny;
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (act) HNR_LINEAR_LAUNCH(2, 1, 1, 0, 2, 0, 256, 65536);
        else HNR_LINEAR_LAUNCH(2, 1, 0, 0, 2, 0, 256, 65536);
    }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
