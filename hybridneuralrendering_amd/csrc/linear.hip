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
// be lane-linear, so rows are [32] un-padded and the 16-B chunks of a row are XOR-swizzled by ((row >> 1) & 7) through the per-lane
// SOURCE address; the fragment read applies the same XOR.  ds_read_b128 is served in four groups of 16 lanes (0-3, 12-15, 20-27 | 4-11,
// 16-19, 28-31 | ...): within a group the (row parity, chunk) pairs are then all different = all 64 banks, no conflict.  W (L2-resident) runs one
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

// Same copy with a wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset: no per-piece 64-bit VALU address.
__device__ __forceinline__ void lds_dma16_s(const float *base_uniform, unsigned byte_off, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(byte_off), "s"(base_uniform), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}

// SIDE = 1 compiles the side-operand epilogue in (its index / value arrays cost registers: with it in the plain kernel the
// 128x256 variant spilled 68 B per lane and the big layers ran 4 % slower).
//
// PAIR = 1 (TN == 2 only) deals the output columns of a wave's 64-column strip to its two MFMA tiles as {2c + j} instead of
// {32j + c}: the W rows are permuted through the DMA source address (free), and lane c then owns the ADJACENT columns 2c, 2c+1
// of every row it holds, so the epilogue issues one 8-B store (and one 8-B side-operand load) where it issued two 4-B ones.
// The store tail of a tile is bound by the number of store INSTRUCTIONS, not by bytes.
template <int TM, int TN, int ACT, int DBG = 0, int WN = 2, int SIDE = 0, int PAIR = 0, int MASK = 0>   // ACT: 0 none, 1 LeakyReLU(slope); DBG: ablation switches (probe only); MASK: 8-wave kernel with K % 32 != 0
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
    // dynamic LDS, one object (sized by the launcher: (3*BM + 2*BN) * BK floats; the 8-wave kernel: (4*BM + 3*BN) * BK)
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    float *const As = lds_all;                                             // A: two tiles ahead (3 buffers; 4 in the 8-wave kernel)
    float *const Ws = lds_all + (WN == 4 ? 4 : 3) * BM * BK;               // W: one tile ahead (2 buffers; 3 in the 8-wave kernel)
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
    static_assert(NW % 2 == 0, "piece parity must equal wave parity");
    const int swz = ((lane & 7) ^ ((lane >> 4) | ((wave & 1) << 2))) << 2;   // slot ^ ((row >> 1) & 7), row = 8 * piece + lane / 8
    // rows past M are clamped to the last row (never stored); chunks past lda (partial last K tile) are clamped into the
    // row -- their values are zeroed in the fragment registers by compute_partial, they never enter a product
    // per-lane byte offsets of the pieces, relative to a wave-uniform tile base: computed once
    unsigned a_off[A_DMA], w_off[W_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) a_off[i] = (unsigned)(8 * (wave + NW * i) + (lane >> 3)) * (unsigned)lda * 4u;
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) {
        int rl = 8 * (wave + NW * i) + (lane >> 3);                   // LDS row = (strip, tile j, lane c)
        if (PAIR) rl = (rl & ~63) + 2 * (rl & 31) + ((rl >> 5) & 1);   // ... holds the W row of output column strip + 2c + j
        w_off[i] = (unsigned)rl * (unsigned)K_pad * 4u + (unsigned)swz * 4u;
    }
    // rows past M are clamped to the last row (never stored); chunks past lda (partial last K tile) are clamped into the
    // row -- their values are zeroed in the fragment registers, they never enter a product
    auto dma_a = [&](int buf, int mt, int kt) {
        const int mt_s = __builtin_amdgcn_readfirstlane(mt), kt_s = __builtin_amdgcn_readfirstlane(kt);
        const float *base = A + (size_t)mt_s * BM * lda + kt_s * BK;
        const int rows_left = M - mt_s * BM;                          // >= 1
        int sw = swz;
        if (kt_s * BK + BK > lda) sw = sw < lda - 4 - kt_s * BK ? sw : lda - 4 - kt_s * BK;   // only a partial last K tile
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) {
            unsigned off = a_off[i];
            if (rows_left < BM) { const unsigned lim = (unsigned)(rows_left - 1) * (unsigned)lda * 4u; off = off < lim ? off : lim; }
            lds_dma16_s(base, off + (unsigned)sw * 4u, lds_a0 + (unsigned)((buf * BM * BK + (wave + NW * i) * 256) * 4));
        }
    };
    auto dma_w = [&](int buf, int kt) {
        const float *base = Wp + (size_t)n0 * K_pad + __builtin_amdgcn_readfirstlane(kt) * BK;
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) lds_dma16_s(base, w_off[i], lds_w0 + (unsigned)((buf * BN * BK + (wave + NW * i) * 256) * 4));
    };
    const int frag_a = (wr * 32 * TM + (lane & 31)) * BK;
    const int frag_w = (wc * 32 * TN + (lane & 31)) * BK;
    const int frag_h = lane >> 5, frag_x = (lane >> 1) & 7;
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
                    if (((DBG != 4 && !(DBG >= 16 && (DBG & 8))) || v0 == -1.2345e30f) && gm < M && gn < N) *reinterpret_cast<float2 *>(C + (size_t)gm * ldc + gn) = make_float2(v0, v1);
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

    if constexpr (WN == 4) {
        // 8-wave kernel (two waves per SIMD, one workgroup per CU).
        //   * tile f+1 is completely landed (and published by a barrier) while tile f is computed, so its first two fragment
        //     quarters are read into registers UNDER tile f's MFMAs -- a tile starts with its operands in registers;
        //   * the copy instructions (W of tile f+2, A of tile f+3) are issued between the MFMAs of the first quarter;
        //   * at the end of iteration f a counted wait leaves only A(f+3) in flight: tile f+2 has landed.
        // LDS: 4 A buffers (f .. f+3) + 3 W buffers (f .. f+2) = 160 KiB, the whole CU.
        // All per-tile bookkeeping is scalar and incremental (rolling LDS offsets, a running source pointer): the loop carries
        // about one scalar instruction per MFMA; every other instruction in the stream is a slot the matrix pipe can lose when
        // the partner wave of the SIMD is waiting at the barrier (profiles/README.md, dense-layer experiments).
        constexpr unsigned A_BUF_B = BM * BK * 4, W_BUF_B = BN * BK * 4, A_RING_B = 4 * A_BUF_B, W_RING_B = 3 * W_BUF_B;
        constexpr int ABL = DBG >= 16 ? DBG - 16 : DBG == 1 ? 1 : 0;   // probe only: 1 no copies, 2 no fragment reads, 4 no barriers
        constexpr bool MASKED = MASK != 0;                         // K % 32 != 0: the last K tile's k >= K values are zeroed
        const int mt0 = blockIdx.x;
        if (mt0 >= n_mtiles) return;
        const int kvalid_last = K - (nk - 1) * BK;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        const unsigned lds_a_wave = __builtin_amdgcn_readfirstlane(lds_a0) + (unsigned)wave_s * 1024u;
        const unsigned lds_w_wave = __builtin_amdgcn_readfirstlane(lds_w0) + (unsigned)wave_s * 1024u;
        auto barrier_plain = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        const long long t_begin = DBG == 5 ? clock64() : 0;
        long long t_retire = 0, t_barrier = 0, t_mark = 0;      // DBG == 5: cycles in the counted wait / from there to the barrier release
        long long t_line[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // DBG == 5: issue timeline of tile 100 of workgroup 0
        int tile_no = 0;
        long long t_kt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_tile0 = 0;       // DBG == 5: cycles per K-tile position, summed over the M tiles
        auto stamp = [&](int i) { if (DBG == 5 && tile_no == 100) t_line[i] = clock64(); };

        // copy cursor = the tile whose A is copied next; its W copy follows one iteration later
        int cm = mt0, ck = 0;
        const float *a_src = A + (size_t)mt0 * BM * lda;
        const long long a_wrap = (long long)gridDim.x * BM * lda - (long long)(nk - 1) * BK;   // from the last K tile to the next M tile
        unsigned a_lim = 0xffffffffu;                              // row clamp of the cursor's M tile (byte offset of its last valid row)
        auto set_lim = [&]() { const int rows_left = M - cm * BM; a_lim = rows_left < BM ? (unsigned)(rows_left - 1) * (unsigned)lda * 4u : 0xffffffffu; };
        set_lim();
        const float *const w_src0 = Wp + (size_t)n0 * K_pad;
        auto copy_a = [&](unsigned wr_off) {                       // A tile of the cursor -> LDS byte offset wr_off of the A ring
            int sw = swz;
            if (MASKED && ck == nk - 1) sw = sw < lda - 4 - ck * BK ? sw : lda - 4 - ck * BK;   // chunks past the row stride: clamped into the row
#pragma unroll
            for (int i = 0; i < A_DMA; ++i) {
                const unsigned off = a_off[i] < a_lim ? a_off[i] : a_lim;
                lds_dma16_s(a_src, off + (unsigned)sw * 4u, lds_a_wave + wr_off + (unsigned)(NW * i) * 1024u);
            }
        };
        auto copy_w = [&](int k, unsigned wr_off) {
            const float *src = w_src0 + k * BK;
#pragma unroll
            for (int i = 0; i < W_DMA; ++i) lds_dma16_s(src, w_off[i], lds_w_wave + wr_off + (unsigned)(NW * i) * 1024u);
        };
        auto advance = [&]() {                                     // cursor to the next tile of this workgroup
            if (ck + 1 < nk) { ++ck; a_src += BK; }
            else { ck = 0; cm += (int)gridDim.x; a_src += a_wrap; set_lim(); }
        };
        // prologue: W(0) A(0) W(1) A(1) A(2); everything but A(2) lands before the first barrier
        copy_w(0, 0);
        copy_a(0);
        advance();
        bool has_w = false;                                        // a tile whose A copy is out and whose W copy is still to be issued
        int k_w = 0;
        if (cm < n_mtiles) {
            copy_w(ck, W_BUF_B);
            copy_a(A_BUF_B);
            advance();
            if (cm < n_mtiles) { copy_a(2 * A_BUF_B); has_w = true; k_w = ck; advance(); }
        }
        if (has_w) wait_vmcnt<A_DMA>(); else wait_vmcnt<0>();
        barrier_plain();
        unsigned a_rd = 0, w_rd = 0;                               // LDS byte offsets of tile f in the rings
        unsigned a_wr = 3 * A_BUF_B, w_wr = 2 * W_BUF_B;           // ... of the buffers tile f+3 (A) / f+2 (W) are copied into

        float4 sa0[TM], sb0[TN], sa1[TM], sb1[TN], sa2[TM], sb2[TN], sa3[TM], sb3[TN];   // quarter q of the current tile lives in set q
        auto load_q = [&](float4 (&xa)[TM], float4 (&xb)[TN], unsigned a_b, unsigned w_b, int qt, int kvalid) {
            const float *ab = As + (a_b >> 2) + frag_a;
            const float *wb = Ws + (w_b >> 2) + frag_w;
            frag_load(xa, xb, ab, wb, qt);
            if (MASKED) {
                const int kb = 8 * qt + 4 * frag_h;                 // this lane's chunk holds k = kb .. kb+3 of the tile
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    xa[i].x = kb < kvalid ? xa[i].x : 0.f;
                    xa[i].y = kb + 1 < kvalid ? xa[i].y : 0.f;
                    xa[i].z = kb + 2 < kvalid ? xa[i].z : 0.f;
                    xa[i].w = kb + 3 < kvalid ? xa[i].w : 0.f;
                }
            }
        };
        auto mfma4 = [&](const float4 (&xa)[TM], const float4 (&xb)[TN], int e) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = e == 0 ? xa[i].x : e == 1 ? xa[i].y : e == 2 ? xa[i].z : xa[i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bv = e == 0 ? xb[j].x : e == 1 ? xb[j].y : e == 2 ? xb[j].z : xb[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
            }
        };
        const int kv_first = nk == 1 ? kvalid_last : BK;           // valid k of a tile with kt == 0
        load_q(sa0, sb0, 0, 0, 0, kv_first);
        load_q(sa1, sb1, 0, 0, 1, kv_first);
        if (ABL & 2) { load_q(sa2, sb2, 0, 0, 2, kv_first); load_q(sa3, sb3, 0, 0, 3, kv_first); }

        auto body = [&](int kv, int kv_next) __attribute__((always_inline)) {   // one K tile (valid k of this tile / of the next one)
            if (!(ABL & 2)) { load_q(sa2, sb2, a_rd, w_rd, 2, kv); load_q(sa3, sb3, a_rd, w_rd, 3, kv); }
            const bool do_w = !(ABL & 1) && has_w, do_a = !(ABL & 1) && cm < n_mtiles;
            stamp(0);
            mfma4(sa0, sb0, 0);
            __builtin_amdgcn_sched_barrier(0);                     // the copy issue stays between these MFMA groups
            stamp(1);
            if (do_w) copy_w(k_w, w_wr);
            stamp(2);
            __builtin_amdgcn_sched_barrier(0);
            mfma4(sa0, sb0, 1);
            mfma4(sa0, sb0, 2);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            if (do_a) copy_a(a_wr);
            stamp(4);
            __builtin_amdgcn_sched_barrier(0);
            mfma4(sa0, sb0, 3);
            __builtin_amdgcn_sched_barrier(0);
            stamp(5);
            // the first two quarters of tile f+1 go into the sets that have just been consumed
            // (past the last tile the reads hit a stale buffer and are never used)
            const unsigned a_rd1 = (a_rd + A_BUF_B) & (A_RING_B - 1), w_rd1 = w_rd + W_BUF_B == W_RING_B ? 0 : w_rd + W_BUF_B;
            if (!(ABL & 2)) load_q(sa0, sb0, a_rd1, w_rd1, 0, kv_next);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!MASKED || kv > 8) mfma4(sa1, sb1, e);      // a partial last K tile skips its empty quarters
            if (DBG == 5) __builtin_amdgcn_sched_barrier(0);
            stamp(6);
            if (!(ABL & 2)) load_q(sa1, sb1, a_rd1, w_rd1, 1, kv_next);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!MASKED || kv > 16) mfma4(sa2, sb2, e);
            if (DBG == 5) __builtin_amdgcn_sched_barrier(0);
            stamp(7);
#pragma unroll
            for (int e = 0; e < 4; ++e) if (!MASKED || kv > 24) mfma4(sa3, sb3, e);
            __builtin_amdgcn_sched_barrier(0);
            stamp(8);
            long long tw0 = 0;
            if (DBG == 5) tw0 = clock64();
            if (do_a) wait_vmcnt<A_DMA>(); else wait_vmcnt<0>();   // tile f+2 has landed; only A(f+3) stays in flight
            if (DBG == 5) { const long long t = clock64(); t_retire += t - tw0; t_mark = t; }
            stamp(9);
            has_w = cm < n_mtiles;                                 // the tile whose A went out now gets its W next iteration
            k_w = ck;
            if (has_w) advance();
            a_rd = a_rd1;
            w_rd = w_rd1;
            a_wr = (a_wr + A_BUF_B) & (A_RING_B - 1);
            w_wr = w_wr + W_BUF_B == W_RING_B ? 0 : w_wr + W_BUF_B;
        };
        if (DBG == 5) t_tile0 = clock64();
#pragma unroll 1
        for (int mc = mt0; mc < n_mtiles; mc += (int)gridDim.x) {
#pragma unroll 1
            for (int kt = 0; kt + 1 < nk; ++kt) {
                body(BK, kt + 2 == nk ? kvalid_last : BK);
                if (!(ABL & 4)) barrier_plain();
                if (DBG == 5) t_barrier += clock64() - t_mark;
                stamp(10);
                ++tile_no;
                if (DBG == 5) { const long long t = clock64(); t_kt[kt & 7] += t - t_tile0; t_tile0 = t; }
            }
            if constexpr (SIDE && PAIR) {
                // Side-operand layers: this lane's 32 row indices are fetched BEFORE the last K tile (its counted wait retires
                // them), and the side values are gathered in four batches of 8 rows, each batch requested before the previous
                // one is consumed -- the epilogue pays ~2 memory round trips instead of the 6 of the index -> value -> store chain
                // per 16 rows (the K=60 layer of the neighbour MLP spends half of its time in this epilogue).
                const int row0 = mc * BM + wr * 32 * TM + 4 * (lane >> 5);
                int rid[TM][16];
                if (R != nullptr) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int gm = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                            const int gmc = gm < M ? gm : M - 1;
                            rid[i][r] = ridx ? ridx[gmc] : gmc;
                        }
                }
                body(kvalid_last, kv_first);
                if (R != nullptr) {
                    const int gn = col_of(0);                      // launcher: r_cols, ldr, ldc, N even; R and C 8-B aligned
                    const bool use = gn < r_cols;
                    const int gn_safe = use ? gn : r_cols - 2;
                    float2 add_a[8], add_b[8];
                    auto gather8 = [&](float2 (&ad)[8], int i, int rb) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) ad[r] = *reinterpret_cast<const float2 *>(R + (size_t)rid[i][rb + r] * ldr + gn_safe);
                    };
                    auto finish8 = [&](const float2 (&ad)[8], int i, int rb) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const int rr = rb + r;
                            const int gm = row0 + i * 32 + (rr & 3) + 8 * (rr >> 2);
                            float v0 = acc[i][0][rr] + bias_v[0], v1 = acc[i][1][rr] + bias_v[1];
                            if (r_mode == 0) {
                                v0 += use ? ad[r].x : 0.f;
                                v1 += use ? ad[r].y : 0.f;
                                if (ACT == 1) { v0 = v0 > 0.f ? v0 : v0 * slope; v1 = v1 > 0.f ? v1 : v1 * slope; }
                            } else {
                                v0 *= (use && !(ad[r].x > 0.f)) ? slope : 1.f;
                                v1 *= (use && !(ad[r].y > 0.f)) ? slope : 1.f;
                            }
                            if (gm < M && gn < N) *reinterpret_cast<float2 *>(C + (size_t)gm * ldc + gn) = make_float2(v0, v1);
                            acc[i][0][rr] = 0.f;
                            acc[i][1][rr] = 0.f;
                        }
                    };
                    static_assert(TM == 2, "batch plan below is written for two row blocks per wave");
                    gather8(add_a, 0, 0);
                    gather8(add_b, 0, 8);
                    finish8(add_a, 0, 0);
                    gather8(add_a, 1, 0);
                    finish8(add_b, 0, 8);
                    gather8(add_b, 1, 8);
                    finish8(add_a, 1, 0);
                    finish8(add_b, 1, 8);
                } else {
                    epilogue(mc);
                }
            } else {
                body(kvalid_last, kv_first);
                epilogue(mc);                                      // stores are younger than the counted wait above
            }
            barrier_plain();
            if (DBG == 5) { const long long t = clock64(); t_kt[(nk - 1) & 7] += t - t_tile0; t_tile0 = t; ++tile_no; }
        }
        if (DBG == 5 && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
            long long *o = reinterpret_cast<long long *>(const_cast<float *>(R)) + 4 * wave;
            o[0] = clock64() - t_begin; o[1] = t_retire; o[2] = t_barrier;
            long long *tl = reinterpret_cast<long long *>(const_cast<float *>(R)) + 32 + 12 * wave;
            for (int i = 0; i < 12; ++i) tl[i] = t_line[i];
            if (wave == 0) for (int i = 0; i < 8; ++i) tl[96 + i] = t_kt[i];
        }
        return;
    }

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

#define HNR_LINEAR_ARGS d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope, d_R, d_ridx, ldr, r_cols, r_mode_x
#define HNR_LINEAR_LAUNCH_M(TM_, TN_, ACT_, DBG_, WN_, PAIR_, MASK_, THREADS_, LDS_)                                                 \
    do {                                                                                                                          \
        if (d_R) linear_f32_kernel<TM_, TN_, ACT_, DBG_, WN_, 1, PAIR_, MASK_><<<grid, THREADS_, LDS_, st>>>(HNR_LINEAR_ARGS);    \
        else linear_f32_kernel<TM_, TN_, ACT_, DBG_, WN_, 0, PAIR_, MASK_><<<grid, THREADS_, LDS_, st>>>(HNR_LINEAR_ARGS);        \
    } while (0)
#define HNR_LINEAR_LAUNCH(TM_, TN_, ACT_, DBG_, WN_, PAIR_, THREADS_, LDS_) HNR_LINEAR_LAUNCH_M(TM_, TN_, ACT_, DBG_, WN_, PAIR_, 0, THREADS_, LDS_)
// 8-wave kernel: the K % 32 != 0 variant masks the last K tile
#define HNR_LINEAR_LAUNCH8(ACT_, DBG_, PAIR_)                                                                     \
    do {                                                                                                          \
        if (K % 32) HNR_LINEAR_LAUNCH_M(2, 2, ACT_, DBG_, 4, PAIR_, 1, 512, 163840);                              \
        else HNR_LINEAR_LAUNCH_M(2, 2, ACT_, DBG_, 4, PAIR_, 0, 512, 163840);                                     \
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
    const int n_cu = device_num_cus();
    const int n_mtiles = cdiv(M, 128);
    static int dbg = -1;
    if (dbg < 0) { const char *e = getenv("HNR_LINEAR_DBG"); dbg = e ? atoi(e) : 0; }
    // column pairing (8-B epilogue accesses) needs even strides / column counts and 8-B aligned bases
    static int pair_env = -1;
    if (pair_env < 0) { const char *e = getenv("HNR_LINEAR_PAIR"); pair_env = e ? atoi(e) : 1; }
    const bool pair = pair_env && N >= 128 && !(N & 1) && !(ldc & 1) && !((uintptr_t)d_C & 7) &&
                      (!d_R || (!(ldr & 1) && !(r_cols & 1) && !((uintptr_t)d_R & 7)));
    const int r_mode_x = r_mode;
    if (N > 128 && Np % 256 == 0 && dbg != 3) {
        // 128 x 256 block tile, 8 waves (2 x 4): the A tile is fetched ONCE for all 256 output columns.  (With two 128-column
        // workgroups per M tile the PMC counters show A coming from HBM twice: 56.7 GB fetched per 24.3 GB of A.)
        const int ny = Np / 256;
        int gx = n_cu / ny;                                       // one 512-thread workgroup per CU (110 KB of LDS)
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
#ifdef HNR_LINEAR_PROBE                                            /* ablation / timing instantiations: probe builds only */
        if (dbg == 5 && pair && !d_R) {
            static long long *d_dbg = nullptr;
            if (!d_dbg && hipMalloc(&d_dbg, (32 + 8 * 12 + 8) * sizeof(long long)) != hipSuccess) return HNR_ERR_HIP;
            linear_f32_kernel<2, 2, 1, 5, 4, 0, 1><<<grid, 512, 163840, st>>>(d_A, lda, d_Wp, Kp, d_bias_p, d_C, ldc, M, N, K, slope,
                                                                              reinterpret_cast<const float *>(d_dbg), nullptr, 0, 0, 0);
            long long h[32 + 96 + 8];
            if (hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return HNR_ERR_HIP;
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "[linear dbg5] wave %d: %lld cycles (%d M tiles x %d K tiles); in the counted wait %lld, wait end -> barrier release %lld\n", w,
                        h[4 * w], (n_mtiles + gx - 1) / gx, (K + 31) / 32, h[4 * w + 1], h[4 * w + 2]);
            fprintf(stderr, "[linear dbg5] wave 0, cycles per K-tile position (avg over the M tiles):");
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", h[32 + 96 + i] / ((n_mtiles + gx - 1) / gx));
            fprintf(stderr, "\n");
            const long long t00 = h[32] < h[32 + 48] ? h[32] : h[32 + 48];
            for (int w = 0; w < 8; w += 4) {
                fprintf(stderr, "[linear dbg5] wave %d tile 100 timeline (cycles after the earlier wave's start):", w);
                for (int i = 0; i < 11; ++i) fprintf(stderr, " %lld", h[32 + 12 * w + i] - t00);
                fprintf(stderr, "\n");
            }
        }
        else if (dbg >= 16 && pair && !d_R && !(K % 32)) {
#define HNR_ABL(X_) case X_: linear_f32_kernel<2, 2, 1, 16 + X_, 4, 0, 1, 0><<<grid, 512, 163840, st>>>(HNR_LINEAR_ARGS); break;
            switch (dbg - 16) { HNR_ABL(1) HNR_ABL(2) HNR_ABL(4) HNR_ABL(8) HNR_ABL(3) HNR_ABL(6) HNR_ABL(7) HNR_ABL(15) HNR_ABL(11) default: break; }
        }
        else
#endif
        if (pair) { if (act) HNR_LINEAR_LAUNCH8(1, 0, 1); else HNR_LINEAR_LAUNCH8(0, 0, 1); }
        else if (act) HNR_LINEAR_LAUNCH8(1, 0, 0);
        else HNR_LINEAR_LAUNCH8(0, 0, 0);
    } else if (N >= 128) {
        const int ny = Np / 128;
        int gx = (2 * n_cu) / ny;
        if (gx < 1) gx = 1;
        if (gx > n_mtiles) gx = n_mtiles;
        dim3 grid(gx, ny);
        if (pair) { if (act) HNR_LINEAR_LAUNCH(2, 2, 1, 0, 2, 1, 256, 81920); else HNR_LINEAR_LAUNCH(2, 2, 0, 0, 2, 1, 256, 81920); }
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
