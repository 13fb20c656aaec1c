// Three dense layers (nn.Linear + LeakyReLU) of width <= 128 in ONE kernel, fp32 in / fp32 out, "f16x2" arithmetic (hnr_h2.h):
// the per-SAMPLE MLPs of PointAggregator.viewmlp,
//   color_feature_branch   280 -> 128 -> 128 -> 128                              models/aggregators/point_aggregators.py:1028-1037
//   aux_merge_weight_block [imgfeat45 | ddir3] (+ per-sample colour-feature addend) -> 64 -> 64 -> 64    :1199 (first three layers)
//   color_mixup_block      90 -> 45 -> 45 -> 45 (last layer without activation)  :1285-1292
// which the per-layer path runs as 3 + 3 + 3 fp32-MFMA launches with every [rows, width] activation written to and re-read from
// HBM (13.7 ms of the round-1 frame for 0.35 TFLOP).  Here a workgroup carries a tile of 128 rows through the three layers with
// the activations staged in LDS as fp16 (h, m) planes in MFMA fragment order, exactly like csrc/chain.hip: wave w owns output
// columns 32 w .. 32 w + 31 (waves whose columns are padding idle through the layer), weights stream L2 -> registers.
#include <stdlib.h>

#include "hnr_h2.h"

namespace hnr {

constexpr int ML_PAD = 32;                         // bytes between the k steps' slots: the prologue's 8-byte plane stores of one row go to 4 .. 16 k steps at once, and at a
                                                   // stride of a whole slot (a multiple of 128 B) they all hit the same banks (rocprofv3: half of the kernels' LDS cycles were conflicts)
constexpr int ML_SLOT = 8192 + ML_PAD;             // LDS bytes per k step of the activation planes: [row tile 4][plane 2][64 lanes][16 B] + pad
constexpr int ML_WSTEP = 8192;                     // weight image bytes per k step: [column tile 4][plane 2][64 lanes][16 B]
constexpr int ML_META_FLOATS = 4 * 128 + 4 + 4;    // bias[4][128], descale[4], max|W| bits[4]
constexpr int ML_DESC = 4 * 128, ML_WMAX = 4 * 128 + 4;

struct MlpArgs {
    const float *A; int lda;           // [M, lda] input rows
    const float *R; const int32_t *ridx; int ldr;      // optional addend of layer 0: R[ridx[row], 0:N0]
    const char *wimg;                  // packed weights (hnr_mlp_pack)
    int wbase[4];                      // byte offset of every layer's image
    int K0, N[4], act[4];
    float *C2; int ldc2;               // optional TAIL layer (S3 > 0): a fourth layer on layer 2's output, written to C2 [M, ldc2] (N[3] columns)
    // MODE 1 (merge stage: reprojection + image-feature gather -> merge-weight MLP -> weighted merge; rows = 4 views x samples):
    const float *loc_w; const int32_t *vs_item;         // [R,SR,3], valid-sample list
    const float *w2c, *Kmat, *campos, *campos_n;        // [4,4,4] inverse(c2w_nearest), [3,3], [3], [4,3]
    const float *fm; int H, W;                          // reference-view feature map [4,H,W,48]
    const float *frame_w;                               // optional [4]
    const float *w_last, *b_last;                       // aux_merge_weight_block.6: [64], [1]
    const float *CF; int ldcf;                          // colour feature [S,128] (its first 45 columns open the mix-up row)
    float *X7; int ld7;                                 // out [S, ld7 >= 90]: [colfeat[:45] | merged45]
    float slope;
    const unsigned long long *counts; int count_index, count_mult; long long M_cap;     // M = min(M_cap, counts[index] * mult) (counts may be NULL)
    int seg_stride;                    // > 0: the rows are count_mult segments of counts[index] rows each, segment v starting at row v * seg_stride
    float *C; int ldc;                 // [M, ldc] output rows (N[2] columns)
    // training forward: the outputs of layers 0 and 1 are kept too (the backward pass needs them), and the three layers' output maxima
    float *T0; int ldt0; float *T1; int ldt1;
    unsigned *tmax;                    // [3] bit patterns (atomicMax), may be NULL
};

typedef float f32x4m __attribute__((ext_vector_type(4)));

#ifdef HNR_MLP_PROBE
__device__ long long g_mlp_probe[24];
#define MLP_STAMP(i_) do { const long long t_ = clock64(); tm_[i_] += t_ - tp_; tp_ = t_; if ((i_) == 0) ++ntile_; } while (0)
#else
#define MLP_STAMP(i_) do { } while (0)
#endif
// RT = row tiles of 32 rows per workgroup tile.  RT = 4 (128 rows): one workgroup per CU when layer 0 is wide (280 inputs = 144 KiB of operand
// planes); RT = 2 (64 rows): two workgroups per CU, so that one's prologue (row loads, scaling, fp16 split: latency and VALU) runs under the
// other's MFMAs -- the phases of a single workgroup are serial.
// workgroups per CU a variant is built for (registers) and launched with (the LDS planes of layer 0 allow them): the phases of a tile are
// serial (load + split, MFMA, epilogue, barriers), so co-resident workgroups are what keeps the CU busy
constexpr int mlp3_wgs_per_cu(int S0, int RT, int MODE) { return MODE == 1 ? (RT >= 4 ? 2 : 4) : RT >= 4 ? 1 : (RT == 1 || S0 <= 6) ? 4 : 2; }

// rendezvous that waits for this wave's LDS traffic only (a __syncthreads() also waits for every global load in flight)
#ifdef HNR_MLP_FULL_BARRIER
#define MLP_LDS_BARRIER() __syncthreads()
#else
#define MLP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#ifndef HNR_MLP_LPR16
#define HNR_MLP_LPR16 1
#endif
template <int S0, int S1, int S2, int S3, int MODE, int RT = 4>
__global__ __launch_bounds__(256, mlp3_wgs_per_cu(S0, RT, MODE)) void mlp3_kernel(MlpArgs a)
{
    static_assert(MODE != 1 || RT == 4 || RT == 2, "the merge stage is built on 128- or 64-row tiles (32 / 16 samples x 4 views)");
    constexpr int SLOT = RT * 2048 + ML_PAD, ROWS = 32 * RT;               // LDS bytes per k step of the planes: [row tile RT][plane 2][64 lanes][16 B] + pad
    constexpr int SMAX3 = S0 > S1 ? (S0 > S2 ? S0 : S2) : (S1 > S2 ? S1 : S2), SMAX = SMAX3 > S3 ? SMAX3 : S3;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    long long M = a.M_cap;
    if (a.counts) { const long long c = (long long)a.counts[a.count_index] * a.count_mult; if (c < M) M = c; }
    // logical row m -> physical row of A / ridx / C
    long long n_unit = 1;
    if (a.seg_stride > 0) {
        n_unit = a.counts ? (long long)a.counts[a.count_index] : a.M_cap / (a.count_mult > 0 ? a.count_mult : 1);
        if (n_unit > a.seg_stride) n_unit = a.seg_stride;
        if (n_unit < 1) n_unit = 1;
        if (M > n_unit * a.count_mult) M = n_unit * a.count_mult;
    }
    // (no integer division: at most 7 compares -- a 64-bit divide per lane and use cost more than the tile's MFMAs)
    auto phys = [&](long long m) -> long long {
        if (a.seg_stride <= 0) return m;
        int q = 0;
        for (int v = 1; v < a.count_mult && v < 8; ++v) q += (m >= (long long)v * n_unit) ? 1 : 0;
        return (long long)q * a.seg_stride + (m - (long long)q * n_unit);
    };
    const int n_tiles = (int)((M + ROWS - 1) / ROWS);
    const int total_steps = S0 + S1 + S2 + S3;
    const float *meta = reinterpret_cast<const float *>(a.wimg + (size_t)total_steps * ML_WSTEP);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, total_steps * ML_WSTEP, 0x00020000);
    float *exch = reinterpret_cast<float *>(lds + SMAX * SLOT);            // [row ROWS][wave 4]
    float *rowinv = exch + ROWS * 4;                                       // [row ROWS]: 2^-k of the input row's scale
    // MODE 1 extras: fp32 rows [128][48] = [imgfeat45 | ddir3] of the tile's (sample, view) rows, per-row pixel offset / validity / merge weight
    float *s_f = rowinv + ROWS;
    int *s_pix = reinterpret_cast<int *>(s_f + ROWS * 48);
    float *s_vm = reinterpret_cast<float *>(s_pix + ROWS), *s_w = s_vm + ROWS;
    float *s_wl = s_w + ROWS;                                              // MODE 1: the last merge-weight layer's 64 weights (+ zero padding to 128)
    // Which (row tile, column tile) pairs a wave multiplies.  Wide layers (N = 128): wave w = column tile w of all RT row tiles.  When every layer is
    // at most 64 wide (merge weights 48 -> 64 -> 64 -> 64, mix-up 90 -> 45 -> 45 -> 45) that left waves 2 and 3 idle through every MFMA loop and
    // epilogue: there the rows are split too (RS) -- wave w = column tile w & 1 of the row tiles (w >> 1) RTW .. + RTW - 1.  Same arithmetic per element.
    constexpr bool RS = S1 <= 4 && S2 <= 4 && S3 == 0 && RT >= 2;
    constexpr int RTW = RS ? RT / 2 : RT;                                  // row tiles per wave
    const int cw = RS ? (wave & 1) : wave, rt0 = RS ? (wave >> 1) * RTW : 0;
    const int col0 = 32 * cw + 16 * h;                                     // this lane's columns: col0 + r
    const unsigned woff = (unsigned)cw * 2048u + (unsigned)lane * 16u;
    const char *lds_b = lds + rt0 * 2048;                                  // the wave's first row tile in every k step's slot
    const f32x2 slope2 = {a.slope, a.slope};

    if (MODE == 1) {                                                       // (the first tile's prologue barriers order this before its first use)
        for (int i = tid; i < 128; i += 256) s_wl[i] = i < a.N[2] ? a.w_last[i] : 0.f;
    }
    float tmax_run[3] = {0.f, 0.f, 0.f};
#ifdef HNR_MLP_PROBE
    long long tm_[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = clock64(), ntile_ = 0;
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row_base = (long long)tile * ROWS;
        int tid_t = tid;                                                   // laundered per tile: the per-thread index arithmetic below (idx / 45 ...) is cheap, but hoisted out of the
        asm volatile("" : "+v"(tid_t));                                    // tile loop it becomes a dozen 64-bit addresses per thread that the register allocator spills to scratch
        MLP_STAMP(0);
        float inv[RTW];
        if (MODE == 1) {
            // ---- merge-stage prologue.  Row t of the tile = (sample ls = t >> 2, view v = t & 3): a sample's four views sit in adjacent rows.
            // (a) reprojection into the view (w2iproject, neural_points_volumetric_model.py:248-255), truncation to a pixel + bounds rule
            //     (point_aggregators.py:1077-1088), delta view direction (:296-310) -- the arithmetic of proj_rows_kernel;
            //     threads 0..127 take the projection of row tid, threads 128..255 the direction deltas of row tid - 128 (waves 2, 3 used to idle here)
            if (tid < 2 * ROWS) {
                const int row_t = tid % ROWS, ls = row_t >> 2, v = row_t & 3;
                long long sidx = row_base / 4 + ls;
                if (sidx * 4 >= M) sidx = M / 4 - 1;
                const float *pw = a.loc_w + (size_t)a.vs_item[sidx] * 3;
                const float x = pw[0], y = pw[1], z = pw[2];
              if (tid < ROWS) {
                int px, py;
                const bool inval = hnr_project_pixel(x, y, z, a.w2c + 16 * v, a.Kmat, a.W, a.H, px, py);
                s_pix[tid] = ((v * a.H + py) * a.W + px) * 48;
                s_vm[tid] = inval ? 0.f : 1.f;
              } else {
                const float cx = x - a.campos[0], cy = y - a.campos[1], cz = z - a.campos[2];
                const float cn = sqrtf(cx * cx + cy * cy + cz * cz) + 1e-6f;
                const float nx = x - a.campos_n[3 * v], ny = y - a.campos_n[3 * v + 1], nz = z - a.campos_n[3 * v + 2];
                const float nn = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f;
                s_f[row_t * 48 + 45] = hnr_div(nx, nn) - hnr_div(cx, cn); s_f[row_t * 48 + 46] = hnr_div(ny, nn) - hnr_div(cy, cn); s_f[row_t * 48 + 47] = hnr_div(nz, nn) - hnr_div(cz, cn);
              }
            }
            MLP_STAMP(13);
            __syncthreads();
            // (b) the 45 feature channels of the pixel (192-B contiguous per row; channels 45..47 of the map are padding);
            {
                constexpr int NF = ROWS * 12 / 256;                          // scattered 16-B loads per thread, all in flight together
                float4 f4[NF];
#pragma unroll
                for (int it = 0; it < NF; ++it) {
                    const int idx = tid_t + 256 * it, row = idx / 12, q = idx - row * 12;
                    f4[it] = *reinterpret_cast<const float4 *>(a.fm + (size_t)s_pix[row] + 4 * q);
                }
#pragma unroll
                for (int it = 0; it < NF; ++it) {
                    const int idx = tid_t + 256 * it, row = idx / 12, q = idx - row * 12;
                    float *d = s_f + row * 48 + 4 * q;
                    if (q < 11) *reinterpret_cast<float4 *>(d) = f4[it]; else d[0] = f4[it].x;      // column 44; 45..47 hold the direction deltas
                }
            }
            __syncthreads();
            // (c) layer-0 operand planes of the rows 32 wave + j
            if (wave < RT) {
                const float *src = s_f + (32 * wave + j) * 48 + 8 * h;
                float x[3][8];
                float m = 0.f;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(src + 16 * s), v1 = *reinterpret_cast<const float4 *>(src + 16 * s + 4);
                    const float t[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) { x[s][e] = t[e]; m = fmaxf(m, fabsf(t[e])); }
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                const int k = row_scale_exp(m);
                const float sc = pow2f(k);
                if (h == 0) rowinv[32 * wave + j] = pow2f(-k);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    unsigned ph[4], pm[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split2h(__fmul_rn(x[s][2 * q], sc), __fmul_rn(x[s][2 * q + 1], sc), ph[q], pm[q]);
                    char *dst = lds + s * SLOT + (wave * 2) * 1024 + lane * 16;
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                }
            }
        } else
        // ---- prologue: wave w converts rows 8 RT w .. 8 RT (w + 1) - 1.  LPR lanes share a row and read it as ONE contiguous burst (16 B per lane; a
        // second burst for columns >= 4 LPR): a lane-per-row layout touched 32 cache lines with every load instruction and ran at 5 B/clk.
        // Row maximum by a butterfly over the LPR lanes, power-of-two scale, fp16 split, and each lane drops its 4 columns (8 B per plane)
        // into the fragment slot (k step = col >> 4, lane half = (col >> 3) & 1, element = col & 7) of the layer-0 operand planes.
        {
            // lanes per row: the fewest float4 slots over the row's columns -- 288 columns on 64 lanes x 2 bursts left 44 % of the lanes converting
            // zeros (the second burst holds 32 real columns); on 16 lanes x 5 bursts 10 %
            constexpr int COLS = 16 * S0, LPR = HNR_MLP_LPR16 && COLS == 288 ? 16 : (COLS > 128 ? 64 : (COLS > 64 ? 32 : 16)), RPI = 64 / LPR, NB = (COLS + 4 * LPR - 1) / (4 * LPR);
            const int lr = lane % LPR, sub = lane / LPR;                   // lane within the row, row within the instruction
            constexpr int RW = 8 * RT;                                     // rows of this wave
            constexpr int ROWS_B = ((RW / RPI) * NB * 4 <= (RT < 4 ? 128 : 256)) ? RW : RW / 2;   // rows in flight per batch (registers: BATCH x NB float4)
            constexpr int BATCH = ROWS_B / RPI;
#pragma unroll 1
            for (int r0 = 0; r0 < RW; r0 += ROWS_B) {
                float4 v[BATCH][NB];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int rl = RW * wave + r0 + b * RPI + sub;         // row of the tile
                    long long row = row_base + rl;
                    if (row >= M) row = M - 1;
                    const float *src = a.A + (size_t)phys(row) * a.lda;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int c = 4 * (nb * LPR + lr);
                        v[b][nb] = *reinterpret_cast<const float4 *>(src + (c + 4 <= a.lda ? c : 0));      // past the row's end: re-read its start (masked below)
                    }
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int rl = RW * wave + r0 + b * RPI + sub;
                    float m = 0.f;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int c = 4 * (nb * LPR + lr);
                        float *t = reinterpret_cast<float *>(&v[b][nb]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t[e] = (c + e < a.K0) ? t[e] : 0.f; m = fmaxf(m, fabsf(t[e])); }
                    }
                    // maximum over the row's LPR lanes by DPP (no LDS round trips): quads, half rows, rows of 16, then row broadcasts
                    m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
                    m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
                    m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x141, 0xf, 0xf, false));     // row_half_mirror
                    m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x140, 0xf, 0xf, false));     // row_mirror: every lane of a 16-lane row holds its maximum
                    if (LPR >= 32) m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x142, 0xa, 0xf, false));     // row_bcast15 -> rows 1, 3
                    if (LPR >= 64) m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x143, 0xc, 0xf, false));     // row_bcast31 -> rows 2, 3
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
        }
        MLP_STAMP(1);
        __syncthreads();
        MLP_STAMP(2);
        {
            const float dw0 = meta[ML_DESC + 0];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) inv[rt] = __fmul_rn(rowinv[32 * (rt0 + rt) + j], dw0);
        }

        f32x16 acc[RTW][1];
        auto zero_acc = [&]() {
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][0][r] = 0.f;
        };
        // the gathered first-layer addend rows of all row tiles: asked for BEFORE the layer's MFMA loop, used in its epilogue
        float4 ad[2][4];                                                   // ring of two row tiles: two rows' loads in flight while one is used
        auto load_addend = [&](int rt) {
            long long row = row_base + 32 * (rt0 + rt) + j;
            if (row >= M) row = M - 1;
            const float *rrow = a.R + (size_t)(MODE == 1 ? row / 4 : (long long)a.ridx[phys(row)]) * a.ldr + col0;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) ad[rt & 1][q4] = *reinterpret_cast<const float4 *>(rrow + 4 * q4);
        };
        // v = acc * inv + bias (+ addend) (+ LeakyReLU); returns the per-row maxima of this wave's columns
        auto activate = [&](int layer, float (&amax)[RTW], auto with_addend) {
            constexpr bool ADD = decltype(with_addend)::value;
            f32x2 bias[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4 *>(meta + layer * 128 + col0 + 4 * q);
                bias[2 * q] = f32x2{b.x, b.y}; bias[2 * q + 1] = f32x2{b.z, b.w};
            }
            const bool act = a.act[layer] != 0;
            if (ADD && MODE != 1) { load_addend(0); if (RTW > 1) load_addend(1); }         // (the merge stage asks for them under its layer-0 MFMAs)
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                float m = 0.f;
                const f32x2 inv2 = {inv[rt], inv[rt]};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f32x2 add = bias[q];
                    if (ADD) { const float4 t4 = ad[rt & 1][q >> 1]; add = add + ((q & 1) ? f32x2{t4.z, t4.w} : f32x2{t4.x, t4.y}); }
                    f32x2 v = f32x2_fma(f32x2{acc[rt][0][2 * q], acc[rt][0][2 * q + 1]}, inv2, add);
                    if (act) { const f32x2 sv = v * slope2; v.x = fmaxf(v.x, sv.x); v.y = fmaxf(v.y, sv.y); }
                    acc[rt][0][2 * q] = v.x; acc[rt][0][2 * q + 1] = v.y;
                    m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
                }
                amax[rt] = m;
                if (ADD && rt + 2 < RTW) load_addend(rt + 2);
            }
        };
        auto publish = [&](int next_layer, bool active, float (&amax)[RTW]) {
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                const float m = active ? fmaxf(amax[rt], __shfl_xor(amax[rt], 32)) : 0.f;
                if (h == 0) exch[(32 * (rt0 + rt) + j) * 4 + cw] = m;
            }
            __syncthreads();                                               // every wave has finished reading the previous planes
            const float dw = meta[ML_DESC + next_layer];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                float4 m4 = *reinterpret_cast<const float4 *>(exch + (32 * (rt0 + rt) + j) * 4);
                if (RS) { m4.z = 0.f; m4.w = 0.f; }                        // two column tiles only (what the idle waves 2, 3 used to contribute: 0)
                const int k = row_scale_exp(fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w)));
                const float sc = pow2f(k);
                const f32x2 sc2 = {sc, sc};
                inv[rt] = __fmul_rn(pow2f(-k), dw);
                if (active) {
                    unsigned ph[8], pm[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x2 vs = f32x2{acc[rt][0][2 * q], acc[rt][0][2 * q + 1]} * sc2;
                        split2h(vs.x, vs.y, ph[q], pm[q]);
                    }
                    char *dst = lds + (2 * cw + h) * SLOT + ((rt0 + rt) * 2) * 1024 + j * 16;
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<u32x4 *>(dst + 512) = u32x4{ph[4], ph[5], ph[6], ph[7]};
                    *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                    *reinterpret_cast<u32x4 *>(dst + 1024 + 512) = u32x4{pm[4], pm[5], pm[6], pm[7]};
                }
            }
            __syncthreads();
        };
        auto store = [&](float *C, int ldc) {
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                const long long row = row_base + 32 * (rt0 + rt) + j;
                if (row < M) {
                    float *o = C + (size_t)phys(row) * ldc + col0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = col0 + 4 * q;
                        if (c + 4 <= ldc)
                            *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(acc[rt][0][4 * q], acc[rt][0][4 * q + 1], acc[rt][0][4 * q + 2], acc[rt][0][4 * q + 3]);
                    }
                }
            }
        };
        const bool act0 = 32 * cw < a.N[0], act1 = 32 * cw < a.N[1], act2 = 32 * cw < a.N[2];
        float amax[RTW];
        // ---- layer 0
        zero_acc();
        if (act0) {
            h2_mfma_layer<RTW, 1, S0, 0, ML_WSTEP, SLOT>(wsrd, a.wbase[0], woff, lds_b, lane, acc, []() {}, [&]() { if (MODE == 1) { load_addend(0); load_addend(1); } });
            MLP_STAMP(3);
            if (a.R) activate(0, amax, std::true_type{}); else activate(0, amax, std::false_type{});
            MLP_STAMP(4);
            if (MODE == 0 && a.T0) {
                store(a.T0, a.ldt0);
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) tmax_run[0] = fmaxf(tmax_run[0], amax[rt]);
            }
        }
        publish(1, act0, amax);
        MLP_STAMP(5);
        // ---- layer 1
        zero_acc();
        if (act1) {
            h2_mfma_layer<RTW, 1, S1, 0, ML_WSTEP, SLOT>(wsrd, a.wbase[1], woff, lds_b, lane, acc, []() {});
            MLP_STAMP(6);
            activate(1, amax, std::false_type{});
            if (MODE == 0 && a.T1) {
                store(a.T1, a.ldt1);
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) tmax_run[1] = fmaxf(tmax_run[1], amax[rt]);
            }
        }
        publish(2, act1, amax);
        MLP_STAMP(7);
        // ---- layer 2 -> fp32 rows
        float cfv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 wl4[4];                                                     // MODE 1: this lane's 16 weights of aux_merge_weight_block's last layer
        zero_acc();
        if (act2) {
            h2_mfma_layer<RTW, 1, S2, 0, ML_WSTEP, SLOT>(wsrd, a.wbase[2], woff, lds_b, lane, acc, []() {});
            MLP_STAMP(8);
            activate(2, amax, std::false_type{});
            MLP_STAMP(9);
            if (MODE != 1) store(a.C, a.ldc);
            if (MODE == 0 && a.tmax) {
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) tmax_run[2] = fmaxf(tmax_run[2], amax[rt]);
            }
            MLP_STAMP(10);
        }
        if (MODE == 1) {
            // the last layer's weights and the colour-feature columns that open the mix-up rows: asked for here, after the tile's last weight load (a wait for a weight
            // fragment also waits for every older load), used after the two barriers of the merge-weight epilogue
#pragma unroll
            for (int it = 0; it < (8 * RT * 45 + 255) / 256; ++it) {
                const int idx = tid_t + 256 * it, ls = idx / 45, ch = idx - ls * 45;
                const long long sidx = row_base / 4 + ls;
                cfv[it] = (idx < 8 * RT * 45 && sidx * 4 < M) ? a.CF[(size_t)sidx * a.ldcf + ch] : 0.f;
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) wl4[q4] = *reinterpret_cast<const float4 *>(s_wl + col0 + 4 * q4);
        }
        if (MODE == 1) {
            // ---- last layer of aux_merge_weight_block (64 -> 1) + sigmoid, validity / frame weights (:1199), weighted merge over the 4 views
            // (:1217) and the mix-up row (:1286-1292)
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) {
                float d = 0.f;
                if (act2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) d = fmaf(acc[rt][0][r], reinterpret_cast<const float *>(wl4)[r], d);
                    d = __fadd_rn(d, __shfl_xor(d, 32));
                }
                if (h == 0) exch[(32 * (rt0 + rt) + j) * 4 + cw] = d;
            }
            MLP_LDS_BARRIER();                                             // LDS traffic only: the colour-feature loads asked for above stay in flight across it
            MLP_STAMP(14);
            if (tid < ROWS) {
                float4 d4 = *reinterpret_cast<const float4 *>(exch + tid * 4);
                if (RS) { d4.z = 0.f; d4.w = 0.f; }
                const float d = __fadd_rn(__fadd_rn(d4.x, d4.y), __fadd_rn(d4.z, d4.w));
                float wv = hnr_div(1.f, 1.f + expf(-(d + a.b_last[0])));
                wv *= s_vm[tid];
                if (a.frame_w) wv *= a.frame_w[tid & 3];
                s_w[tid] = wv;
            }
            MLP_LDS_BARRIER();
            MLP_STAMP(15);
            {
#pragma unroll
                for (int it = 0; it < (8 * RT * 45 + 255) / 256; ++it) {
                    const int idx = tid_t + 256 * it, ls = idx / 45, ch = idx - ls * 45;
                    const long long sidx = row_base / 4 + ls;
                    if (idx < 8 * RT * 45 && sidx * 4 < M) {
                        float fsum = 0.f, wsum = 0.f;
#pragma unroll
                        for (int v = 0; v < 4; ++v) { const float wv = s_w[4 * ls + v]; fsum += s_f[(4 * ls + v) * 48 + ch] * wv; wsum += wv; }
                        float *o = a.X7 + (size_t)sidx * a.ld7;
                        __builtin_nontemporal_store(cfv[it], o + ch);
                        __builtin_nontemporal_store(hnr_div(fsum, wsum + 1e-6f), o + 45 + ch);
                    }
                }
                MLP_STAMP(17);
            }
        }
        if (S3 > 0) {
            // ---- tail layer on layer 2's output -> second fp32 output
            publish(3, act2, amax);
            zero_acc();
            if (32 * wave < a.N[3]) {
                h2_mfma_layer<RTW, 1, (S3 > 0 ? S3 : 1), 0, ML_WSTEP, SLOT>(wsrd, a.wbase[3], woff, lds_b, lane, acc, []() {});
                activate(3, amax, std::false_type{});
                store(a.C2, a.ldc2);
            }
        }
        MLP_STAMP(11);
        __syncthreads();                                                   // the planes and rowinv are rewritten by the next tile's prologue
        MLP_STAMP(12);
    }
    if (MODE == 0 && a.tmax) {
        // one atomic per workgroup and layer
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            float m = tmax_run[l];
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) exch[4 * l + wave] = m;
        }
        __syncthreads();
        if (tid < 3) { const float m = fmaxf(fmaxf(exch[4 * tid], exch[4 * tid + 1]), fmaxf(exch[4 * tid + 2], exch[4 * tid + 3])); if (m > 0.f) atomicMax(a.tmax + tid, __float_as_uint(m)); }
    }
#ifdef HNR_MLP_PROBE
    if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i = 0; i < 20; ++i) g_mlp_probe[i] = tm_[i]; g_mlp_probe[20] = ntile_; }
#endif
}

// ---- merge stage, one WAVE per 32-row tile (8 samples x 4 views) --------------------------------------------------------------------------
// The arithmetic of mlp3_kernel<3, 4, 4, 0, 1, RT> element for element (same products, same accumulation order, same scales), laid out so
// that a wave never waits for another: it owns both 32-column tiles of its 32 rows, so the row maxima, the operand planes of the next layer,
// the 64 -> 1 last layer and the merge over the four views all stay inside the wave -- no workgroup barrier anywhere in the tile loop.
// What bounded the workgroup-per-tile form (and a first wave-per-tile form that still streamed its weights from L2) was neither HBM nor the
// matrix pipe but the CU's vector-memory front end: 115 load instructions per 32 rows, 44 KiB of them weight fragments, kept the texture
// addresser 95 % busy (rocprofv3 TA_TA_BUSY) while VALU sat at 42 % and MFMA at 17 %.  Here the weights (44 KiB: two column tiles of the
// 11 k steps), biases and camera matrices live in LDS for the whole kernel, the fp32 feature rows are not staged at all (layer-0 operand planes
// straight from the gathered registers, row maxima through 32 LDS atomics), the merge re-reads the (L2-resident) feature rows with one lane per
// (sample, 4 channels) so that its sum over the views needs no exchange, and the mix-up rows leave through LDS as whole 16-B chunks:
// 28 loads and 3 stores per tile.  12 waves per CU (one workgroup): a tile is a serial chain of ~1250 VALU instructions and 66 MFMAs, and it
// is the other waves of the SIMD that fill its gaps.
constexpr int MW_SLOT = 2048 + ML_PAD;                                     // one k step of a wave's operand planes: [plane 2][64 lanes][16 B] + pad
constexpr int MW_WAVE_LDS = 4 * MW_SLOT + 4 * 32 * 4 + 32 * 3 * 4;         // planes; pixel offset / validity / merge weight / row maximum per row; direction deltas [32][3]
constexpr int MW_WAVES = 12;
constexpr int MW_WBYTES = 11 * 4096;                                       // weights [k step 11][column tile 2][plane 2][64 lanes][16 B]
constexpr int MW_SHARED = MW_WBYTES + 3 * 64 * 4 + 64 * 4 + 128 * 4;       // + bias [3][64], last layer [64], w2c [64] | K [12] | campos [4] | campos_nearest [12]
constexpr int MW_LDS = MW_SHARED + MW_WAVES * MW_WAVE_LDS;
#define MW_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")  /* LDS traffic of ONE wave is in order: this only pins the compiler */
template <int S>
__device__ __forceinline__ void mw_layer(const char *wp, const char *bp, f32x16 (&acc)[2])
{
    // fragment (s, c, p) of the layer at wp + s * 4096 + (c * 2 + p) * 1024, operand planes (s, p) at bp + s * MW_SLOT + p * 1024 (both + lane * 16)
    u32x4 wf[2][2][2], bf[2][2];
    auto ld = [&](int slot, int s) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) wf[slot][c][q] = *reinterpret_cast<const u32x4 *>(wp + s * 4096 + (c * 2 + q) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q) bf[slot][q] = *reinterpret_cast<const u32x4 *>(bp + s * MW_SLOT + q * 1024);
    };
    ld(0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) ld((s + 1) & 1, s + 1);
        const int b = s & 1;
#define MW_W(c, q) __builtin_bit_cast(f16x8, wf[b][c][q])
#define MW_X(q) __builtin_bit_cast(f16x8, bf[b][q])
        // smallest terms first (wm*xh, wh*xm, wh*xh), the order of h2_mfma_layer
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MW_W(c, 1), MW_X(0), acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MW_W(c, 0), MW_X(1), acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MW_W(c, 0), MW_X(0), acc[c], 0, 0, 0);
#undef MW_W
#undef MW_X
    }
}

__global__ __launch_bounds__(64 * MW_WAVES, 1) void merge_wp_kernel(MlpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long M = a.M_cap;
    if (a.counts) { const long long c = (long long)a.counts[a.count_index] * a.count_mult; if (c < M) M = c; }
    const long long n_samp = M / 4;
    const int n_tiles = (int)((n_samp + 7) / 8);
    const float *meta = reinterpret_cast<const float *>(a.wimg + (size_t)11 * ML_WSTEP);
    float *s_bias = reinterpret_cast<float *>(lds + MW_WBYTES);            // [3][64]
    float *s_wl = s_bias + 3 * 64;                                         // the last layer's 64 weights
    float *s_cam = s_wl + 64;                                              // w2c [4][16] | K [9] (+3) | campos [3] (+1) | campos_nearest [4][3]
    char *wl = lds + MW_SHARED + wave * MW_WAVE_LDS;
    int *s_pix = reinterpret_cast<int *>(wl + 4 * MW_SLOT);
    float *s_vm = reinterpret_cast<float *>(s_pix + 32), *s_w = s_vm + 32;
    unsigned *s_max = reinterpret_cast<unsigned *>(s_w + 32);
    float *s_dd = reinterpret_cast<float *>(s_max + 32);
    for (int i = tid; i < MW_WBYTES / 16; i += 64 * MW_WAVES) {            // chunk i = (s, c, p, lane)
        const int s = i >> 8, cp = (i >> 6) & 3, ln = i & 63;
        *reinterpret_cast<u32x4 *>(lds + i * 16) = *reinterpret_cast<const u32x4 *>(a.wimg + (size_t)s * ML_WSTEP + cp * 1024 + ln * 16);
    }
    for (int i = tid; i < 3 * 64; i += 64 * MW_WAVES) s_bias[i] = meta[(i >> 6) * 128 + (i & 63)];
    for (int i = tid; i < 64; i += 64 * MW_WAVES) { s_wl[i] = a.w_last[i]; s_cam[i] = a.w2c[i]; }
    if (tid < 9) s_cam[64 + tid] = a.Kmat[tid];
    if (tid < 3) s_cam[76 + tid] = a.campos[tid];
    if (tid < 12) s_cam[80 + tid] = a.campos_n[tid];
    __syncthreads();
    const float dw0 = meta[ML_DESC + 0], dw1 = meta[ML_DESC + 1], dw2 = meta[ML_DESC + 2], b_last = a.b_last[0];
    const f32x2 slope2 = {a.slope, a.slope};
    for (int tile = blockIdx.x * MW_WAVES + wave; tile < n_tiles; tile += gridDim.x * MW_WAVES) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                                     // per-tile opaque (see mlp3_kernel)
        const int h = lane >> 5, j = lane & 31;
        const long long sbase = (long long)tile * 8;
        // (a) lanes 0..31: reprojection of row lane; lanes 32..63: direction deltas of row lane - 32 (arithmetic of mlp3_kernel MODE 1 / proj_rows_kernel)
        {
            const int ls = j >> 2, v = j & 3;
            long long sidx = sbase + ls;
            if (sidx >= n_samp) sidx = n_samp - 1;
            const float *pw = a.loc_w + (size_t)a.vs_item[sidx] * 3;
            const float x = pw[0], y = pw[1], z = pw[2];
            if (h == 0) {
                int px, py;
                const bool inval = hnr_project_pixel(x, y, z, s_cam + 16 * v, s_cam + 64, a.W, a.H, px, py);
                s_pix[j] = ((v * a.H + py) * a.W + px) * 48;
                s_vm[j] = inval ? 0.f : 1.f;
                s_max[j] = 0u;
            } else {
                const float *cp = s_cam + 76, *cnp = s_cam + 80 + 3 * v;
                const float cx = x - cp[0], cy = y - cp[1], cz = z - cp[2];
                const float cn = sqrtf(cx * cx + cy * cy + cz * cz) + 1e-6f;
                const float nx = x - cnp[0], ny = y - cnp[1], nz = z - cnp[2];
                const float nn = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f;
                s_dd[j * 3] = hnr_div(nx, nn) - hnr_div(cx, cn); s_dd[j * 3 + 1] = hnr_div(ny, nn) - hnr_div(cy, cn); s_dd[j * 3 + 2] = hnr_div(nz, nn) - hnr_div(cz, cn);
            }
        }
        MW_WAVE_FENCE();
        // (b) the rows [imgfeat45 | ddir3] of the 32 (sample, view) pairs as 32 x 12 chunks of 4 columns, 6 per lane: the feature chunk of the
        //     row's pixel (scattered 16-B loads, all in flight together), the row maxima through LDS, then the layer-0 operand planes
        float inv;
        {
            float4 f4[6];
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = lane + 64 * it, row = idx / 12, q = idx - row * 12;
                f4[it] = *reinterpret_cast<const float4 *>(a.fm + (size_t)s_pix[row] + 4 * q);
            }
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = lane + 64 * it, row = idx / 12, q = idx - row * 12;
                if (q == 11) { f4[it].y = s_dd[row * 3]; f4[it].z = s_dd[row * 3 + 1]; f4[it].w = s_dd[row * 3 + 2]; }   // columns 45..47: the direction deltas
                const float m = fmaxf(fmaxf(fabsf(f4[it].x), fabsf(f4[it].y)), fmaxf(fabsf(f4[it].z), fabsf(f4[it].w)));
                atomicMax(s_max + row, __float_as_uint(m));
            }
            MW_WAVE_FENCE();
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = lane + 64 * it, row = idx / 12, q = idx - row * 12, c = 4 * q;
                const float sc = pow2f(row_scale_exp(__uint_as_float(s_max[row])));
                unsigned ph0, pm0, ph1, pm1;
                split2h(__fmul_rn(f4[it].x, sc), __fmul_rn(f4[it].y, sc), ph0, pm0);
                split2h(__fmul_rn(f4[it].z, sc), __fmul_rn(f4[it].w, sc), ph1, pm1);
                char *dst = wl + (c >> 4) * MW_SLOT + ((((c >> 3) & 1) * 32 + row) * 16) + (c & 7) * 2;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
                *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(pm0, pm1);
            }
            inv = __fmul_rn(pow2f(-row_scale_exp(__uint_as_float(s_max[j]))), dw0);
        }
        // the sample's colour-feature addend of layer 0 (this lane's 2 x 16 columns): asked for here, used in the layer's epilogue
        float4 ad[2][4];
        {
            long long sidx = sbase + (j >> 2);
            if (sidx >= n_samp) sidx = n_samp - 1;
            const float *rrow = a.R + (size_t)sidx * a.ldr + 16 * h;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) ad[c][q4] = *reinterpret_cast<const float4 *>(rrow + 32 * c + 4 * q4);
        }
        MW_WAVE_FENCE();
        f32x16 acc[2];
        auto zero_acc = [&]() {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        };
        // v = acc * inv + bias (+ addend), LeakyReLU; returns the row maximum over this lane's 32 columns
        auto activate = [&](int layer, auto with_addend) -> float {
            constexpr bool ADD = decltype(with_addend)::value;
            float m = 0.f;
            const f32x2 inv2 = {inv, inv};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 b = *reinterpret_cast<const float4 *>(s_bias + layer * 64 + 32 * c + 16 * h + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * q4 + e;
                        f32x2 add = e ? f32x2{b.z, b.w} : f32x2{b.x, b.y};
                        if (ADD) { const float4 t4 = ad[c][q4]; add = add + (e ? f32x2{t4.z, t4.w} : f32x2{t4.x, t4.y}); }
                        f32x2 v = f32x2_fma(f32x2{acc[c][2 * q], acc[c][2 * q + 1]}, inv2, add);
                        const f32x2 sv = v * slope2;
                        v.x = fmaxf(v.x, sv.x); v.y = fmaxf(v.y, sv.y);
                        acc[c][2 * q] = v.x; acc[c][2 * q + 1] = v.y;
                        m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
                    }
                }
            }
            return m;
        };
        auto publish = [&](float dw, float m) {
            m = fmaxf(m, __shfl_xor(m, 32));
            const int k = row_scale_exp(m);
            const float sc = pow2f(k);
            const f32x2 sc2 = {sc, sc};
            inv = __fmul_rn(pow2f(-k), dw);
            MW_WAVE_FENCE();                                               // (this wave's reads of the previous planes have returned)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned ph[8], pm[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x2 vs = f32x2{acc[c][2 * q], acc[c][2 * q + 1]} * sc2;
                    split2h(vs.x, vs.y, ph[q], pm[q]);
                }
                char *dst = wl + (2 * c + h) * MW_SLOT + j * 16;
                *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                *reinterpret_cast<u32x4 *>(dst + 512) = u32x4{ph[4], ph[5], ph[6], ph[7]};
                *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                *reinterpret_cast<u32x4 *>(dst + 1024 + 512) = u32x4{pm[4], pm[5], pm[6], pm[7]};
            }
            MW_WAVE_FENCE();
        };
        const char *wp = lds + lane * 16, *bp = wl + lane * 16;
        zero_acc();
        mw_layer<3>(wp, bp, acc);
        publish(dw1, activate(0, std::true_type{}));
        zero_acc();
        mw_layer<4>(wp + 3 * 4096, bp, acc);
        publish(dw2, activate(1, std::false_type{}));
        zero_acc();
        mw_layer<4>(wp + 7 * 4096, bp, acc);
        (void)activate(2, std::false_type{});
        // ---- last layer (64 -> 1) + sigmoid, validity / frame weights
        {
            float d2[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float d = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(s_wl + 32 * c + 16 * h + 4 * q4);
                    d = fmaf(acc[c][4 * q4], w4.x, d); d = fmaf(acc[c][4 * q4 + 1], w4.y, d);
                    d = fmaf(acc[c][4 * q4 + 2], w4.z, d); d = fmaf(acc[c][4 * q4 + 3], w4.w, d);
                }
                d2[c] = __fadd_rn(d, __shfl_xor(d, 32));
            }
            const float d = __fadd_rn(d2[0], d2[1]);
            float wv = hnr_div(1.f, 1.f + expf(-(d + b_last)));
            wv *= s_vm[j];
            if (a.frame_w) wv *= a.frame_w[j & 3];
            if (h == 0) s_w[j] = wv;
        }
        MW_WAVE_FENCE();
        // ---- weighted merge over the 4 views (:1217) and the mix-up row (:1286-1292): lane = (sample ls, channels 4 q .. 4 q + 3), 96 of them per
        // tile; the rows [colfeat45 | merged45 | 0 0] are put together in LDS (over the planes, which are dead now) and leave as 16-B chunks
        float *s_x7 = reinterpret_cast<float *>(wl);                       // [8][92]
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p = lane + 64 * it, ls = p / 12, q = p - ls * 12;
            long long sidx = sbase + ls;
            if (sidx >= n_samp) sidx = n_samp - 1;
            if (p < 96) {
                float4 fv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) fv[v] = *reinterpret_cast<const float4 *>(a.fm + (size_t)s_pix[4 * ls + v] + 4 * q);
                const float4 c4 = *reinterpret_cast<const float4 *>(a.CF + (size_t)sidx * a.ldcf + 4 * q);
                float fs[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float wv = s_w[4 * ls + v];
                    fs[0] += fv[v].x * wv; fs[1] += fv[v].y * wv; fs[2] += fv[v].z * wv; fs[3] += fv[v].w * wv;
                    wsum += wv;
                }
                const float den = wsum + 1e-6f;
                float *o = s_x7 + ls * 92;
                o[4 * q] = c4.x; o[45 + 4 * q] = hnr_div(fs[0], den);
                if (q < 11) {
                    o[4 * q + 1] = c4.y; o[4 * q + 2] = c4.z; o[4 * q + 3] = c4.w;
                    o[45 + 4 * q + 1] = hnr_div(fs[1], den); o[45 + 4 * q + 2] = hnr_div(fs[2], den); o[45 + 4 * q + 3] = hnr_div(fs[3], den);
                } else { o[90] = 0.f; o[91] = 0.f; }
            }
        }
        MW_WAVE_FENCE();
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = lane + 64 * it, ls = idx / 23, c4 = idx - ls * 23;
            const long long sidx = sbase + ls;
            if (idx < 8 * 23 && sidx < n_samp) {
                const f32x4m v = *reinterpret_cast<const f32x4m *>(s_x7 + ls * 92 + 4 * c4);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4m *>(a.X7 + (size_t)sidx * a.ld7 + 4 * c4));
            }
        }
        MW_WAVE_FENCE();                                                   // the per-row words and planes are rewritten by the wave's next tile
    }
}

// ---- mix-up stage, one wave per 32 samples: color_mixup_block (90 -> 45 -> 45 -> 45, :1285-1292) + learn_residuals (:1294) +
// color_final_block + sigmoid decode (:1295, :1334, :478-482), scattered with sigma into the decoded rows.  Same layout as merge_wp_kernel
// (weights, biases in LDS; no workgroup barrier in the tile loop); the arithmetic is that of mlp3_kernel<6, 3, 3, 0, 0, RT> followed by
// final_color_kernel element for element -- including the order of final_color_kernel's sum over the 128 columns (16 partial sums of 8
// columns, combined as a tree), which is rebuilt here from the two lanes that hold a sample's row.
struct MixArgs {
    const float *X7; int ld7;                            // [S, ld7 >= 92] mix-up rows (90 columns + padding)
    const char *wimg;                                    // hnr_mlp_pack image of the three layers (12 k steps)
    const float *CF; int ldcf;                           // colour feature [S,128]
    const float *w_fin, *b_fin;                          // color_final_block.0 [3,128], [3]
    const float *sigma; const int32_t *vs_item;
    const unsigned long long *counts; long long S_cap;
    float slope;
    float *Y; int ldy;                                   // optional: the mix-up output rows [S, ldy >= 48]
    float *decoded;                                      // [R*SR,4]
};
constexpr int MX_WAVES = 12;
constexpr int MX_WAVE_LDS = 3 * MW_SLOT + 32 * 4;                          // planes of 3 k steps (layer 0 goes through them in two halves of 48 columns), row maxima
constexpr int MX_WBYTES = 12 * 4096;
constexpr int MX_SHARED = MX_WBYTES + 3 * 64 * 4 + 3 * 128 * 4;            // + bias [3][64], color_final_block weights [3][128]
constexpr int MX_LDS = MX_SHARED + MX_WAVES * MX_WAVE_LDS;
__global__ __launch_bounds__(64 * MX_WAVES, 1) void mixfinal_wp_kernel(MixArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long n_samp = a.S_cap;
    if (a.counts) { const long long c = (long long)a.counts[HNR_CNT_SAMPLES_VALID]; if (c < n_samp) n_samp = c; }
    const int n_tiles = (int)((n_samp + 31) / 32);
    const float *meta = reinterpret_cast<const float *>(a.wimg + (size_t)12 * ML_WSTEP);
    float *s_bias = reinterpret_cast<float *>(lds + MX_WBYTES);            // [3][64]
    float *s_fin = s_bias + 3 * 64;                                        // [3][128]
    char *wl = lds + MX_SHARED + wave * MX_WAVE_LDS;
    unsigned *s_max = reinterpret_cast<unsigned *>(wl + 3 * MW_SLOT);
    for (int i = tid; i < MX_WBYTES / 16; i += 64 * MX_WAVES) {            // chunk i = (s, c, p, lane)
        const int s = i >> 8, cp = (i >> 6) & 3, ln = i & 63;
        *reinterpret_cast<u32x4 *>(lds + i * 16) = *reinterpret_cast<const u32x4 *>(a.wimg + (size_t)s * ML_WSTEP + cp * 1024 + ln * 16);
    }
    for (int i = tid; i < 3 * 64; i += 64 * MX_WAVES) s_bias[i] = meta[(i >> 6) * 128 + (i & 63)];
    for (int i = tid; i < 3 * 128; i += 64 * MX_WAVES) s_fin[i] = a.w_fin[i];
    __syncthreads();
    const float dw0 = meta[ML_DESC + 0], dw1 = meta[ML_DESC + 1], dw2 = meta[ML_DESC + 2];
    const float bfin[3] = {a.b_fin[0], a.b_fin[1], a.b_fin[2]};
    const f32x2 slope2 = {a.slope, a.slope};
    for (int tile = blockIdx.x * MX_WAVES + wave; tile < n_tiles; tile += gridDim.x * MX_WAVES) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int h = lane >> 5, j = lane & 31;
        const long long sbase = (long long)tile * 32;
        if (h == 0) s_max[j] = 0u;
        MW_WAVE_FENCE();
        // (a) the 32 rows as 32 x 24 chunks of 4 columns (90 real columns, 96 in the planes), 12 per lane, read as contiguous bursts;
        //     row maxima through LDS, power-of-two scale, fp16 split; the layer-0 operand planes are written (and multiplied) in two halves
        //     of 48 columns, so that a wave needs the LDS of 3 k steps only
        float inv;
        float4 f4[12];
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int idx = lane + 64 * it, row = idx / 24, q = idx - row * 24;
            long long sidx = sbase + row;
            if (sidx >= n_samp) sidx = n_samp - 1;
            if (q < 23) {
                f4[it] = *reinterpret_cast<const float4 *>(a.X7 + (size_t)sidx * a.ld7 + 4 * q);
            } else f4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int idx = lane + 64 * it, row = idx / 24, q = idx - row * 24;
            if (q == 22) { f4[it].z = 0.f; f4[it].w = 0.f; }               // columns 90, 91: padding of the rows
            const float m = fmaxf(fmaxf(fabsf(f4[it].x), fabsf(f4[it].y)), fmaxf(fabsf(f4[it].z), fabsf(f4[it].w)));
            atomicMax(s_max + row, __float_as_uint(m));
        }
        MW_WAVE_FENCE();
        auto planes_half = [&](int half) {
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int idx = lane + 64 * it, row = idx / 24, q = idx - row * 24, c = 4 * q - 48 * half;
                if (c >= 0 && c < 48) {
                    const float sc = pow2f(row_scale_exp(__uint_as_float(s_max[row])));
                    unsigned ph0, pm0, ph1, pm1;
                    split2h(__fmul_rn(f4[it].x, sc), __fmul_rn(f4[it].y, sc), ph0, pm0);
                    split2h(__fmul_rn(f4[it].z, sc), __fmul_rn(f4[it].w, sc), ph1, pm1);
                    char *dst = wl + (c >> 4) * MW_SLOT + ((((c >> 3) & 1) * 32 + row) * 16) + (c & 7) * 2;
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(ph0, ph1);
                    *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(pm0, pm1);
                }
            }
        };
        planes_half(0);
        inv = __fmul_rn(pow2f(-row_scale_exp(__uint_as_float(s_max[j]))), dw0);
        MW_WAVE_FENCE();
        f32x16 acc[2];
        auto zero_acc = [&]() {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        };
        auto activate = [&](int layer, bool act) -> float {
            float m = 0.f;
            const f32x2 inv2 = {inv, inv};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 b = *reinterpret_cast<const float4 *>(s_bias + layer * 64 + 32 * c + 16 * h + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * q4 + e;
                        const f32x2 add = e ? f32x2{b.z, b.w} : f32x2{b.x, b.y};
                        f32x2 v = f32x2_fma(f32x2{acc[c][2 * q], acc[c][2 * q + 1]}, inv2, add);
                        if (act) { const f32x2 sv = v * slope2; v.x = fmaxf(v.x, sv.x); v.y = fmaxf(v.y, sv.y); }
                        acc[c][2 * q] = v.x; acc[c][2 * q + 1] = v.y;
                        m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
                    }
                }
            }
            return m;
        };
        auto publish = [&](float dw, float m) {
            m = fmaxf(m, __shfl_xor(m, 32));
            const int k = row_scale_exp(m);
            const float sc = pow2f(k);
            const f32x2 sc2 = {sc, sc};
            inv = __fmul_rn(pow2f(-k), dw);
            MW_WAVE_FENCE();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned ph[8], pm[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x2 vs = f32x2{acc[c][2 * q], acc[c][2 * q + 1]} * sc2;
                    split2h(vs.x, vs.y, ph[q], pm[q]);
                }
                if (2 * c + h < 3) {                                       // (columns 48..63 are padding: the next layers read 3 k steps)
                    char *dst = wl + (2 * c + h) * MW_SLOT + j * 16;
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<u32x4 *>(dst + 512) = u32x4{ph[4], ph[5], ph[6], ph[7]};
                    *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                    *reinterpret_cast<u32x4 *>(dst + 1024 + 512) = u32x4{pm[4], pm[5], pm[6], pm[7]};
                }
            }
            MW_WAVE_FENCE();
        };
        const char *wp = lds + lane * 16, *bp = wl + lane * 16;
        long long srow = sbase + j;
        const bool ok = srow < n_samp;
        if (!ok) srow = n_samp - 1;
        zero_acc();
        mw_layer<3>(wp, bp, acc);
        MW_WAVE_FENCE();
        planes_half(1);
        MW_WAVE_FENCE();
        mw_layer<3>(wp + 3 * 4096, bp, acc);
        publish(dw1, activate(0, true));
        zero_acc();
        mw_layer<3>(wp + 6 * 4096, bp, acc);
        publish(dw2, activate(1, true));
        // the sample's colour feature: this lane's columns 16 h + [0, 16), 32 + 16 h + [0, 16), 64 + 32 h + [0, 32) (asked for under the last layer)
        const float *cfrow = a.CF + (size_t)srow * a.ldcf;
        float4 cfa[2][4], cfb[8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) cfa[c][q4] = *reinterpret_cast<const float4 *>(cfrow + 32 * c + 16 * h + 4 * q4);
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) cfb[q4] = *reinterpret_cast<const float4 *>(cfrow + 64 + 32 * h + 4 * q4);
        zero_acc();
        mw_layer<3>(wp + 9 * 4096, bp, acc);
        (void)activate(2, false);
        if (a.Y && ok) {
            float *o = a.Y + (size_t)srow * a.ldy;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int col = 32 * c + 16 * h + 4 * q4;
                    if (col + 4 <= 48) *reinterpret_cast<float4 *>(o + col) = make_float4(acc[c][4 * q4], acc[c][4 * q4 + 1], acc[c][4 * q4 + 2], acc[c][4 * q4 + 3]);
                }
        }
        // ---- residual (mix-up output + colour feature on the first 45 columns) and the three dot products over the 128 columns.
        // final_color_kernel sums them as 16 partial sums p_i of 8 columns (lane i of 16), then by the tree
        //   ((p0+p1)+(p2+p3)) + ((p7+p6)+(p5+p4))  +  (((p15+p14)+(p13+p12)) + ((p8+p9)+(p10+p11)));
        // lane (j, h = 0) holds the columns of p0 p1 p4 p5 p8..p11, lane (j, 1) those of p2 p3 p6 p7 p12..p15.
        float x[2][16];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float cfv = reinterpret_cast<const float *>(&cfa[c][0])[r];
                x[c][r] = (32 * c + 16 * h + r < 45) ? acc[c][r] + cfv : cfv;
            }
        float rj[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float *wf = s_fin + o * 128;
            float pa[2][2];                                                // [c][half of 8 columns]: p_{4 c + 2 h + half}
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float r = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) r += x[c][8 * hf + e] * wf[32 * c + 16 * h + 8 * hf + e];
                    pa[c][hf] = r;
                }
            float pb[4];                                                   // p_{8 + 4 h + i}
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float r = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) r += reinterpret_cast<const float *>(&cfb[0])[8 * i + e] * wf[64 + 32 * h + 8 * i + e];
                pb[i] = r;
            }
            const float u0 = pa[0][0] + pa[0][1];                          // h = 0: p0+p1   h = 1: p2+p3
            const float u1 = pa[1][0] + pa[1][1];                          // h = 0: p4+p5   h = 1: p6+p7
            const float uh = (pb[0] + pb[1]) + (pb[2] + pb[3]);            // h = 0: (p8+p9)+(p10+p11)   h = 1: (p12+p13)+(p14+p15)
            const float v0 = __shfl_xor(u0, 32), v1 = __shfl_xor(u1, 32), vh = __shfl_xor(uh, 32);
            // (written for h = 0; lane h = 1 computes a mirrored, unused value)
            const float lo = (u0 + v0) + (v1 + u1);
            const float hi = vh + uh;
            rj[o] = lo + hi;
        }
        if (ok && h == 0) {
            float4 out;
            out.x = a.sigma[srow];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float sg = hnr_div(1.f, 1.f + expf(-(rj[o] + bfin[o])));
                const float cch = sg * (1.f + 2.f * 0.001f) - 0.001f;
                if (o == 0) out.y = cch; else if (o == 1) out.z = cch; else out.w = cch;
            }
            reinterpret_cast<float4 *>(a.decoded)[a.vs_item[srow]] = out;
        }
        MW_WAVE_FENCE();
    }
}

struct MlpPackArgs {
    const float *W[4]; int ldw[4], N[4], K[4], S[4], base[4];
    const float *b[4];
    char *out; int total_steps;
};

__global__ void mlp_wmax_kernel(MlpPackArgs a)
{
    const int l = blockIdx.y;
    unsigned *wmax = reinterpret_cast<unsigned *>(a.out + (size_t)a.total_steps * ML_WSTEP) + ML_WMAX;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N[l] * a.K[l]; i += gridDim.x * blockDim.x) {
        const int n = i / a.K[l], k = i - n * a.K[l];
        m = fmaxf(m, fabsf(a.W[l][(size_t)n * a.ldw[l] + k]));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(wmax + l, __float_as_uint(m));
}

__global__ void mlp_pack_kernel(MlpPackArgs a)
{
    const int l = blockIdx.y;
    float *meta = reinterpret_cast<float *>(a.out + (size_t)a.total_steps * ML_WSTEP);
    const unsigned maxbits = reinterpret_cast<const unsigned *>(meta)[ML_WMAX + l];
    int ex = (int)((maxbits >> 23) & 0xffu);
    ex = ex < 110 ? 110 : (ex > 160 ? 160 : ex);
    const int sw = CH_W_EXP + 126 - ex;
    const float scale = pow2f(sw);
    const int total = a.S[l] * 4 * 64 * 8;                                     // (s, ct, lane, e)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 7, ln = (i >> 3) & 63, ct = (i >> 9) & 3, s = i >> 11;
        const int ii = ln & 31, hh = ln >> 5;
        const int n = 32 * ct + 16 * ((ii >> 2) & 1) + (ii & 3) + 4 * (ii >> 3), k = 16 * s + 8 * hh + e;
        const float x = (n < a.N[l] && k < a.K[l]) ? __fmul_rn(a.W[l][(size_t)n * a.ldw[l] + k], scale) : 0.f;
        const _Float16 hv = (_Float16)x;
        const _Float16 mv = (_Float16)__fsub_rn(x, (float)hv);
        _Float16 *dst = reinterpret_cast<_Float16 *>(a.out + a.base[l] + (size_t)s * ML_WSTEP + (ct * 2) * 1024 + ln * 16) + e;
        dst[0] = hv;
        dst[512] = mv;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 128; i += blockDim.x) meta[l * 128 + i] = (a.b[l] && i < a.N[l]) ? a.b[l][i] : 0.f;
        if (threadIdx.x == 0) meta[ML_DESC + l] = pow2f(-sw);
    }
}

}  // namespace hnr

using namespace hnr;

static int mlp_steps(int K) { return (K + 15) / 16; }

extern "C" int64_t hnr_mlp3_packed_bytes(int n_layers, const int *K)
{
    if (!K || (n_layers != 3 && n_layers != 4)) return -1;
    int64_t steps = 0;
    for (int l = 0; l < n_layers; ++l) { if (K[l] <= 0 || K[l] > 288) return -1; steps += mlp_steps(K[l]); }
    return steps * ML_WSTEP + ML_META_FLOATS * 4;
}

extern "C" int hnr_mlp3_pack(int n_layers, const float *const *d_W, const int *ldw, const int *N, const int *K, const float *const *d_bias, void *d_packed,
                             void *stream)
{
    if ((n_layers != 3 && n_layers != 4) || !d_W || !ldw || !N || !K || !d_bias || !d_packed || ((uintptr_t)d_packed & 15)) { set_error("hnr_mlp3_pack: NULL / unaligned pointer or n_layers not 3 / 4"); return HNR_ERR_BADARG; }
    MlpPackArgs a;
    int steps = 0;
    for (int l = 0; l < n_layers; ++l) {
        // layer l > 0 reads layer l-1's output; the tail (l = 3) reads layer 2's
        if (!d_W[l] || N[l] <= 0 || N[l] > 128 || K[l] <= 0 || K[l] > 288 || ldw[l] < K[l] || (l > 0 && K[l] != N[l == 3 ? 2 : l - 1])) {
            set_error("hnr_mlp3_pack: layer %d: N=%d (1..128) K=%d (1..288, = N of the layer it reads) ldw=%d", l, N[l], K[l], ldw[l]);
            return HNR_ERR_BADARG;
        }
        a.W[l] = d_W[l]; a.ldw[l] = ldw[l]; a.N[l] = N[l]; a.K[l] = K[l]; a.S[l] = mlp_steps(K[l]); a.b[l] = d_bias[l];
        a.base[l] = steps * ML_WSTEP;
        steps += a.S[l];
    }
    a.out = (char *)d_packed; a.total_steps = steps;
    hipStream_t st = (hipStream_t)stream;
    HNR_HIP_CHECK(hipMemsetAsync(a.out + (size_t)steps * ML_WSTEP, 0, ML_META_FLOATS * 4, st));
    mlp_wmax_kernel<<<dim3(16, n_layers), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    mlp_pack_kernel<<<dim3(32, n_layers), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

static void mlp3_probe_print(hipStream_t st, int n_layers, int K0)
{
#ifdef HNR_MLP_PROBE
    if (getenv("HNR_MLP_PROBE_PRINT")) {
        (void)hipStreamSynchronize(st);
        long long h[24];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mlp_probe), sizeof(h)) == hipSuccess && h[20] > 0) {
            const char *nm[13] = {"wait/prev", "A load + split", "barrier", "L0 mfma", "L0 act", "L0 publish", "L1 mfma+act", "L1 publish", "L2 mfma", "L2 act", "store", "tail", "end barrier"};
            fprintf(stderr, "[mlp3 probe] n_layers %d K0 %d tiles %lld:", n_layers, K0, h[20]);
            for (int i = 0; i < 13; ++i) fprintf(stderr, " %s %lld", nm[i], h[i] / h[20]);
            fprintf(stderr, " [merge (a) %lld | tail: dot %lld, sigmoid %lld, loads %lld, merge+stores %lld]", h[13] / h[20], h[14] / h[20], h[15] / h[20], h[16] / h[20], h[17] / h[20]);
            fprintf(stderr, "\n");
        }
    }
#else
    (void)st; (void)n_layers; (void)K0;
#endif
}

static int mlp3_forward_impl(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                             const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                             const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, float *d_T0, int ldt0, float *d_T1, int ldt1,
                             uint32_t *d_tmax, void *stream);

extern "C" int hnr_mlp3_forward(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                                const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                                const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, void *stream)
{
    return mlp3_forward_impl(d_A, lda, M_cap, d_counts, count_index, count_mult, seg_stride, d_packed, n_layers, N, K, act, slope, d_R, d_ridx, ldr, d_C, ldc,
                             d_C2, ldc2, nullptr, 0, nullptr, 0, nullptr, stream);
}

// training forward (csrc/render_train.hip): the same launch, keeping the outputs of layers 0 and 1 ([M, ldt0 / ldt1], same row mapping as d_C) and the
// three layers' output maxima for the backward pass
namespace hnr {
int mlp3_forward_train(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                       const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                       const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, float *d_T0, int ldt0, float *d_T1, int ldt1,
                       uint32_t *d_tmax, void *stream)
{
    if (!d_T0 || !d_T1 || ldt0 < N[0] || ldt1 < N[1] || (ldt0 & 3) || (ldt1 & 3) || ((uintptr_t)d_T0 & 15) || ((uintptr_t)d_T1 & 15)) { set_error("mlp3_forward_train: bad activation buffers"); return HNR_ERR_BADARG; }
    return mlp3_forward_impl(d_A, lda, M_cap, d_counts, count_index, count_mult, seg_stride, d_packed, n_layers, N, K, act, slope, d_R, d_ridx, ldr, d_C, ldc,
                             d_C2, ldc2, d_T0, ldt0, d_T1, ldt1, d_tmax, stream);
}
}  // namespace hnr

static int mlp3_forward_impl(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                             const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                             const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, float *d_T0, int ldt0, float *d_T1, int ldt1,
                             uint32_t *d_tmax, void *stream)
{
    if ((n_layers != 3 && n_layers != 4) || !N || !K || !act || M_cap < 0 || seg_stride < 0 || (seg_stride > 0 && (count_mult < 1 || count_mult > 8)) || lda < K[0] || (lda & 3) || ldc < N[2] || (ldc & 3) ||
        !(slope > 0.f && slope < 1.f) || (d_R && (!d_ridx || ldr < N[0] || (ldr & 3) || ((uintptr_t)d_R & 15))) || (n_layers == 4 && (!d_C2 || ldc2 < N[3] || (ldc2 & 3) || ((uintptr_t)d_C2 & 15)))) {
        set_error("hnr_mlp3_forward: bad sizes (n_layers=%d lda=%d ldc=%d ldr=%d ldc2=%d slope=%g)", n_layers, lda, ldc, ldr, ldc2, (double)slope);
        return HNR_ERR_BADARG;
    }
    if (M_cap == 0) return HNR_OK;
    if (!d_A || !d_packed || !d_C || ((uintptr_t)d_A & 15) || ((uintptr_t)d_C & 15) || ((uintptr_t)d_packed & 15) || ((uintptr_t)d_R & 7)) {
        set_error("hnr_mlp3_forward: NULL / unaligned pointer");
        return HNR_ERR_BADARG;
    }
    MlpArgs a;
    a.A = d_A; a.lda = lda; a.R = d_R; a.ridx = d_ridx; a.ldr = ldr; a.wimg = (const char *)d_packed;
    int steps = 0, S[4] = {0, 0, 0, 0};
    a.N[3] = 0; a.act[3] = 0; a.wbase[3] = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (N[l] <= 0 || N[l] > 128 || K[l] <= 0 || K[l] > 288 || (l > 0 && K[l] != N[l == 3 ? 2 : l - 1])) { set_error("hnr_mlp3_forward: bad layer %d (N=%d K=%d)", l, N[l], K[l]); return HNR_ERR_BADARG; }
        S[l] = mlp_steps(K[l]); a.wbase[l] = steps * ML_WSTEP; steps += S[l]; a.N[l] = N[l]; a.act[l] = act[l];
    }
    a.K0 = K[0]; a.slope = slope;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.count_index = count_index; a.count_mult = count_mult; a.M_cap = M_cap;
    a.seg_stride = seg_stride;
    a.C = d_C; a.ldc = ldc; a.C2 = d_C2; a.ldc2 = ldc2;
    a.T0 = d_T0; a.ldt0 = ldt0; a.T1 = d_T1; a.ldt1 = ldt1; a.tmax = d_tmax;
    a.loc_w = nullptr; a.vs_item = nullptr; a.w2c = a.Kmat = a.campos = a.campos_n = a.fm = a.frame_w = a.w_last = a.b_last = a.CF = nullptr;
    a.H = a.W = a.ldcf = a.ld7 = 0; a.X7 = nullptr;
    const int n_cu = device_num_cus();
    hipStream_t st = (hipStream_t)stream;
    // RT_ = row tiles per workgroup tile: 2 (64 rows, two workgroups per CU) where four (128 rows) would leave room for one workgroup only
#define HNR_MLP3_CASE(S0_, S1_, S2_, S3_, RT_)                                                                                         \
    if (S[0] == S0_ && S[1] == S1_ && S[2] == S2_ && S[3] == S3_) {                                                                     \
        if (S1_ <= 4 && S2_ <= 4 && S3_ == 0 && RT_ >= 2 && N[2] > 64) {   /* the row-split wave mapping of the narrow variants: two column tiles */ \
            set_error("hnr_mlp3_forward: N[2] = %d > 64 with 64-wide hidden layers is not built", N[2]); return HNR_ERR_BADARG; }             \
        constexpr int smax3 = S0_ > S1_ ? (S0_ > S2_ ? S0_ : S2_) : (S1_ > S2_ ? S1_ : S2_), smax = smax3 > S3_ ? smax3 : S3_;          \
        constexpr int ldsb = smax * (RT_ * 2048 + ML_PAD) + 32 * RT_ * 4 * 4 + 32 * RT_ * 4;                                                       \
        const int64_t tiles = (M_cap + 32 * RT_ - 1) / (32 * RT_);                                                                      \
        const int wgs = mlp3_wgs_per_cu(S0_, RT_, 0) * n_cu, grid = (int)(tiles < wgs ? tiles : wgs);                                              \
        static PerDeviceOnce attr;                                                                                                      \
        if (attr.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp3_kernel<S0_, S1_, S2_, S3_, 0, RT_>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
        mlp3_kernel<S0_, S1_, S2_, S3_, 0, RT_><<<grid, 256, ldsb, st>>>(a);                                                             \
        mlp3_probe_print(st, n_layers, K[0]);                                                                                          \
        HNR_LAUNCH_CHECK();                                                                                                             \
        return HNR_OK;                                                                                                                  \
    }
    HNR_MLP3_CASE(18, 8, 8, 0, 2)     // color_feature_branch: 280 -> 128 -> 128 -> 128 (32-row tiles with 3 or 4 workgroups per CU: 2.63 / 2.61 vs 2.45 ms)
    HNR_MLP3_CASE(18, 8, 8, 8, 2)     // the same + tail 128 -> 64: the colour-feature columns of aux_merge_weight_block.0, once per sample
    HNR_MLP3_CASE(3, 4, 4, 0, 4)      // aux_merge_weight_block: 48 -> 64 -> 64 -> 64
    HNR_MLP3_CASE(6, 3, 3, 0, 2)      // color_mixup_block: 90 -> 45 -> 45 -> 45 (three workgroups per CU: 0.84 -> 0.66 ms; four spill: 0.80)
#undef HNR_MLP3_CASE
    set_error("hnr_mlp3_forward: no kernel for k steps (%d, %d, %d, %d); built: (18,8,8,0) (18,8,8,8) (3,4,4,0) (6,3,3,0)", S[0], S[1], S[2], S[3]);
    return HNR_ERR_BADARG;
}


// Merge stage of the image branch in one launch: reprojection of every valid sample into the 4 reference views + feature gather
// (hnr_proj_rows), the first three layers of aux_merge_weight_block with the per-sample colour-feature addend (hnr_mlp3_forward), the
// last layer + sigmoid + weighted merge + mix-up row (hnr_merge).  The [4 S, 48] rows, the [4 S, 64] hidden activations and the per-row
// weights never reach HBM.
extern "C" int hnr_merge_stage(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c, const float *d_intrinsic,
                               const float *d_campos, const float *d_campos_nearest, const float *d_featmap, int V, int H, int W, const float *d_frame_w,
                               const float *d_pre, int ldpre, const void *d_mlp_mw, const float *d_w_last, const float *d_b_last, const float *d_CF, int ldcf,
                               int cap_samples, float slope, float *d_X7, int ld7, void *stream)
{
    if (V != 4) { set_error("hnr_merge_stage: built for V = 4 reference views (got %d); use hnr_proj_rows + hnr_mlp3_forward + hnr_merge", V); return HNR_ERR_BADARG; }
    if (cap_samples < 0 || H <= 0 || W <= 0 || ldpre < 64 || (ldpre & 3) || ((uintptr_t)d_pre & 15) || ldcf < 45 || ld7 < 90 || !(slope > 0.f && slope < 1.f)) {
        set_error("hnr_merge_stage: bad sizes (cap_samples=%d H=%d W=%d ldpre=%d ldcf=%d ld7=%d)", cap_samples, H, W, ldpre, ldcf, ld7); return HNR_ERR_BADARG;
    }
    if (cap_samples == 0) return HNR_OK;
    if (!d_sample_loc_w || !d_vs_item || !d_counts || !d_w2c || !d_intrinsic || !d_campos || !d_campos_nearest || !d_featmap || !d_pre || !d_mlp_mw ||
        !d_w_last || !d_b_last || !d_CF || !d_X7 || ((uintptr_t)d_featmap & 15) || ((uintptr_t)d_pre & 7) || ((uintptr_t)d_mlp_mw & 15)) {
        set_error("hnr_merge_stage: NULL / unaligned pointer"); return HNR_ERR_BADARG;
    }
    MlpArgs a;
    a.A = nullptr; a.lda = 48; a.R = d_pre; a.ridx = nullptr; a.ldr = ldpre; a.wimg = (const char *)d_mlp_mw;
    a.wbase[0] = 0; a.wbase[1] = 3 * ML_WSTEP; a.wbase[2] = 7 * ML_WSTEP; a.wbase[3] = 0;
    a.K0 = 48; a.N[0] = a.N[1] = a.N[2] = 64; a.N[3] = 0; a.act[0] = a.act[1] = a.act[2] = 1; a.act[3] = 0;
    a.slope = slope;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.count_index = HNR_CNT_SAMPLES_VALID; a.count_mult = 4; a.M_cap = (long long)cap_samples * 4;
    a.seg_stride = 0; a.C = nullptr; a.ldc = 64; a.C2 = nullptr; a.ldc2 = 0;
    a.T0 = a.T1 = nullptr; a.ldt0 = a.ldt1 = 0; a.tmax = nullptr;
    a.loc_w = d_sample_loc_w; a.vs_item = d_vs_item; a.w2c = d_w2c; a.Kmat = d_intrinsic; a.campos = d_campos; a.campos_n = d_campos_nearest;
    a.fm = d_featmap; a.H = H; a.W = W; a.frame_w = d_frame_w; a.w_last = d_w_last; a.b_last = d_b_last; a.CF = d_CF; a.ldcf = ldcf; a.X7 = d_X7; a.ld7 = ld7;
    const int n_cu = device_num_cus();
    // 64-row tiles (16 samples x 4 views), four workgroups per CU: the stage is a chain of dependent memory round trips (sample -> projection ->
    // pixel -> feature rows) and barriers, so it is the number of co-resident workgroups that keeps a CU busy (128-row tiles, two per CU: HNR_MERGE_RT=4)
    static int rt_sel = 0;
    if (rt_sel == 0) { const char *e = getenv("HNR_MERGE_RT"); rt_sel = e ? atoi(e) : 1; if (rt_sel != 4 && rt_sel != 2) rt_sel = 1; }
    // (the wave-per-tile kernel moves the colour-feature and mix-up rows as 16-B chunks, and writes the two padding columns 90, 91 of the mix-up rows)
    const bool wp_ok = ldcf >= 48 && !(ldcf & 3) && !((uintptr_t)d_CF & 15) && ld7 >= 92 && !(ld7 & 3) && !((uintptr_t)d_X7 & 15);
    if (rt_sel == 1 && wp_ok) {
        const int64_t wtiles = ((int64_t)cap_samples + 7) / 8, wg_tiles = (wtiles + MW_WAVES - 1) / MW_WAVES;
        const int g = (int)(wg_tiles < (int64_t)n_cu ? wg_tiles : (int64_t)n_cu);
        static PerDeviceOnce attr_wp;
        if (attr_wp.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_wp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MW_LDS));
        merge_wp_kernel<<<g, 64 * MW_WAVES, MW_LDS, (hipStream_t)stream>>>(a);
        HNR_LAUNCH_CHECK();
        return HNR_OK;
    }
    const int rt_k = rt_sel == 4 ? 4 : 2;
    const int rows = 32 * rt_k, wgs = mlp3_wgs_per_cu(3, rt_k, 1);
    const int64_t tiles = ((int64_t)cap_samples * 4 + rows - 1) / rows;
    const int grid = (int)(tiles < (int64_t)wgs * n_cu ? tiles : (int64_t)wgs * n_cu);
    const int ldsb = 4 * (rt_k * 2048 + ML_PAD) + rows * 4 * 4 + rows * 4 + rows * 48 * 4 + 3 * rows * 4 + 128 * 4;
    static PerDeviceOnce attr;
    if (attr.first()) {
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp3_kernel<3, 4, 4, 0, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ML_SLOT + 128 * 4 * 4 + 128 * 4 + 128 * 48 * 4 + 3 * 128 * 4 + 128 * 4));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp3_kernel<3, 4, 4, 0, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (2 * 2048 + ML_PAD) + 64 * 4 * 4 + 64 * 4 + 64 * 48 * 4 + 3 * 64 * 4 + 128 * 4));
    }
    if (rt_k == 4) mlp3_kernel<3, 4, 4, 0, 1, 4><<<grid, 256, ldsb, (hipStream_t)stream>>>(a);
    else mlp3_kernel<3, 4, 4, 0, 1, 2><<<grid, 256, ldsb, (hipStream_t)stream>>>(a);
    mlp3_probe_print((hipStream_t)stream, 3, 48);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_mixup_stage(const float *d_X7, int ld7, const void *d_mlp_mx, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                               const float *d_sigma, const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples, float slope, float *d_Y, int ldy,
                               float *d_decoded, void *stream)
{
    if (cap_samples < 0 || ld7 < 92 || (ld7 & 3) || ldcf < 128 || (ldcf & 3) || (d_Y && (ldy < 48 || (ldy & 3))) || !(slope > 0.f && slope < 1.f)) {
        set_error("hnr_mixup_stage: bad sizes (cap_samples=%d ld7=%d ldcf=%d ldy=%d)", cap_samples, ld7, ldcf, ldy); return HNR_ERR_BADARG;
    }
    if (cap_samples == 0) return HNR_OK;
    if (!d_X7 || !d_mlp_mx || !d_CF || !d_w_fin || !d_b_fin || !d_sigma || !d_vs_item || !d_counts || !d_decoded || ((uintptr_t)d_X7 & 15) ||
        ((uintptr_t)d_CF & 15) || ((uintptr_t)d_mlp_mx & 15) || ((uintptr_t)d_Y & 15) || ((uintptr_t)d_decoded & 15)) {
        set_error("hnr_mixup_stage: NULL / unaligned pointer"); return HNR_ERR_BADARG;
    }
    MixArgs a;
    a.X7 = d_X7; a.ld7 = ld7; a.wimg = (const char *)d_mlp_mx; a.CF = d_CF; a.ldcf = ldcf; a.w_fin = d_w_fin; a.b_fin = d_b_fin; a.sigma = d_sigma;
    a.vs_item = d_vs_item; a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.S_cap = cap_samples; a.slope = slope;
    a.Y = d_Y; a.ldy = ldy; a.decoded = d_decoded;
    const int n_cu = device_num_cus();
    const int64_t tiles = ((int64_t)cap_samples + 31) / 32, wg_tiles = (tiles + MX_WAVES - 1) / MX_WAVES;
    const int g = (int)(wg_tiles < (int64_t)n_cu ? wg_tiles : (int64_t)n_cu);
    static PerDeviceOnce attr;
    if (attr.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(mixfinal_wp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS));
    mixfinal_wp_kernel<<<g, 64 * MX_WAVES, MX_LDS, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
