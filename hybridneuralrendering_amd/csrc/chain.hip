// The per-neighbour chain of PointAggregator.viewmlp in ONE kernel:
//   block1 (Linear 284->256, LeakyReLU, Linear 256->256, LeakyReLU)        models/aggregators/point_aggregators.py:948
//   block3 on [block1_out | colour3 | dir - viewdir 3 | dir.viewdir 1]       :957-972
//   alpha_branch (Linear 256->1) + softplus(x - 1)                          :1005, :471-476
//   K-weighted sums of the density and of the 256 features per shading sample :1008-1026
// Activations never leave the CU: each workgroup carries a tile of 128 neighbour rows (16 shading samples x K = 8 slots)
// through the four dense layers with the layer outputs staged in LDS, and only the per-sample sums go back to HBM.
// (The unfused path -- linear.hip + ksum_kernel -- writes and re-reads a [rows, 256] fp32 matrix between all
// launches: 232 GB of HBM traffic per 285 200-ray frame against 4 GB of algorithmic bytes.)
//
// Arithmetic: "f16x2".  gfx950 has no TF32 and runs fp32 MFMA at 1/16 of the 16-bit matrix rate.  Every operand x (an
// activation or a weight, scaled by a power of two so that the row / layer maximum sits just below 2^15 / 2^14) is split into
// two fp16 values h = fp16(x), m = fp16(x - h) (x - h is exact), |x - h - m| <= 2^-22 |x|, and a product is issued as the three
// v_mfma_f32_32x32x16_f16 terms  wm*xh + wh*xm + wh*xh  with fp32 accumulation; the dropped terms are <= 3 * 2^-22 |w x| per
// product with random sign, which over a K >= 60 dot product stays below the rounding error of the fp32 accumulation itself
// (tests/test_chain_gpu.py measures the layer against fp64 beside the fp32-MFMA kernel: same error class).  The scaling makes
// the split exact-to-22-bits for every element down to 2^-18 of its row's maximum and keeps fp16 overflow impossible; the
// power-of-two scales are removed exactly in the epilogue.  Per-ROW activation scales make a row's result independent of
// which other rows share its tile, so chunked / sharded renders reproduce the whole-frame pixels bit for bit.
//
// Row layout: sample-major, K = 8 slots per valid shading sample (row = 8 s + k; empty slots carry weight 0), so the K-sum
// is a sum over 8 adjacent MFMA columns.
//
// Orientation: the MFMA computes OUT^T = W * X^T: the A operand is a weight fragment (32 output columns x 16 k), the B
// operand an activation fragment (32 rows x 16 k), so an accumulator lane (j = lane & 31, h = lane >> 5) holds, for
// activation row j, 16 output columns.  The weight rows are dealt to the MFMA tile so that those 16 columns are CONSECUTIVE
// (32 ct + 16 h + r): they are exactly one k step (16 k) of the next layer's B fragment for row j, i.e. the epilogue writes
// the next layer's operand with two 16-B LDS stores per plane and no cross-lane traffic.
//
// Tiling: 256 threads = 4 waves, one per SIMD (up to 512 registers); wave w owns output columns 64 w .. 64 w + 63 of all
// 128 rows (4 row tiles x 2 column tiles = 128 accumulator registers), streams its own weight fragments global -> registers
// (L2-resident: the 848 KiB weight image is read by every CU), and all four waves read the shared activation fragments from
// LDS (ds_read_b128, 8 per k step for 24 MFMAs).
#include <stdlib.h>

#include "chain_defs.h"

namespace hnr {

// RT = 4: one 128-row workgroup per CU (one wave per SIMD, <= 512 registers).  RT = 2: two 64-row workgroups per CU (<= 256
// registers each): they drift out of phase, so one's epilogue (VALU + LDS stores + barriers) runs under the other's MFMAs; the
// price is that each streams the whole weight image for half as many rows.
template <int RT, int DBG>
__global__ __launch_bounds__(256, RT >= 4 ? 1 : 2) void chain_kernel(ChainArgs a)
{
    constexpr int SLOT = ch_slot(RT), ROWS = 32 * RT, SAMPLES = 4 * RT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    if (n_valid > a.cap_samples) n_valid = a.cap_samples;
    const int n_tiles = (n_valid + SAMPLES - 1) / SAMPLES;
    const float *meta = reinterpret_cast<const float *>(a.wimg + CH_META);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, CH_WBYTES, 0x00020000);
    float *exch = reinterpret_cast<float *>(lds + ch_lds_exch(RT));        // [row][wave 4]
    const int col0 = 64 * wave + 16 * h;                                   // this lane's columns: col0 + 32 c + r
    const f32x2 slope2 = {a.slope, a.slope};
    const unsigned woff = (unsigned)(2 * wave) * 2048u + (unsigned)lane * 16u;       // this wave's column tiles 2 wave, 2 wave + 1 of a weight k step

    // the extras k step (slot 16) carries 7 columns: its k = 8..15 half (lanes 32..63 of every fragment) stays zero
    for (int i = tid; i < 2 * RT * 32; i += 256)
        *reinterpret_cast<u32x4 *>(lds + 16 * SLOT + (i >> 5) * 1024 + (32 + (i & 31)) * 16) = u32x4{0u, 0u, 0u, 0u};

    // XCD-aware tile order: block b runs on XCD b & 7; every XCD walks one contiguous eighth of the tiles so that
    // neighbouring samples (which share points, i.e. rows of the per-point table) meet in one L2
    const int xcd = blockIdx.x & 7, nb = (gridDim.x + 7 - xcd) / 8, bi = blockIdx.x >> 3;
    const int per = (n_tiles + 7) / 8, t_lo = xcd * per, t_hi = (t_lo + per < n_tiles) ? t_lo + per : n_tiles;
    const bool xcd_order = gridDim.x >= 8;

    // Two workgroups share a CU (RT = 2) and run the same phase sequence; started together they stay in phase -- both in their
    // MFMA loops (each at half rate), then both in their epilogues (matrix pipe idle).  The second half of the grid (the second
    // workgroup of every CU, by dispatch order) starts half a tile late, so one's epilogue runs under the other's MFMAs.
    if (RT < 4 && a.skew > 0 && blockIdx.x >= gridDim.x / 2) {
        for (int i = 0; i < a.skew; i += 64) __builtin_amdgcn_s_sleep(64);
    }
    // DBG == 2: per-phase cycle counts of block 0 (s_memtime) -> a.dbg as long long [wave][16]
    long long tm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, t_start = 0, w_start = 0;
    int n_my = 0;
    float hmax_run[4] = {0.f, 0.f, 0.f, 0.f};                              // training: running maxima of the four layers' outputs
    float x5max_run = 0.f;
    if (DBG == 2) { t_start = t_prev = clock64(); w_start = wall_clock64(); }
#define CH_STAMP(i_) do { if (DBG == 2) { const long long t_ = clock64(); tm[i_] += t_ - t_prev; t_prev = t_; } } while (0)
    for (int tile = xcd_order ? t_lo + bi : (int)blockIdx.x; tile < (xcd_order ? t_hi : n_tiles); tile += xcd_order ? nb : (int)gridDim.x) {
        // ---- tile prologue: layer-0 operand image -> LDS slots 0..3; per-row scalars
        {
            // 1-KiB chunk c = (row tile rt = c >> 3, k step s = (c >> 1) & 3, plane p = c & 1); a wave moves one chunk per pass
            const char *src = a.xp + (size_t)tile * RT * CH_XP_GROUP + lane * 16;
            u32x4 v[2 * RT];
#pragma unroll
            for (int i = 0; i < 2 * RT; ++i) v[i] = *reinterpret_cast<const u32x4 *>(src + (i * 4 + wave) * 1024);
#pragma unroll
            for (int i = 0; i < 2 * RT; ++i) {
                const int c = i * 4 + wave, rt = c >> 3, sp = c & 7;
                *reinterpret_cast<u32x4 *>(lds + (sp >> 1) * SLOT + (rt * 2 + (sp & 1)) * 1024 + lane * 16) = v[i];
            }
        }
        const char *aux = a.aux + (size_t)tile * RT * CH_AUX_GROUP;
        int pid[RT];
        float wq[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            pid[rt] = reinterpret_cast<const int32_t *>(aux + rt * CH_AUX_GROUP)[j];
            wq[rt] = reinterpret_cast<const float *>(aux + rt * CH_AUX_GROUP + 128)[j];
        }
        // extras of the rows this wave publishes (row tile = wave)
        float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
        if (wave < RT) {
            e0 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32);
            e1 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32 + 16);
        }
        __syncthreads();
        ++n_my;
        CH_STAMP(0);

        f32x16 acc[RT][2];
        float inv[RT];                                                     // per row tile: 1 / (row scale * layer weight scale) of the running layer
        auto zero_acc = [&]() {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][c][r] = 0.f;
        };
        // bias + LeakyReLU in place, per-row maxima; L0 adds the gathered per-point rows.  Packed fp32 math where the ISA has it
        // (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two values per issue slot) and v_max3_f32 for the running maximum.
        float4 tv[RT][2][4];                                               // layer 0: gathered rows of the per-point table
        auto activate = [&](int layer, float (&amax)[RT], auto with_table) {
            constexpr bool TV = decltype(with_table)::value;
            f32x2 bias[2][8];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = *reinterpret_cast<const float4 *>(meta + layer * 256 + col0 + 32 * c + 4 * q);
                    bias[c][2 * q] = f32x2{b.x, b.y}; bias[c][2 * q + 1] = f32x2{b.z, b.w};
                }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float m = 0.f;
                const f32x2 inv2 = {inv[rt], inv[rt]};
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        f32x2 add = bias[c][q];
                        if (TV) { const float4 t4 = tv[rt][c][q >> 1]; add = add + ((q & 1) ? f32x2{t4.z, t4.w} : f32x2{t4.x, t4.y}); }
                        f32x2 v = f32x2_fma(f32x2{acc[rt][c][2 * q], acc[rt][c][2 * q + 1]}, inv2, add);
                        const f32x2 sv = v * slope2;
                        v.x = fmaxf(v.x, sv.x); v.y = fmaxf(v.y, sv.y);       // LeakyReLU, 0 < slope < 1
                        acc[rt][c][2 * q] = v.x; acc[rt][c][2 * q + 1] = v.y;
                        m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
                        if (DBG == 1) {
                            if (a.dbg && a.dbg_layer == layer) {
                                float *o = a.dbg + ((size_t)tile * ROWS + 32 * rt + j) * 256 + col0 + 32 * c + 2 * q;
                                o[0] = v.x; o[1] = v.y;
                            }
                        }
                    }
                amax[rt] = m;
                if (DBG == 3) {
                    // training: the layer's output rows go to HBM (64 B per lane and column tile: the 16 columns this lane owns)
                    float *o = a.H[layer] + ((size_t)tile * ROWS + 32 * rt + j) * a.ldh[layer] + col0;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4 *>(o + 32 * c + 4 * q) = make_float4(acc[rt][c][4 * q], acc[rt][c][4 * q + 1], acc[rt][c][4 * q + 2], acc[rt][c][4 * q + 3]);
                    hmax_run[layer] = fmaxf(hmax_run[layer], m);
                    if (a.hbits && layer < 3) {                                 // the signs of the lane's 32 values (chain_defs.h)
                        unsigned mb = 0u;
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int r = 0; r < 16; ++r) mb = (mb << 1) | (acc[rt][c][r] > 0.f ? 1u : 0u);
                        a.hbits[(size_t)layer * a.hbits_stride + (((size_t)tile * RT + rt) * 4 + wave) * 64 + lane] = mb;
                    }
                }
            }
        };
        // per-row scale from the maxima of all four waves, split, publish the next layer's operand planes
        auto publish = [&](int next_layer, float (&amax)[RT], bool with_extras) {
            float emax = 0.f;
            if (with_extras) emax = fmaxf(fmaxf(fmaxf(fabsf(e0.x), fabsf(e0.y)), fmaxf(fabsf(e0.z), fabsf(e0.w))), fmaxf(fmaxf(fabsf(e1.x), fabsf(e1.y)), fabsf(e1.z)));
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float m = fmaxf(amax[rt], __shfl_xor(amax[rt], 32));
                if (with_extras && rt == wave) m = fmaxf(m, emax);
                if (h == 0) exch[(32 * rt + j) * 4 + wave] = m;
            }
            __syncthreads();                                               // every wave has finished reading the previous planes
            const float dw = meta[CH_META_DESCALE + next_layer];           // 2^-sw of the next layer's weights
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 m4 = *reinterpret_cast<const float4 *>(exch + (32 * rt + j) * 4);
                const int k = row_scale_exp(fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w)));
                const float sc = pow2f(k);
                const f32x2 sc2 = {sc, sc};
                inv[rt] = __fmul_rn(pow2f(-k), dw);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    unsigned ph[8], pm[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x2 vs = f32x2{acc[rt][c][2 * q], acc[rt][c][2 * q + 1]} * sc2;
                        split2h(vs.x, vs.y, ph[q], pm[q]);
                    }
                    char *dst = lds + (2 * (2 * wave + c) + h) * SLOT + (rt * 2) * 1024 + j * 16;
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<u32x4 *>(dst + 512) = u32x4{ph[4], ph[5], ph[6], ph[7]};
                    *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                    *reinterpret_cast<u32x4 *>(dst + 1024 + 512) = u32x4{pm[4], pm[5], pm[6], pm[7]};
                }
                if (DBG == 3 && with_extras && rt == wave && h == 0) {
                    float *o = a.H[1] + ((size_t)tile * ROWS + 32 * rt + j) * a.ldh[1] + 256;       // X3 = [H2 | colour3 | dir - viewdir | dir . viewdir | 0]
                    *reinterpret_cast<float4 *>(o) = e0;
                    *reinterpret_cast<float4 *>(o + 4) = make_float4(e1.x, e1.y, e1.z, 0.f);
                    hmax_run[1] = fmaxf(hmax_run[1], emax);
                }
                if (with_extras && rt == wave && h == 0) {
                    unsigned ph[4], pm[4];
                    split2h(__fmul_rn(e0.x, sc), __fmul_rn(e0.y, sc), ph[0], pm[0]);
                    split2h(__fmul_rn(e0.z, sc), __fmul_rn(e0.w, sc), ph[1], pm[1]);
                    split2h(__fmul_rn(e1.x, sc), __fmul_rn(e1.y, sc), ph[2], pm[2]);
                    split2h(__fmul_rn(e1.z, sc), 0.f, ph[3], pm[3]);
                    char *dst = lds + 16 * SLOT + (rt * 2) * 1024 + j * 16;
                    *reinterpret_cast<u32x4 *>(dst) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                    *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{pm[0], pm[1], pm[2], pm[3]};
                }
            }
            __syncthreads();
        };

        // ---- layer 0: PE5(dists6) (60 columns, scale 2^14 fixed: |sin|, |cos| <= 1) + per-point addend
        {
            const float dw0 = meta[CH_META_DESCALE + 0];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) inv[rt] = __fmul_rn(pow2f(-14), dw0);
            zero_acc();
            // all 16 weight fragments first, THEN the gathered table rows: the MFMAs wait for the (older) weight loads only
            h2_mfma_layer<RT, 2, CH_S0, 1, CH_WSTEP, ch_slot(RT)>(wsrd, CH_W0, woff, lds, lane, acc, [&]() {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    int prow = pid[rt] < 0 ? 0 : pid[rt];
                    if (DBG == 3 && a.uidx) { prow = pid[rt] < 0 ? 0 : a.uidx[prow]; if (prow < 0) prow = 0; }
                    const float *trow = a.ptab + (size_t)prow * a.ldt + col0;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) tv[rt][c][q] = *reinterpret_cast<const float4 *>(trow + 32 * c + 4 * q);
                }
            });
            CH_STAMP(1);
            float amax[RT];
            activate(0, amax, std::true_type{});
            CH_STAMP(2);
            publish(1, amax, false);
            CH_STAMP(3);
        }
        // ---- layer 1 (block1.2) -> operand of block3.0 = [H2 | extras]
        {
            zero_acc();
            h2_mfma_layer<RT, 2, CH_S1, 0, CH_WSTEP, ch_slot(RT)>(wsrd, CH_W1, woff, lds, lane, acc, []() {});
            CH_STAMP(4);
            float amax[RT];
            activate(1, amax, std::false_type{});
            CH_STAMP(5);
            publish(2, amax, true);
            CH_STAMP(6);
        }
        // ---- layer 2 (block3.0)
        {
            zero_acc();
            h2_mfma_layer<RT, 2, CH_S2, 0, CH_WSTEP, ch_slot(RT)>(wsrd, CH_W2, woff, lds, lane, acc, []() {});
            CH_STAMP(7);
            float amax[RT];
            activate(2, amax, std::false_type{});
            CH_STAMP(5);
            publish(3, amax, false);
            CH_STAMP(6);
        }
        // ---- layer 3 (block3.2) + alpha branch + K-weighted sums
        {
            zero_acc();
            h2_mfma_layer<RT, 2, CH_S3, 0, CH_WSTEP, ch_slot(RT)>(wsrd, CH_W3, woff, lds, lane, acc, []() {});
            CH_STAMP(8);
            float amax[RT];
            activate(3, amax, std::false_type{});
            float aw[2][16];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = *reinterpret_cast<const float4 *>(meta + 4 * 256 + col0 + 32 * c + 4 * q);
                    aw[c][4 * q] = b.x; aw[c][4 * q + 1] = b.y; aw[c][4 * q + 2] = b.z; aw[c][4 * q + 3] = b.w;
                }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float ap = 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ap = fmaf(acc[rt][c][r], aw[c][r], ap);
                ap = __fadd_rn(ap, __shfl_xor(ap, 32));
                if (h == 0) exch[(32 * rt + j) * 4 + wave] = ap;
            }
            __syncthreads();                                               // also: the planes are free for the next tile
            const float ab = meta[4 * 256 + 256];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                // weighted feature sums over the sample's 8 rows = 8 adjacent lanes
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float f = __fmul_rn(acc[rt][c][r], wq[rt]);
                        f = __fadd_rn(f, __builtin_amdgcn_update_dpp(0.f, f, 0xB1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
                        f = __fadd_rn(f, __builtin_amdgcn_update_dpp(0.f, f, 0x4E, 0xf, 0xf, false));       // quad_perm [2,3,0,1]
                        f = __fadd_rn(f, __builtin_amdgcn_update_dpp(0.f, f, 0x141, 0xf, 0xf, false));      // row_half_mirror
                        acc[rt][c][r] = f;
                    }
                const int s = tile * SAMPLES + 4 * rt + (j >> 3);
                if ((j & 7) == 0 && s < n_valid) {
                    float *o = a.X5 + (size_t)s * a.ld5 + col0;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4 *>(o + 32 * c + 4 * q) = make_float4(acc[rt][c][4 * q], acc[rt][c][4 * q + 1], acc[rt][c][4 * q + 2], acc[rt][c][4 * q + 3]);
                    if (DBG == 3) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int r = 0; r < 16; ++r) x5max_run = fmaxf(x5max_run, fabsf(acc[rt][c][r]));
                    }
                }
                if (rt == wave) {
                    const float4 d4 = *reinterpret_cast<const float4 *>(exch + (32 * rt + j) * 4);
                    const float d = __fadd_rn(__fadd_rn(d4.x, d4.y), __fadd_rn(d4.z, d4.w));
                    float sg = __fmul_rn(chain_softplus_m1(__fadd_rn(d, ab)), wq[rt]);
                    sg = __fadd_rn(sg, __builtin_amdgcn_update_dpp(0.f, sg, 0xB1, 0xf, 0xf, false));
                    sg = __fadd_rn(sg, __builtin_amdgcn_update_dpp(0.f, sg, 0x4E, 0xf, 0xf, false));
                    sg = __fadd_rn(sg, __builtin_amdgcn_update_dpp(0.f, sg, 0x141, 0xf, 0xf, false));
                    if (h == 0 && (j & 7) == 0 && s < n_valid) a.sigma[s] = sg;
                }
            }
            __syncthreads();                                               // exch is rewritten by the next tile's layer 0
            CH_STAMP(9);
        }
    }
    if (DBG == 2 && blockIdx.x == 0 && lane == 0 && a.dbg) {
        long long *o = reinterpret_cast<long long *>(a.dbg) + wave * 16;
        for (int i = 0; i < 10; ++i) o[i] = tm[i];
        o[10] = clock64() - t_start; o[11] = wall_clock64() - w_start; o[12] = n_my;
    }
    if (DBG == 2 && tid == 0 && a.dbg) {                                   // every block: {cycles, wall ticks, tiles, XCC id}
        long long *o = reinterpret_cast<long long *>(a.dbg) + 64 + 4 * (size_t)blockIdx.x;
        o[0] = clock64() - t_start; o[1] = wall_clock64() - w_start; o[2] = n_my; o[3] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf) | ((long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8);
    }
    if (DBG == 3 && a.hmax) {
        // one atomic per workgroup and layer
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float m = hmax_run[l];
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) exch[4 * l + wave] = m;
        }
        __syncthreads();
        if (tid < 4) { const float m = fmaxf(fmaxf(exch[4 * tid], exch[4 * tid + 1]), fmaxf(exch[4 * tid + 2], exch[4 * tid + 3])); if (m > 0.f) atomicMax(a.hmax + tid, __float_as_uint(m)); }
        if (a.x5max) {
            __syncthreads();
            float m = x5max_run;
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) exch[wave] = m;
            __syncthreads();
            if (tid == 0) { const float mm = fmaxf(fmaxf(exch[0], exch[1]), fmaxf(exch[2], exch[3])); if (mm > 0.f) atomicMax(a.x5max, __float_as_uint(mm)); }
        }
    }
#undef CH_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------
// Gather + geometry for the fused chain (NeuralPoints.forward gather neural_points.py:709-720, w2pers :607-613, dists
// point_aggregators.py:1472-1480, inverse-distance weights :825-833, :1500-1501, x clamp(conf) :1508-1512, positional
// encoding of the distances :930): one 256-thread block per 4 groups (16 valid samples = 128 rows) writes, per group of 32 rows,
//   xp[group]  = layer 0's activation operand PE5(dists6) * 2^14, already split into fp16 (h, m) planes in MFMA fragment order,
//   aux[group] = per row: point id (-1: empty slot), aggregation weight, block3's 7 extra inputs (:957-971),
//   X5[s, 256:280] = view-direction encoding of the sample's ray (:909-913),
// and optionally the reference's `weight` / `conf_coefficient` outputs [R,SR,K].
struct ChainGatherArgs {
    const float *xyz, *conf, *pdir, *color;              // the four point buffers, or
    const float4 *rec;                                   // hnr_point_records: [N][3] float4 {x y z conf | dir.xyz r | g b 0 0} (REC kernels)
    const int32_t *pidx;                                 // [R,SR,8]
    const float *loc_w, *raydir, *campos, *camrot;
    const int32_t *vs_item;
    const unsigned long long *counts;
    int SR, cap_samples;
    char *xp, *aux;
    float *X5; int ld5;
    float *weight_out, *conf_out;
    float *Xd;                                           // training: [rows, 64] fp32 PE5(dists6) (60 columns + 4 zeros): block1.0's distance inputs, for its weight gradient
    int32_t *row_pid;                                    // training: [rows] point id of every row (-1: empty slot)
};

__device__ __forceinline__ void chain_w2pers(const float *p, const float *campos, const float *camrot, float out[3])
{
    const float s0 = __fsub_rn(p[0], campos[0]), s1 = __fsub_rn(p[1], campos[1]), s2 = __fsub_rn(p[2], campos[2]);
    float c[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
        c[q] = __fadd_rn(__fadd_rn(__fmul_rn(camrot[q], s0), __fmul_rn(camrot[3 + q], s1)), __fmul_rn(camrot[6 + q], s2));
    out[0] = hnr_div(c[0], c[2]); out[1] = hnr_div(c[1], c[2]); out[2] = c[2];          // (not `/`: see hnr_div)
}

typedef float f32x4g __attribute__((ext_vector_type(4)));

template <bool REC>
__global__ __launch_bounds__(256) void chain_gather_kernel(ChainGatherArgs a)
{
    __shared__ float s_d[128][7];                        // dists6 per row (odd stride: rows of consecutive lanes on distinct banks)
    const ChainClasses cls = chain_classes(a.counts, a.cap_samples);
    const int tid = threadIdx.x;
    for (int blk = blockIdx.x; blk < cls.n_tiles; blk += gridDim.x) {         // one 128-row tile = 4 groups per pass; the grid is sized from the capacity
    // first class: 16 samples x 8 slots; second class (hnr_chain_plan): 32 samples x 4 slots; third: 64 samples x 2 slots
    const int kc = chain_tile_class(cls, blk);
    const int s_base = chain_tile_first(cls, blk, kc), s_end = chain_class_end(cls, kc);
    if (tid < 128) {
        const int ls = tid >> (3 - kc), kk = tid & ((8 >> kc) - 1);
        const int s = s_base + ls;
        char *aux = a.aux + (size_t)(blk * 4 + (tid >> 5)) * CH_AUX_GROUP;
        const int jr = tid & 31;
        const float cp[3] = {a.campos[0], a.campos[1], a.campos[2]};
        float cr[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) cr[i] = a.camrot[i];
        int pid = -1, item = 0;
        if (s < s_end) { item = a.vs_item[s]; pid = a.pidx[(size_t)item * 8 + kk]; }
        float wraw = 0.f, confc = 0.f, ext[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pid >= 0) {
            const float *lw = a.loc_w + (size_t)item * 3;
            float sp[3], pp[3];
            chain_w2pers(lw, cp, cr, sp);
            // REC: the point's ten floats from ONE 48-byte record (three 16-B loads, one or two 64-B sectors) instead of four scattered
            // 4..12-byte reads from four buffers: the same values, fewer sectors fetched per neighbour
            float4 r0, r1, r2;
            if constexpr (REC) { r0 = a.rec[3 * (size_t)pid]; r1 = a.rec[3 * (size_t)pid + 1]; r2 = a.rec[3 * (size_t)pid + 2]; }
            else {
                r0 = make_float4(a.xyz[3 * (size_t)pid], a.xyz[3 * (size_t)pid + 1], a.xyz[3 * (size_t)pid + 2], a.conf[pid]);
                r1 = make_float4(a.pdir[3 * (size_t)pid], a.pdir[3 * (size_t)pid + 1], a.pdir[3 * (size_t)pid + 2], a.color[3 * (size_t)pid]);
                r2 = make_float4(a.color[3 * (size_t)pid + 1], a.color[3 * (size_t)pid + 2], 0.f, 0.f);
            }
            const float px = r0.x, py = r0.y, pz = r0.z;
            const float pw[3] = {px, py, pz};
            chain_w2pers(pw, cp, cr, pp);
            const float dx = __fsub_rn(px, lw[0]), dy = __fsub_rn(py, lw[1]), dz = __fsub_rn(pz, lw[2]);
            d6[0] = dx; d6[1] = dy; d6[2] = dz;
            d6[3] = __fsub_rn(__fmul_rn(pp[0], pp[2]), __fmul_rn(sp[0], sp[2]));
            d6[4] = __fsub_rn(__fmul_rn(pp[1], pp[2]), __fmul_rn(sp[1], sp[2]));
            d6[5] = __fsub_rn(pp[2], sp[2]);
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            wraw = hnr_div(1.0f, fmaxf(nrm, 1e-6f));
            confc = fminf(fmaxf(r0.w, 0.0001f), 1.0f);
            const int ray = item / a.SR;
            const float vx = a.raydir[3 * (size_t)ray], vy = a.raydir[3 * (size_t)ray + 1], vz = a.raydir[3 * (size_t)ray + 2];
            const float ddx = r1.x, ddy = r1.y, ddz = r1.z;
            ext[0] = r1.w; ext[1] = r2.x; ext[2] = r2.y;
            ext[3] = __fsub_rn(ddx, vx); ext[4] = __fsub_rn(ddy, vy); ext[5] = __fsub_rn(ddz, vz);
            ext[6] = __fadd_rn(__fadd_rn(__fmul_rn(ddx, vx), __fmul_rn(ddy, vy)), __fmul_rn(ddz, vz));
        }
        float sum = wraw;
        sum += __shfl_xor(sum, 1);
        { const float s2 = __shfl_xor(sum, 2); sum += kc > 1 ? 0.f : s2; }  // (a sample of the third class: 2 lanes)
        { const float s4 = __shfl_xor(sum, 4); sum += kc > 0 ? 0.f : s4; }  // (a sample of the second class: 4 lanes)
        const float w = pid >= 0 ? hnr_div(wraw, fmaxf(sum, 1e-8f)) : 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s_d[tid][i] = d6[i];
        if (a.row_pid) a.row_pid[(size_t)blk * 128 + tid] = pid;
        __builtin_nontemporal_store(pid, reinterpret_cast<int32_t *>(aux) + jr);
        __builtin_nontemporal_store(__fmul_rn(w, confc), reinterpret_cast<float *>(aux + 128) + jr);
        __builtin_nontemporal_store(f32x4g{ext[0], ext[1], ext[2], ext[3]}, reinterpret_cast<f32x4g *>(aux + 256 + jr * 32));
        __builtin_nontemporal_store(f32x4g{ext[4], ext[5], ext[6], 0.f}, reinterpret_cast<f32x4g *>(aux + 256 + jr * 32 + 16));
        if (a.weight_out && pid >= 0) { a.weight_out[(size_t)item * 8 + kk] = w; a.conf_out[(size_t)item * 8 + kk] = confc; }
    } else
    for (int t = tid - 128; t < (64 << kc); t += 128) {
        // view-direction encoding: positional_encoding(viewdirs, 4, ori=True)[3:] = [sin(d*4+f) x12 | cos x12]; 6 values per thread
        const int ls = t >> 2, part = t & 3;
        const int s = s_base + ls;
        if (s < s_end) {
            const int ray = a.vs_item[s] / a.SR;
            float *o = a.X5 + (size_t)s * a.ld5 + 256;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int l = part * 6 + i, jj = l % 12, d = jj >> 2, f = jj & 3;
                const float x = __fmul_rn(a.raydir[3 * (size_t)ray + d], (float)(1 << f));
                o[l] = l < 12 ? sinf(x) : cosf(x);
            }
        }
    }
    __syncthreads();
    // operand image of group rt: fragment (k step s, plane): lane (row = L & 31, k = 16 s + 8 (L >> 5) + e), e = 0..7
    // = 4 (dist, freq) pairs [sin, cos]; positional_encoding interleaves [sin, cos] per (dim, freq): column 2 (5 d + f) + {0, 1}
    const int L = tid & 63, rt = tid >> 6;
    char *xp = a.xp + (size_t)(blk * 4 + rt) * CH_XP_GROUP;
#pragma unroll
    for (int s = 0; s < CH_S0; ++s) {
        const int row = 32 * rt + (L & 31), k0 = 16 * s + 8 * (L >> 5);
        unsigned ph[4], pm[4];
        float xd8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = (k0 >> 1) + e;
            float sv = 0.f, cv = 0.f;
            if (p < 30) {
                const int d = p / 5, f = p - 5 * d;
                sincosf(__fmul_rn(s_d[row][d], (float)(1 << f)), &sv, &cv);
            }
            xd8[2 * e] = sv; xd8[2 * e + 1] = cv;
            split2h(__fmul_rn(sv, 16384.f), __fmul_rn(cv, 16384.f), ph[e], pm[e]);
        }
        if (a.Xd) {
            float *o = a.Xd + ((size_t)blk * 128 + row) * 64 + k0;
            *reinterpret_cast<float4 *>(o) = make_float4(xd8[0], xd8[1], xd8[2], xd8[3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(xd8[4], xd8[5], xd8[6], xd8[7]);
        }
        // streaming (nt) stores: the 6.7 GB image is written once here and read once by the chain kernel; with the default policy the kernel
        // ran at 4.1 TB/s of writes (3.44 ms on the 3.5 M-sample probe frame), with nt stores 2.60 ms
        char *dst = xp + (s * 2) * 1024 + L * 16;
        __builtin_nontemporal_store(u32x4{ph[0], ph[1], ph[2], ph[3]}, reinterpret_cast<u32x4 *>(dst));
        __builtin_nontemporal_store(u32x4{pm[0], pm[1], pm[2], pm[3]}, reinterpret_cast<u32x4 *>(dst + 1024));
    }
    __syncthreads();                                     // s_d is rewritten by the next pass
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight image.  Layer l, k step s, column tile ct, plane p (0: h, 1: m): lane (i = lane & 31, hh = lane >> 5) holds
// W[n][16 s + 8 hh + e] * 2^sw, e = 0..7, with n = 32 ct + 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3): MFMA row i of the A operand
// lands in accumulator register r = (i & 3) + 4 (i >> 3) of lane half (i >> 2) & 1, so a lane's 16 registers are 16 consecutive
// output columns.
struct ChainPackArgs {
    const float *W[4]; int ldw[4]; int K[4];
    const float *b[4];
    const float *alpha_w, *alpha_b;
    char *out;
    unsigned *wmax;                    // [4] bit patterns of max |W| per layer
};

__global__ void chain_wmax_kernel(ChainPackArgs a)
{
    const int l = blockIdx.y;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 256 * a.K[l]; i += gridDim.x * blockDim.x) {
        const int n = i / a.K[l], k = i - n * a.K[l];
        m = fmaxf(m, fabsf(a.W[l][(size_t)n * a.ldw[l] + k]));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(a.wmax + l, __float_as_uint(m));     // non-negative floats order like their bit patterns
}

__device__ __forceinline__ int chain_weight_exp(unsigned maxbits)
{
    int ex = (int)((maxbits >> 23) & 0xffu);
    ex = ex < 110 ? 110 : (ex > 160 ? 160 : ex);                               // sw in [-20, 30]
    return CH_W_EXP + 126 - ex;
}

__global__ void chain_pack_kernel(ChainPackArgs a)
{
    const int l = blockIdx.y;
    constexpr int steps[4] = {CH_S0, CH_S1, CH_S2, CH_S3};
    constexpr int base[4] = {CH_W0, CH_W1, CH_W2, CH_W3};
    const int sw = chain_weight_exp(a.wmax[l]);
    const float scale = pow2f(sw);
    const int total = steps[l] * 8 * 64 * 8;                                   // (s, ct, lane, e)
    float *meta = reinterpret_cast<float *>(a.out + CH_META);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 7, ln = (i >> 3) & 63, ct = (i >> 9) & 7, s = i >> 12;
        const int ii = ln & 31, hh = ln >> 5;
        const int n = 32 * ct + 16 * ((ii >> 2) & 1) + (ii & 3) + 4 * (ii >> 3), k = 16 * s + 8 * hh + e;
        const float x = k < a.K[l] ? __fmul_rn(a.W[l][(size_t)n * a.ldw[l] + k], scale) : 0.f;
        const _Float16 hv = (_Float16)x;
        const _Float16 mv = (_Float16)__fsub_rn(x, (float)hv);
        _Float16 *dst = reinterpret_cast<_Float16 *>(a.out + base[l] + (size_t)s * CH_WSTEP + (ct * 2) * 1024 + ln * 16) + e;
        dst[0] = hv;
        dst[512] = mv;
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) {
            meta[l * 256 + i] = a.b[l] ? a.b[l][i] : 0.f;
            if (l == 0) meta[4 * 256 + i] = a.alpha_w[i];
        }
        if (threadIdx.x == 0) {
            meta[CH_META_DESCALE + l] = pow2f(-sw);
            if (l == 0) meta[4 * 256 + 256] = a.alpha_b[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// hnr_chain_plan: the chain's list of valid samples.  Two-level scan over the kept-sample work list (no atomics, deterministic), like
// hnr_sample_plan; with `classes` the samples with more than 4 neighbours come first, those with 1..4 after them (stable in both).
// class of a kept sample: 1 = more than four neighbours (or any, with one class), 2 = three or four (one to four with two classes), 3 = one or two,
// 0 = none.  Valid ids are a prefix of the K slots (reference :494-496; the sorted order too), so slots 0, 2 and 4 decide -- independent loads, not
// a dependent walk over the slots
__device__ __forceinline__ int chain_sample_class(const int32_t *__restrict__ p, int K, int classes)
{
    const int p0 = p[0], p4 = (classes && K > 4) ? p[4] : -1, p2 = (classes > 1 && K > 2) ? p[2] : 0;
    if (p0 < 0) return 0;
    if (!classes || p4 >= 0) return 1;
    return p2 >= 0 ? 2 : 3;
}

__global__ __launch_bounds__(1024) void chain_plan_sum_kernel(const int32_t *__restrict__ work, const int32_t *__restrict__ pidx,
                                                              const unsigned long long *__restrict__ counts, int K, int classes,
                                                              int32_t *__restrict__ block_sums)
{
    __shared__ int s_a[16], s_b[16], s_c[16];
    const int n_items = (int)counts[HNR_CNT_SAMPLES];
    if ((long long)blockIdx.x * 1024 >= n_items) return;             // (the scan kernel reads the sums of the blocks that hold items only)
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int cl = i < n_items ? chain_sample_class(pidx + (size_t)work[i] * K, K, classes) : 0;
    int big = cl == 1 ? 1 : 0, small = cl == 2 ? 1 : 0, tiny = cl == 3 ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { big += __shfl_xor(big, o); small += __shfl_xor(small, o); tiny += __shfl_xor(tiny, o); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = big; s_b[threadIdx.x >> 6] = small; s_c[threadIdx.x >> 6] = tiny; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0, t = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; t += s_c[k]; }
        block_sums[3 * blockIdx.x] = a;
        block_sums[3 * blockIdx.x + 1] = b;
        block_sums[3 * blockIdx.x + 2] = t;
    }
}

__global__ __launch_bounds__(1024) void chain_plan_scan_kernel(const int32_t *__restrict__ work, const int32_t *__restrict__ pidx,
                                                               unsigned long long *__restrict__ counts, int K, int classes,
                                                               const int32_t *__restrict__ block_sums, int n_blocks,
                                                               int32_t *__restrict__ vs_item, int cap_samples)
{
    __shared__ int s_a[16], s_b[16], s_c[16], s_ta[16], s_tb[16];
    __shared__ int s_base[5];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n_items = (int)counts[HNR_CNT_SAMPLES];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    // the grid is sized for the capacity (R * SR kept samples); the blocks past the last kept sample have nothing to place (block 0 stays: it
    // writes the class counters) and their sums -- never written by the sum kernel's early exit -- are not read
    if (blockIdx.x > 0 && (long long)blockIdx.x * 1024 >= n_items) return;
    const int nb_used = (n_items + 1023) / 1024 < n_blocks ? (n_items + 1023) / 1024 : n_blocks;
    int pa = 0, pb = 0, pc = 0, ta = 0, tb = 0;       // big / small / tiny samples in the blocks before this one; big and small samples in all blocks
    for (int k = threadIdx.x; k < nb_used; k += 1024) {
        const int a = block_sums[3 * k], b = block_sums[3 * k + 1], c = block_sums[3 * k + 2];
        ta += a; tb += b;
        if (k < (int)blockIdx.x) { pa += a; pb += b; pc += c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        pa += __shfl_xor(pa, o); pb += __shfl_xor(pb, o); pc += __shfl_xor(pc, o); ta += __shfl_xor(ta, o); tb += __shfl_xor(tb, o);
    }
    if (lane == 0) { s_a[wid] = pa; s_b[wid] = pb; s_c[wid] = pc; s_ta[wid] = ta; s_tb[wid] = tb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0, c = 0, t = 0, u = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; c += s_c[k]; t += s_ta[k]; u += s_tb[k]; }
        s_base[0] = a; s_base[1] = b; s_base[2] = c; s_base[3] = t; s_base[4] = u;
    }
    __syncthreads();
    const int total_big = s_base[3], total_small = s_base[4];
    int item = 0, cl = 0;
    if (i < n_items) { item = work[i]; cl = chain_sample_class(pidx + (size_t)item * K, K, classes); }
    const int big = cl == 1 ? 1 : 0, small = cl == 2 ? 1 : 0, tiny = cl == 3 ? 1 : 0;
    int ia = big, ib = small, ic = tiny;
    for (int o = 1; o < 64; o <<= 1) {
        const int t0 = __shfl_up(ia, o), t1 = __shfl_up(ib, o), t2 = __shfl_up(ic, o);
        if (lane >= o) { ia += t0; ib += t1; ic += t2; }
    }
    __syncthreads();
    if (lane == 63) { s_a[wid] = ia; s_b[wid] = ib; s_c[wid] = ic; }
    __syncthreads();
    int oa = s_base[0] + ia - big, ob = s_base[1] + ib - small, oc = s_base[2] + ic - tiny;
    for (int k = 0; k < wid; ++k) { oa += s_a[k]; ob += s_b[k]; oc += s_c[k]; }
    const int pos = big ? oa : (small ? total_big + ob : total_big + total_small + oc);
    if (cl != 0 && pos < cap_samples) vs_item[pos] = item;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the later classes as the consumers see them: what is left of the (possibly clamped) valid-sample count after the classes before
        long long nv = (long long)counts[HNR_CNT_SAMPLES_VALID];
        if (nv > cap_samples) nv = cap_samples;
        const long long first = total_big < nv ? total_big : nv;
        const long long second = total_small < nv - first ? total_small : nv - first;
        counts[HNR_CNT_SAMPLES_SMALL] = (unsigned long long)(classes > 1 ? second : nv - first);
        counts[HNR_CNT_SAMPLES_TINY] = (unsigned long long)(classes > 1 ? nv - first - second : 0);
    }
}

}  // namespace hnr

using namespace hnr;

static int chain_num_cus() { return device_num_cus(); }

extern "C" int64_t hnr_chain_packed_bytes(void) { return (int64_t)CH_WBYTES + CH_META_FLOATS * 4; }

extern "C" int64_t hnr_chain_workspace_bytes(int cap_samples)
{
    if (cap_samples < 0) return -1;
    const int64_t groups = 4 * (((int64_t)cap_samples + 15) / 16 + 2);            // whole 4-group blocks of the gather kernel (+ 2: each of the three slot classes ends in a partial block)
    return groups * (CH_XP_GROUP + CH_AUX_GROUP + 32 * 4);                        // operand images, row scalars, density inputs (ChainArgs::dsig)
}

extern "C" int hnr_chain_pack(const float *d_w_b1_0_dist, int ldw0, const float *d_b_b1_0, const float *d_w_b1_2, const float *d_b_b1_2,
                              const float *d_w_b3_0, const float *d_b_b3_0, const float *d_w_b3_2, const float *d_b_b3_2,
                              const float *d_alpha_w, const float *d_alpha_b, void *d_packed, void *stream)
{
    if (!d_w_b1_0_dist || !d_w_b1_2 || !d_w_b3_0 || !d_w_b3_2 || !d_alpha_w || !d_alpha_b || !d_packed || ldw0 < 60 || ((uintptr_t)d_packed & 15)) {
        set_error("hnr_chain_pack: NULL / unaligned pointer or ldw0 < 60");
        return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    ChainPackArgs a;
    a.W[0] = d_w_b1_0_dist; a.ldw[0] = ldw0; a.K[0] = 60; a.b[0] = d_b_b1_0;
    a.W[1] = d_w_b1_2; a.ldw[1] = 256; a.K[1] = 256; a.b[1] = d_b_b1_2;
    a.W[2] = d_w_b3_0; a.ldw[2] = 263; a.K[2] = 263; a.b[2] = d_b_b3_0;
    a.W[3] = d_w_b3_2; a.ldw[3] = 256; a.K[3] = 256; a.b[3] = d_b_b3_2;
    a.alpha_w = d_alpha_w; a.alpha_b = d_alpha_b;
    a.out = (char *)d_packed;
    a.wmax = reinterpret_cast<unsigned *>(a.out + CH_META) + CH_META_WMAX;      // four words of the meta block the pack kernel only reads
    HNR_HIP_CHECK(hipMemsetAsync(a.wmax, 0, 16, st));
    chain_wmax_kernel<<<dim3(64, 4), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    chain_pack_kernel<<<dim3(64, 4), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_chain_classes(void)
{
    // the 4- and 2-slot classes need the weight-stationary kernel (the default); HNR_CHAIN_CLASSES = 0 / 1 limits them
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("HNR_CHAIN_RT"), *c = getenv("HNR_CHAIN_CLASSES");
        const bool ws = !e || atoi(e) != 4;
        const int want = c ? atoi(c) : 2;
        on = ws ? (want < 0 ? 0 : want > 2 ? 2 : want) : 0;
    }
    return on;
}

extern "C" int hnr_chain_plan(const int32_t *d_work, const int32_t *d_sample_pidx, int64_t *d_counts, int K, int max_items, int classes,
                              int32_t *d_vs_item, int cap_samples, int32_t *d_scratch, void *stream)
{
    if (!d_work || !d_sample_pidx || !d_counts || !d_vs_item || !d_scratch || K <= 0 || max_items < 0 || cap_samples < 0 || classes < 0 || classes > 2) {
        set_error("hnr_chain_plan: bad argument"); return HNR_ERR_BADARG;
    }
    if (classes && (K != 8 || classes > hnr_chain_classes())) {
        set_error("hnr_chain_plan: the 4- / 2-slot sample classes need K = 8 and the weight-stationary chain kernel (classes <= hnr_chain_classes() = %d)", hnr_chain_classes()); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (max_items == 0) return HNR_OK;
    const int nb = cdiv(max_items, 1024);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d_counts);
    chain_plan_sum_kernel<<<nb, 1024, 0, st>>>(d_work, d_sample_pidx, cnt, K, classes, d_scratch);
    chain_plan_scan_kernel<<<nb, 1024, 0, st>>>(d_work, d_sample_pidx, cnt, K, classes, d_scratch, nb, d_vs_item, cap_samples);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

namespace hnr {
__global__ void point_records_kernel(const float *__restrict__ xyz, const float *__restrict__ conf, const float *__restrict__ pdir,
                                     const float *__restrict__ color, int N, float4 *__restrict__ rec)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        rec[3 * (size_t)i] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], conf[i]);
        rec[3 * (size_t)i + 1] = make_float4(pdir[3 * (size_t)i], pdir[3 * (size_t)i + 1], pdir[3 * (size_t)i + 2], color[3 * (size_t)i]);
        rec[3 * (size_t)i + 2] = make_float4(color[3 * (size_t)i + 1], color[3 * (size_t)i + 2], 0.f, 0.f);
    }
}
}  // namespace hnr

extern "C" int hnr_point_records(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color, int N, float *d_rec,
                                 void *stream)
{
    if (N < 0 || (N > 0 && (!d_xyz || !d_conf || !d_dir || !d_color || !d_rec || ((uintptr_t)d_rec & 15)))) {
        set_error("hnr_point_records: NULL / unaligned pointer or N < 0"); return HNR_ERR_BADARG;
    }
    if (N == 0) return HNR_OK;
    const int blocks = cdiv(N, 256);
    point_records_kernel<<<blocks < 8192 ? blocks : 8192, 256, 0, (hipStream_t)stream>>>(d_xyz, d_conf, d_dir, d_color, N, reinterpret_cast<float4 *>(d_rec));
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

static int chain_gather_impl(const float *d_rec, const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color,
                             const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir, const float *d_campos,
                             const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR, int K, int cap_samples,
                             void *d_workspace, float *d_X5, int ld5, float *d_weight_out, float *d_conf_out, float *d_Xd, int32_t *d_row_pid, void *stream);

extern "C" int hnr_chain_gather_rec(const float *d_rec, const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir,
                                    const float *d_campos, const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR,
                                    int K, int cap_samples, void *d_workspace, float *d_X5, int ld5, float *d_weight_out,
                                    float *d_conf_out, void *stream)
{
    if (cap_samples > 0 && (!d_rec || ((uintptr_t)d_rec & 15))) { set_error("hnr_chain_gather_rec: NULL / unaligned point records"); return HNR_ERR_BADARG; }
    return chain_gather_impl(d_rec, nullptr, nullptr, nullptr, nullptr, d_sample_pidx, d_sample_loc_w, d_raydir, d_campos, d_camrot, d_vs_item, d_counts,
                             SR, K, cap_samples, d_workspace, d_X5, ld5, d_weight_out, d_conf_out, nullptr, nullptr, stream);
}

extern "C" int hnr_chain_gather(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color,
                                const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir, const float *d_campos,
                                const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR, int K, int cap_samples,
                                void *d_workspace, float *d_X5, int ld5, float *d_weight_out, float *d_conf_out, void *stream)
{
    if (cap_samples > 0 && (!d_xyz || !d_conf || !d_dir || !d_color)) { set_error("hnr_chain_gather: NULL / unaligned pointer"); return HNR_ERR_BADARG; }
    return chain_gather_impl(nullptr, d_xyz, d_conf, d_dir, d_color, d_sample_pidx, d_sample_loc_w, d_raydir, d_campos, d_camrot, d_vs_item, d_counts,
                             SR, K, cap_samples, d_workspace, d_X5, ld5, d_weight_out, d_conf_out, nullptr, nullptr, stream);
}

static int chain_gather_impl(const float *d_rec, const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color,
                             const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir, const float *d_campos,
                             const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR, int K, int cap_samples,
                             void *d_workspace, float *d_X5, int ld5, float *d_weight_out, float *d_conf_out, float *d_Xd, int32_t *d_row_pid, void *stream)
{
    if (K != 8) { set_error("hnr_chain_gather: the fused chain is built for K = 8 (got %d); use the per-layer path", K); return HNR_ERR_BADARG; }
    if (cap_samples < 0 || SR <= 0 || ld5 < 280 || (ld5 & 3)) { set_error("hnr_chain_gather: bad sizes (cap_samples=%d SR=%d ld5=%d)", cap_samples, SR, ld5); return HNR_ERR_BADARG; }
    if (cap_samples == 0) return HNR_OK;
    if (!d_sample_pidx || !d_sample_loc_w || !d_raydir || !d_campos || !d_camrot || !d_vs_item ||
        !d_counts || !d_workspace || !d_X5 || ((uintptr_t)d_workspace & 15) || (!d_weight_out != !d_conf_out)) {
        set_error("hnr_chain_gather: NULL / unaligned pointer");
        return HNR_ERR_BADARG;
    }
    const int blocks = cdiv(cap_samples, 16) + 2;
    ChainGatherArgs a;
    a.xyz = d_xyz; a.conf = d_conf; a.pdir = d_dir; a.color = d_color; a.rec = reinterpret_cast<const float4 *>(d_rec); a.pidx = d_sample_pidx; a.loc_w = d_sample_loc_w;
    a.raydir = d_raydir; a.campos = d_campos; a.camrot = d_camrot; a.vs_item = d_vs_item;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.SR = SR; a.cap_samples = cap_samples;
    a.xp = (char *)d_workspace; a.aux = (char *)d_workspace + (size_t)blocks * 4 * CH_XP_GROUP;
    a.X5 = d_X5; a.ld5 = ld5; a.weight_out = d_weight_out; a.conf_out = d_conf_out; a.Xd = d_Xd; a.row_pid = d_row_pid;
    if (d_rec) chain_gather_kernel<true><<<blocks < 16384 ? blocks : 16384, 256, 0, (hipStream_t)stream>>>(a);
    else chain_gather_kernel<false><<<blocks < 16384 ? blocks : 16384, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_chain_forward(const void *d_workspace, const float *d_point_table, int ldt, const void *d_packed, const int64_t *d_counts,
                                 int cap_samples, float slope, float *d_X5, int ld5, float *d_sigma, float *d_dbg, int dbg_layer, void *stream)
{
    if (cap_samples < 0 || ldt < 256 || (ldt & 3) || ld5 < 256 || (ld5 & 3) || !(slope > 0.f && slope < 1.f)) {
        set_error("hnr_chain_forward: bad sizes (cap_samples=%d ldt=%d ld5=%d slope=%g; LeakyReLU slope must be in (0,1))", cap_samples, ldt, ld5, (double)slope);
        return HNR_ERR_BADARG;
    }
    if (cap_samples == 0) return HNR_OK;
    if (!d_workspace || !d_point_table || !d_packed || !d_counts || !d_X5 || !d_sigma || ((uintptr_t)d_workspace & 15) || ((uintptr_t)d_packed & 15) ||
        ((uintptr_t)d_point_table & 15) || ((uintptr_t)d_X5 & 15)) {
        set_error("hnr_chain_forward: NULL / unaligned pointer");
        return HNR_ERR_BADARG;
    }
    const int blocks = cdiv(cap_samples, 16) + 2;
    ChainArgs a;
    a.xp = (const char *)d_workspace; a.aux = (const char *)d_workspace + (size_t)blocks * 4 * CH_XP_GROUP;
    a.dsig = reinterpret_cast<float *>((char *)d_workspace + (size_t)blocks * 4 * (CH_XP_GROUP + CH_AUX_GROUP));
    a.ptab = d_point_table; a.ldt = ldt; a.wimg = (const char *)d_packed;
    { static int tab0 = -1; if (tab0 < 0) { const char *e = getenv("HNR_CHAIN_PROBE_TAB0"); tab0 = e ? atoi(e) : 0; } if (tab0) a.ldt = 0; }      // probe: every row reads table row 0 (what the gather's latency costs; results are garbage)
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.X5 = d_X5; a.ld5 = ld5; a.sigma = d_sigma; a.slope = slope; a.cap_samples = cap_samples; a.dbg = d_dbg; a.dbg_layer = dbg_layer;
    const int n_cu = chain_num_cus();
    hipStream_t st = (hipStream_t)stream;
    // HNR_CHAIN_RT = 4: the compiler-scheduled one-workgroup-per-CU kernel of this file (the training forward runs its activation-keeping form);
    // default: the weight-stationary pipelined kernel (csrc/chain_ws.hip)
    static int rt_mode = 0;
    if (rt_mode == 0) { const char *e = getenv("HNR_CHAIN_RT"); rt_mode = (e && atoi(e) == 4) ? 4 : 16; }
    a.skew = 0; a.uidx = nullptr; a.hmax = nullptr; a.x5max = nullptr; a.row_u = nullptr; a.ucap = 0; a.hbits = nullptr; a.hbits_stride = 0;
    for (int l = 0; l < 4; ++l) { a.H[l] = nullptr; a.ldh[l] = 0; }
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, ch_lds_bytes(4)));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, ch_lds_bytes(4)));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ch_lds_bytes(4)));
    }
    const int tiles = cdiv(cap_samples, 16), grid = tiles < n_cu ? tiles : n_cu;
    if (rt_mode == 16)                                                      // weight-stationary pipelined kernel (csrc/chain_ws.hip)
        return launch_chain_ws(a, grid, st, (d_dbg && dbg_layer <= -3 && dbg_layer >= -7) ? -dbg_layer : (d_dbg && dbg_layer < 0) ? 2 : d_dbg ? 1 : 0);
    if (d_dbg && dbg_layer < 0) chain_kernel<4, 2><<<grid, 256, ch_lds_bytes(4), st>>>(a);      // probe: per-phase cycle counts of block 0
    else if (d_dbg) chain_kernel<4, 1><<<grid, 256, ch_lds_bytes(4), st>>>(a);
    else chain_kernel<4, 0><<<grid, 256, ch_lds_bytes(4), st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Training forward of the chain (csrc/render_train.hip): the gather also leaves block1.0's distance inputs in fp32 and the rows' point ids,
// the chain kernel keeps every layer's output for the backward pass (chain_kernel<4, 3>) and reads the per-point table through d_uidx.
namespace hnr {
int chain_gather_train(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color, const int32_t *d_sample_pidx,
                       const float *d_sample_loc_w, const float *d_raydir, const float *d_campos, const float *d_camrot, const int32_t *d_vs_item,
                       const int64_t *d_counts, int SR, int K, int cap_samples, void *d_workspace, float *d_X5, int ld5, float *d_weight_out,
                       float *d_conf_out, float *d_Xd, int32_t *d_row_pid, void *stream)
{
    return chain_gather_impl(nullptr, d_xyz, d_conf, d_dir, d_color, d_sample_pidx, d_sample_loc_w, d_raydir, d_campos, d_camrot, d_vs_item, d_counts,
                             SR, K, cap_samples, d_workspace, d_X5, ld5, d_weight_out, d_conf_out, d_Xd, d_row_pid, stream);
}

int chain_forward_train(const void *d_workspace, const float *d_point_table, int ldt, const int32_t *d_uidx, const void *d_packed, const int64_t *d_counts,
                        int cap_samples, float slope, float *d_X5, int ld5, float *d_sigma, float *const *d_H, const int *ldh, uint32_t *d_hmax, uint32_t *d_x5max, void *stream,
                        const int32_t *d_row_u, int ucap, uint32_t *d_hbits, long long hbits_stride)
{
    if (cap_samples <= 0) return HNR_OK;
    const int blocks = cdiv(cap_samples, 16) + 2;
    ChainArgs a;
    a.xp = (const char *)d_workspace; a.aux = (const char *)d_workspace + (size_t)blocks * 4 * CH_XP_GROUP;
    a.dsig = reinterpret_cast<float *>((char *)d_workspace + (size_t)blocks * 4 * (CH_XP_GROUP + CH_AUX_GROUP));
    a.ptab = d_point_table; a.ldt = ldt; a.wimg = (const char *)d_packed;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.X5 = d_X5; a.ld5 = ld5; a.sigma = d_sigma; a.slope = slope; a.cap_samples = cap_samples; a.dbg = nullptr; a.dbg_layer = 0; a.skew = 0;
    for (int l = 0; l < 4; ++l) { a.H[l] = d_H[l]; a.ldh[l] = ldh[l]; }
    a.uidx = d_uidx; a.hmax = d_hmax; a.x5max = d_x5max; a.row_u = d_row_u; a.ucap = ucap; a.hbits = d_hbits; a.hbits_stride = hbits_stride;
    const int n_cu = chain_num_cus();
    const int tiles = cdiv(cap_samples, 16), grid = tiles < n_cu ? tiles : n_cu;
    // The weight-stationary pipelined kernel in its activation-keeping form (csrc/chain_ws.hip, chain_ws_kernel<8>: the render path's kernel + eight
    // 16-B stores per pass and lane; same arithmetic, same results bit for bit) when the row -> table-row map is there and every kept buffer's byte
    // offsets fit 31 bits; HNR_TRAIN_CHAIN_WS=0: the layer-by-layer kernel below
    static int use_ws = -1;
    if (use_ws < 0) { const char *e = getenv("HNR_TRAIN_CHAIN_WS"); use_ws = e ? atoi(e) : 1; }
    bool fits = d_row_u != nullptr && d_hmax != nullptr;
    for (int l = 0; l < 4; ++l) fits = fits && d_H[l] != nullptr && (long long)blocks * 128 * ldh[l] * 4 < 0x7fffffffLL && (ldh[l] & 3) == 0;
    if (use_ws && fits) return launch_chain_ws(a, grid, (hipStream_t)stream, 8);
    HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ch_lds_bytes(4)));
    chain_kernel<4, 3><<<grid, 256, ch_lds_bytes(4), (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
}  // namespace hnr
