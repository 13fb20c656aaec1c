// Shared definitions of the fused per-neighbour chain kernels (csrc/chain.hip: one-group and dual-group kernels, packing, gather;
// csrc/chain_ws.hip: the weight-stationary pipelined kernel).  Layouts are described in the header of chain.hip.
#pragma once
#include "hnr_h2.h"

namespace hnr {

// a GROUP = 32 rows = one MFMA row tile = 4 shading samples x 8 slots
constexpr int CH_WSTEP = 16384;                    // weight image bytes per k step: [column tile 8][plane 2][64 lanes][16 B]
constexpr int CH_S0 = 4, CH_S1 = 16, CH_S2 = 17, CH_S3 = 16;      // k steps of the four layers (K = 60, 256, 263, 256)
constexpr int CH_W0 = 0, CH_W1 = CH_S0 * CH_WSTEP, CH_W2 = CH_W1 + CH_S1 * CH_WSTEP, CH_W3 = CH_W2 + CH_S2 * CH_WSTEP;
constexpr int CH_WBYTES = CH_W3 + CH_S3 * CH_WSTEP;               // 53 k steps = 848 KiB
constexpr int CH_META = CH_WBYTES;                 // floats after the image: bias[4][256], alpha_w[256], alpha_b, descale_w[4], max|W| bits[4], pad
constexpr int CH_META_DESCALE = 4 * 256 + 256 + 1, CH_META_WMAX = CH_META_DESCALE + 4;
constexpr int CH_META_FLOATS = CH_META_WMAX + 4 + 3;
constexpr int CH_XP_GROUP = CH_S0 * 2048;          // bytes of one group's layer-0 operand image: [k step 4][plane 2][64 lanes][16 B] = 8 KiB
constexpr int CH_AUX_GROUP = 32 * 4 + 32 * 4 + 32 * 8 * 4;        // pid[32] i32, wagg[32] f32, ext[32][8] f32 = 1280 B
// workgroup tile = RT groups; LDS: 17 k-step slots of [row tile RT][plane 2][64 lanes][16 B], then the float[32 RT][4] exchange area
constexpr int ch_slot(int RT) { return RT * 2048; }
constexpr int ch_lds_exch(int RT) { return 17 * ch_slot(RT); }
constexpr int ch_lds_bytes(int RT) { return ch_lds_exch(RT) + 32 * RT * 4 * 4; }

struct ChainArgs {
    const char *xp;                    // [groups][CH_XP_GROUP] layer-0 operand image (chain_gather_kernel)
    const char *aux;                   // [groups][CH_AUX_GROUP]
    float *dsig;                       // [groups][32] the rows' density inputs (alpha dot of block3's output), one float per row: chain_ws_kernel -> chain_sigma_kernel.  (They were
                                       // parked in the eighth float of the rows' 32-byte `ext` records: 4-byte accesses 32 bytes apart move whole sectors -- 1.1 GB per frame for 0.1 GB of values.)
    const float *ptab; int ldt;        // per-point addend of block1.0: [N, ldt >= 256]
    const char *wimg;                  // packed weights (hnr_chain_pack)
    const unsigned long long *counts;  // device counters of the query (n_valid samples)
    float *X5; int ld5;                // [S_v, ld5 >= 256]: weighted feature sums
    float *sigma;                      // [S_v]
    float slope;
    int cap_samples;
    float *dbg; int dbg_layer;         // probe: post-activation output of layer dbg_layer -> [rows, 256]
    int skew;                          // RT = 2: the second half of the grid (the CUs' second workgroups) starts skew x 64 cycles late
    // TRAINING forward (chain_kernel<4, 3>): every layer's post-activation output is kept for the backward pass
    float *H[4]; int ldh[4];           // [rows, ldh >= 256] block1.0 / block1.2 / block3.0 / block3.2 outputs; H[1] has ldh >= 264: columns 256..263 = block3's 7 extras + 0
    const int32_t *uidx;               // optional: point id -> row of ptab (the table holds the batch's touched points only)
    const int32_t *row_u; int ucap;    // chain_ws_kernel<8>: row -> row of ptab directly (>= ucap: none, row 0 is used)
    uint32_t *hbits; long long hbits_stride;   // training, optional: the signs of H[0..2] as one word per (32-row tile, wave, lane): bit 31 - i = (value i of the lane's 32 columns > 0);
                                               // layer l at hbits + l * hbits_stride words, word (((row >> 5) * 4 + wave) * 64 + lane).  The input gradients' LeakyReLU' reads these.
    unsigned *hmax;                    // [4] bit patterns of max |H[l]| (atomicMax; H[1]'s includes the extras): scales of the weight-gradient GEMMs
    unsigned *x5max;                   // optional: bit pattern of max |weighted feature sum| (the first 256 columns of X5), atomicMax
};

// Row-slot classes (hnr_chain_plan): samples [0, n_big) own 8 row slots each (16 samples per 128-row tile), the next n_small 4 (32 per tile), the
// rest -- one or two neighbours -- 2 (64 per tile); the tiles / 4-group blocks of a class follow those of the class before it in the workspace.
struct ChainClasses { int n_valid, n_big, n_small, big_tiles, small_tiles, n_tiles; };
__device__ __forceinline__ ChainClasses chain_classes(const unsigned long long *counts, int cap_samples)
{
    ChainClasses c;
    long long nv = (long long)counts[HNR_CNT_SAMPLES_VALID];
    if (nv > cap_samples) nv = cap_samples;
    long long nt = (long long)counts[HNR_CNT_SAMPLES_TINY];
    if (nt > nv) nt = nv;
    long long ns = (long long)counts[HNR_CNT_SAMPLES_SMALL];
    if (ns > nv - nt) ns = nv - nt;
    c.n_valid = (int)nv; c.n_small = (int)ns; c.n_big = (int)(nv - ns - nt);
    c.big_tiles = (c.n_big + 15) / 16;
    c.small_tiles = (int)((ns + 31) / 32);
    c.n_tiles = c.big_tiles + c.small_tiles + (int)((nt + 63) / 64);
    return c;
}
// class k of tile t (0: 8 row slots per sample, 1: 4, 2: 2); first sample of the tile; end of the class's sample range
__device__ __forceinline__ int chain_tile_class(const ChainClasses &c, int t) { return t < c.big_tiles ? 0 : (t < c.big_tiles + c.small_tiles ? 1 : 2); }
__device__ __forceinline__ int chain_tile_first(const ChainClasses &c, int t, int k)
{
    return k == 0 ? 16 * t : (k == 1 ? c.n_big + 32 * (t - c.big_tiles) : c.n_big + c.n_small + 64 * (t - c.big_tiles - c.small_tiles));
}
__device__ __forceinline__ int chain_class_end(const ChainClasses &c, int k) { return k == 0 ? c.n_big : (k == 1 ? c.n_big + c.n_small : c.n_valid); }

__device__ __forceinline__ float chain_softplus_m1(float x)
{
    const float y = __fsub_rn(x, 1.0f);                // raw2out_density: softplus(x - 1), beta = 1, threshold = 20 (:471-476)
    return y > 20.f ? y : log1pf(expf(y));
}

// csrc/chain_ws.hip: weight-stationary, software-pipelined form of the chain kernel.  mode 0: product, 1: layer dump (a.dbg, a.dbg_layer),
// 2: phase timing, 8: training forward (a.H, a.hmax, a.x5max, a.row_u).
int launch_chain_ws(const ChainArgs &a, int grid, hipStream_t st, int mode);

}  // namespace hnr
