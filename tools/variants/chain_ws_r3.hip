// Weight-stationary, software-pipelined form of the fused per-neighbour chain (same arithmetic, layouts and results as chain_kernel<4>
// in chain.hip; see the header there for the maths, the f16x2 split and the operand orientation).
//
// chain_kernel runs a 128-row tile through a layer as [MFMA loop over the k steps][epilogue], and the matrix pipe idles through every
// epilogue (bias, LeakyReLU, row scale, fp16 split, LDS publish: ~40 % of the tile time).  A second wave per SIMD cannot fill that gap
// (profiles/README.md: a wave waiting to issue its next MFMA blocks the SIMD's VALU port for the other waves), but the SAME wave can:
// VALU instructions placed between two of its MFMAs execute while the matrix pipe works on the first.  So here a wave
//   * keeps its 64 output columns' weight fragments of the WHOLE layer in registers (16 k steps x 2 column tiles x 2 planes = 256
//     AGPRs, read by the MFMAs directly; the 17th k step of block3.0 -- the 7 extra inputs -- is re-fetched per pass into VGPRs), and
//   * streams the tile's four 32-row blocks past them one after the other: a PASS (layer L, row tile rt) is S_L k steps x 6 MFMAs on
//     two accumulator tiles, and the epilogue of the previous pass's row tile is cut into small pieces issued between those MFMAs:
//     first half of the pass: bias + LeakyReLU + row maxima -> exchange buffer; one LDS barrier; second half: row scale, fp16 split,
//     publish the next layer's operand planes (for layer 3: alpha dot, then the K-sums, the X5 / sigma stores and the next tile's
//     layer-0 operand image).
// The next layer's weights are fetched into a k step's registers as soon as the last pass of the layer has used them, the per-point
// table rows of layer 0 one pass ahead.  One barrier per pass; the MFMA stream never waits for an epilogue.
#include <utility>

#include "chain_defs.h"

namespace hnr {

constexpr int CW_CST = ch_lds_exch(4) + 2048;          // LDS copy of bias[4][256], alpha_w[256], alpha_b, descale[4] (meta floats 0..1284)
constexpr int CW_CST_FLOATS = CH_META_DESCALE + 4;
constexpr int CW_DSUM = CW_CST + ((CW_CST_FLOATS * 4 + 15) & ~15);   // alpha-branch partial dot products of a tile: [4 row tiles][32 rows][4 waves]
constexpr int cw_lds_bytes() { return CW_DSUM + 4 * 32 * 4 * 4; }

// The resident weight fragments live in AGPRs that this file numbers itself: fragment (k step s, column tile c, plane p) = a[16 s + 8 c + 4 p .. +3].
// Loads into them and the MFMAs that read them are inline asm with the register numbers in the text.  (Compiler-allocated fragments -- builtin
// loads consumed through an "a" constraint -- worked but were shuffled between AGPRs through VGPRs by the register allocator: 280 copies per
// tile, transit VGPRs that pushed the kernel into scratch, and a v_accvgpr_write -> MFMA hazard the compiler cannot see around inline asm.)
// The compiler itself must not touch AGPRs in this kernel: it only would to spill, the VGPR budget below is sized so that it does not, and
// the Makefile checks the generated code for v_accvgpr / scratch instructions.  Since the compiler's s_waitcnt pass does not see these loads,
// the waits for them are explicit (cw_wait_vm); its own waits for its own loads then over-wait by the asm loads still in flight, so the
// asm loads are issued where no compiler-tracked load is about to be consumed.
#define CW_LOAD_FRAG(N_, rsrc_, voff_, soff_, IMM_) asm volatile("buffer_load_dwordx4 a[%2:%3], %0, %1, %4 offen offset:%5" :: "v"(voff_), "s"(rsrc_), "n"(N_), "n"((N_) + 3), "s"(soff_), "n"(IMM_))
template <int N> __device__ __forceinline__ void cw_mfma(f32x16 &acc, const u32x4 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, %0" : "+v"(acc) : "v"(x), "n"(N), "n"(N + 3));
}
template <int N> __device__ __forceinline__ void cw_mfma_first(f32x16 &acc, const u32x4 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, 0" : "=&v"(acc) : "v"(x), "n"(N), "n"(N + 3));
}
__device__ __forceinline__ void cw_mfma_vw(f32x16 &acc, const u32x4 &w, const u32x4 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
template <int K> __device__ __forceinline__ void cw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory"); }
// LDS traffic of this wave complete, then rendezvous -- no vmcnt wait (weight / table loads stay in flight across it)
__device__ __forceinline__ void cw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compile-time loops: the pass / k step / piece indices must be constants at every use (register arrays, asm operands), and the body is too
// large for `#pragma unroll` to accept
template <class F, int... Is> __device__ __forceinline__ void cw_static_seq(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void cw_static_for(F &&f) { cw_static_seq(f, std::make_integer_sequence<int, N>{}); }

constexpr int cw_steps(int L) { return L == 0 ? CH_S0 : L == 1 ? CH_S1 : L == 2 ? CH_S2 : CH_S3; }
constexpr int cw_wbase(int L) { return L == 0 ? CH_W0 : L == 1 ? CH_W1 : L == 2 ? CH_W2 : CH_W3; }
constexpr int cw_goff(int P) { int g = 0; for (int p = 0; p < P; ++p) g += cw_steps(p >> 2); return g; }     // k steps of the tile before pass P
constexpr int CW_GTOT = cw_goff(16);

template <int DBG>
__global__ __launch_bounds__(256, 1) void chain_ws_kernel(ChainArgs a)
{
    constexpr int SLOT = ch_slot(4), SAMPLES = 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, j = lane & 31;   // wave in an SGPR: its tests are scalar branches
    const ChainClasses cls = chain_classes(a.counts, a.cap_samples);      // tiles [0, big_tiles): 16 samples x 8 row slots; the rest: 32 samples x 4
    const int n_valid = cls.n_valid, n_tiles = cls.n_tiles;
    const float *meta = reinterpret_cast<const float *>(a.wimg + CH_META);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, CH_WBYTES, 0x00020000);
    const int col0 = 64 * wave + 16 * h;
    const unsigned woff = (unsigned)(2 * wave) * 2048u + (unsigned)lane * 16u;
    // LDS addressing: a handful of per-lane byte offsets; every access is `base register + immediate` (a ds offset field holds 16 bits, hence one
    // base per 64 KiB window).  CW_KEEP launders a base where it is used, so that the compiler folds the constant part into the instruction
    // instead of hoisting one precomputed address per use out of the tile loop (that cost ~100 registers, parked in the AGPRs this kernel owns).
    const int o_b0 = (int)lane * 16, o_b1 = o_b0 + 65536, o_b2 = o_b0 + 131072;       // fragment (s, rt, p) at s * SLOT + (rt * 2 + p) * 1024
    const int o_pub = (int)(4 * wave + h) * SLOT + (int)j * 16;                    // publish: + c * 2 SLOT + rt * 2048 + {0, 512, 1024, 1536}
    const int o_ext = 16 * SLOT + (int)j * 16;                                          // extras k step, lanes 0..31: + rt * 2048 + p * 1024
    const int o_exw = (int)ch_lds_exch(4) + (int)(j * 4 + wave) * 4, o_exr = (int)ch_lds_exch(4) + (int)j * 16;   // exchange [2][32 rows][4 waves]
    const int o_dsw = (int)CW_DSUM + (int)(j * 4 + wave) * 4, o_dsr = (int)CW_DSUM + (int)wave * 512 + (int)j * 16;    // write: + rt * 512; read: this wave's row tile
    const int o_cst = (int)CW_CST + (int)col0 * 4;                                  // constants: + layer * 1024 + c * 128 + q4 * 16
    const int o_xp = (int)(wave >> 1) * SLOT + (int)(wave & 1) * 1024 + (int)lane * 16;   // layer-0 image chunks wave, 4 + wave: + i * 2 SLOT + rt * 2048
#define CW_KEEP(x_) ({ int k_ = (x_); asm volatile("" : "+v"(k_)); k_; })
#define CW_LDS(T_, off_) (*reinterpret_cast<T_ *>(lds + (off_)))
#define CW_AT(T_, ptr_, off_) (*reinterpret_cast<T_ *>((ptr_) + (off_)))
    char *q_b0 = lds, *q_b1 = lds, *q_b2 = lds, *q_pub = lds, *q_ext = lds, *q_exw = lds, *q_exr = lds, *q_cst = lds, *q_xp = lds, *q_dsw = lds, *q_dsr = lds;      // refreshed at every pass start
#define CW_REFRESH() do { q_b0 = lds + CW_KEEP(o_b0); q_b1 = lds + CW_KEEP(o_b1); q_b2 = lds + CW_KEEP(o_b2); q_pub = lds + CW_KEEP(o_pub); q_ext = lds + CW_KEEP(o_ext); \
        q_exw = lds + CW_KEEP(o_exw); q_exr = lds + CW_KEEP(o_exr); q_cst = lds + CW_KEEP(o_cst); q_xp = lds + CW_KEEP(o_xp); q_dsw = lds + CW_KEEP(o_dsw); q_dsr = lds + CW_KEEP(o_dsr); } while (0)

    const int xcd = blockIdx.x & 7, nb = (gridDim.x + 7 - xcd) / 8, bi = blockIdx.x >> 3;
    const int per = (n_tiles + 7) / 8, t_lo = xcd * per, t_hi = (t_lo + per < n_tiles) ? t_lo + per : n_tiles;
    const bool xcd_order = gridDim.x >= 8;
    const int t_first = xcd_order ? t_lo + bi : (int)blockIdx.x, t_end = xcd_order ? t_hi : n_tiles, t_step = xcd_order ? nb : (int)gridDim.x;
    if (t_first >= t_end) return;

    for (int i = tid; i < 2 * 4 * 32; i += 256)                            // extras k step: its k = 8..15 half stays zero
        CW_LDS(u32x4, 16 * SLOT + (i >> 5) * 1024 + (32 + (i & 31)) * 16) = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < CW_CST_FLOATS; i += 256) CW_LDS(float, CW_CST + 4 * i) = meta[i];

    asm volatile("" ::: "a255");                                          // the kernel owns all 256 AGPRs (see above)
    u32x4 wx[2][2];                                                        // block3.0's 17th k step (VGPRs, per pass)
    f32x16 acc[2][2];                                                      // [row-tile parity][column tile]
    u32x4 bf[3][2];                                                        // activation fragment ring [k step % 3][plane]
    float inv[4] = {0.f, 0.f, 0.f, 0.f};
    int pid[4], pid_n[4] = {0, 0, 0, 0};
    float wq[4], wq_fin = 0.f;
    float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
    float4 tv[2][2][4];                                                    // layer 0: rows of the per-point table, two row tiles in flight ([row tile & 1][column tile][16 B])
    u32x4 xv[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    int tile_fin = -1;

    // weight fragments of (layer L, k step s) -> a[16 s ..]: four loads  (probe build -DHNR_CHAIN_WS_SAME_W=1: every k step reads the layer's
    // first one -- 64 KiB instead of 848 KiB of weights per tile from L2; results are garbage, the time shows what the weight stream costs)
#ifndef HNR_CHAIN_WS_SAME_W
#define HNR_CHAIN_WS_SAME_W 0
#endif
#define CW_LOAD_W(L_, s_) do { \
        const int so_ = cw_wbase(L_) + (HNR_CHAIN_WS_SAME_W ? 0 : (s_)) * CH_WSTEP; \
        CW_LOAD_FRAG(16 * (s_) + 0, wsrd, woff, so_, 0); CW_LOAD_FRAG(16 * (s_) + 4, wsrd, woff, so_, 1024); \
        CW_LOAD_FRAG(16 * (s_) + 8, wsrd, woff, so_, 2048); CW_LOAD_FRAG(16 * (s_) + 12, wsrd, woff, so_, 3072); } while (0)
    // per-row scalars of a tile: the point ids are needed first (table rows of layer 0: fetched into pid_n during the previous tile's last pass),
    // the aggregation weights (layer 3) and the extras (layer 1) only later: they are fetched in pass (0, 1) of their own tile
    auto load_ids = [&](int tile, int (&pd)[4]) __attribute__((always_inline)) {
        const char *aux = a.aux + (size_t)tile * 4 * CH_AUX_GROUP;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) pd[rt] = reinterpret_cast<const int32_t *>(aux + rt * CH_AUX_GROUP)[j];
    };
    auto load_rest = [&](int tile) __attribute__((always_inline)) {
        const char *aux = a.aux + (size_t)tile * 4 * CH_AUX_GROUP;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) wq[rt] = reinterpret_cast<const float *>(aux + rt * CH_AUX_GROUP + 128)[j];
        e0 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32);     // extras of the rows this wave publishes (row tile = wave)
        e1 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32 + 16);
    };
    // layer-0 operand image of (tile, row tile rt): 8 chunks of 1 KiB (k step s = c >> 1, plane p = c & 1), this wave moves chunks wave, 4 + wave
    auto xp_src = [&](int tile, int rt, int i) __attribute__((always_inline)) { return a.xp + ((size_t)tile * 4 + rt) * CH_XP_GROUP + (i * 4 + wave) * 1024 + lane * 16; };

    // ---- prologue: first tile's row scalars, layer-0 images of row tiles 0..2 (row tile 3 is staged by pass (0,0) like in every later tile),
    //      layer-0 weights
    load_ids(t_first, pid);
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int i = 0; i < 2; ++i) CW_LDS(u32x4, o_xp + i * 2 * SLOT + rt * 2048) = *reinterpret_cast<const u32x4 *>(xp_src(t_first, rt, i));
    {
        const float *trow = a.ptab + (size_t)(pid[0] < 0 ? 0 : pid[0]) * a.ldt + col0;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[0][cc][q] = *reinterpret_cast<const float4 *>(trow + 32 * cc + 4 * q);
    }
    CW_LOAD_W(0, 0); CW_LOAD_W(0, 1); CW_LOAD_W(0, 2); CW_LOAD_W(0, 3);
    cw_wait_vm<0>();
    __syncthreads();

    long long t_start = 0, w_start = 0, t_prev = 0, tm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int n_my = 0;
    if (DBG >= 2) { t_start = t_prev = clock64(); w_start = wall_clock64(); }

    // uniform scalars of the packed image (SGPRs)
    const float inv0 = __fmul_rn(pow2f(-14), meta[CH_META_DESCALE]), dw1 = meta[CH_META_DESCALE + 1], dw2 = meta[CH_META_DESCALE + 2], dw3 = meta[CH_META_DESCALE + 3],
                alpha_b = meta[4 * 256 + 256];
    float amax = 0.f, ap = 0.f, sc_run = 1.f;                               // epilogue state carried between the pieces of one pass
    f32x2 bias_c = {0.f, 0.f}, bias_n = {0.f, 0.f}, aw_c = {0.f, 0.f};     // constants of the running item (read from LDS one micro-stage ahead)
    unsigned ph[8], pm[8];
    float s1x = 0.f, s1y = 0.f, s1a = 0.f, s1b = 0.f;                       // values handed from an item's first micro-stage to its second
    unsigned s1h = 0u;
    // X5 rows of the row tile whose sums are being stored: a buffer descriptor on its first sample (uniform) + ONE per-lane byte offset, out of
    // range (the store is dropped) in the lanes that hold no sum; x5_flag = 1 / 0: the third step of the K-sum runs / is a no-op (4-slot samples)
    __amdgpu_buffer_rsrc_t x5_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.X5), 0, 0, 0x00020000);
    int x5_voff = 0x40000000;
    float x5_flag = 1.f, x5_flag2 = 1.f;                                    // (x5_flag2: the second step, a no-op for 2-slot samples)
    float ks0 = 0.f, ks1 = 0.f, wq_sig = 0.f;                               // wq_sig: this wave's own row tile (= wave) of the tile whose densities are due
    float4 ex4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- epilogue of pass PP = (PL, PR), accumulator set se, cut into MICRO-STAGES of 3..8 VALU instructions.  A wave issues in order, and a
    //      dependent VALU instruction issues 8 cycles after its producer, an independent one 4: so every micro-stage holds two independent
    //      chains (the x and the y value of an item), an item's chain is cut in two stages that land behind different MFMAs, and LDS reads
    //      are issued a stage or more before their first use.  `slot` counts the 6 S MFMAs of the RUNNING pass, H = first slot of its second
    //      half; micro-stages [slot * MS / H, (slot + 1) * MS / H) run behind MFMA `slot`.  All arguments are constants after unrolling.
    auto epilogue_piece = [&](int PL, int PR, int S, int slot, int tile_e, bool stage_next, int tile_stage) __attribute__((always_inline)) {
        const int T = 6 * S, H = T / 2, se = PR & 1;
        const int exb = (PR & 1) * 512;
        const float inv_l = PL == 0 ? inv0 : inv[PR];
        // bias pair of item `it` (and for layer 3 its alpha weights): 8 B each from the LDS copy of the constants
        auto read_bias = [&](int it) __attribute__((always_inline)) {
            return (DBG == 5 || DBG == 7) ? f32x2{a.slope, 1.f} : CW_AT(const f32x2, q_cst, PL * 1024 + (it >> 3) * 128 + (it & 7) * 8);
        };
        auto read_aw = [&](int it) __attribute__((always_inline)) {
            return (DBG == 5 || DBG == 7) ? f32x2{a.slope, 1.f} : CW_AT(const f32x2, q_cst, 4 * 1024 + (it >> 3) * 128 + (it & 7) * 8);
        };
        if (slot < H) {
            // -- first half.  micro-stage 0: bias of item 0; 1 + 2 i, 2 + 2 i: item i = (c, q), two values each (the first stage also asks for
            //    the next item's bias and, in layer 3, this item's alpha weights); 33: row maximum / alpha partial -> exchange buffer
            const int MS = 34, m0 = slot * MS / H, m1 = (slot + 1) * MS / H;
#pragma unroll
            for (int ms = m0; ms < m1; ++ms) {
                if (ms == 0) {
                    amax = 0.f; ap = 0.f;
                    bias_n = read_bias(0);
                } else if (ms < 33) {
                    const int it = (ms - 1) >> 1, st = (ms - 1) & 1, c = it >> 3, q = it & 7;
                    if (st == 0) {
                        // scalar fp32 VALU on purpose: packed fp32 instructions (v_pk_fma_f32 ...) do not overlap with this wave's MFMAs -- one
                        // of them behind an MFMA costs 18 cycles of matrix-pipe time, a v_fma_f32 none (tools/interleave_probe.hip)
                        bias_c = bias_n;
                        if (it + 1 < 16) bias_n = read_bias(it + 1);
                        if (PL == 3) aw_c = read_aw(it);
                        float ax = bias_c.x, ay = bias_c.y;
                        if (PL == 0) { const float4 t4 = tv[PR & 1][c][q >> 1]; ax = __fadd_rn(ax, (q & 1) ? t4.z : t4.x); ay = __fadd_rn(ay, (q & 1) ? t4.w : t4.y); }
                        s1x = fmaf(acc[se][c][2 * q], inv_l, ax); s1y = fmaf(acc[se][c][2 * q + 1], inv_l, ay);
                        s1a = __fmul_rn(s1x, a.slope); s1b = __fmul_rn(s1y, a.slope);
                    } else {
                        const float vx = fmaxf(s1x, s1a), vy = fmaxf(s1y, s1b);
                        acc[se][c][2 * q] = vx; acc[se][c][2 * q + 1] = vy;
                        if (PL == 3) { ap = fmaf(vx, aw_c.x, ap); ap = fmaf(vy, aw_c.y, ap); }
                        else amax = fmaxf(fmaxf(amax, fabsf(vx)), fabsf(vy));
                        if (DBG == 1) {
                            if (a.dbg && a.dbg_layer == PL && tile_e >= 0) {
                                int te = tile_e;                                        // laundered: no 64-bit induction variable
                                asm volatile("" : "+s"(te));
                                float *o = a.dbg + ((size_t)te * 128 + 32 * PR + j) * 256 + col0 + 32 * c + 2 * q;
                                o[0] = vx; o[1] = vy;
                            }
                        }
                    }
                } else {
                    float m;
                    if (PL == 3) { m = __fadd_rn(ap, __shfl_xor(ap, 32)); if (h == 0) CW_AT(float, q_dsw, PR * 512) = m; }
                    else {
                        m = fmaxf(amax, __shfl_xor(amax, 32));
                        if (PL == 1 && PR == wave)
                            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(e0.x), fabsf(e0.y)), fmaxf(fabsf(e0.z), fabsf(e0.w))), fmaxf(fmaxf(fabsf(e1.x), fabsf(e1.y)), fabsf(e1.z))));
                    }
                    if (PL != 3 && h == 0 && DBG != 6 && DBG != 7) CW_AT(float, q_exw, exb) = m;
                }
            }
            return;
        }
        const int k2 = slot - H, N2 = T - H;
        if (PL != 3) {
            // -- second half, hidden layers: micro-stage 0: exchange read; 1: row scale; 2 + 2 i, 3 + 2 i: item i: scale + fp16 high parts, then
            //    residuals + low parts (+ the operand-plane stores after q = 3, 7); 34: the 7 extra inputs of block3.0
            const int MS = 35, m0 = k2 * MS / N2, m1 = (k2 + 1) * MS / N2;
#pragma unroll
            for (int ms = m0; ms < m1; ++ms) {
                if (ms == 0) ex4 = (DBG == 5 || DBG == 7) ? make_float4(amax, 1.f, 2.f, 3.f) : CW_AT(const float4, q_exr, exb);
                else if (ms == 1) {
                    const int k = row_scale_exp(fmaxf(fmaxf(ex4.x, ex4.y), fmaxf(ex4.z, ex4.w)));
                    sc_run = pow2f(k);
                    inv[PR] = __fmul_rn(pow2f(-k), PL == 0 ? dw1 : PL == 1 ? dw2 : dw3);
                } else if (ms < 34) {
                    const int it = (ms - 2) >> 1, st = (ms - 2) & 1, c = it >> 3, q = it & 7;
                    if (st == 0) {
                        s1x = __fmul_rn(acc[se][c][2 * q], sc_run); s1y = __fmul_rn(acc[se][c][2 * q + 1], sc_run);
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(s1h) : "v"(s1x), "v"(s1y));
                    } else {
                        float r0, r1;
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(s1h), "v"(s1x));
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(s1h), "v"(s1y));
                        ph[q] = s1h;
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pm[q]) : "v"(r0), "v"(r1));
                        if ((q == 3 || q == 7) && (DBG == 6 || DBG == 7)) { asm volatile("" :: "v"(ph[q]), "v"(pm[q]), "v"(ph[q - 1]), "v"(pm[q - 1]), "v"(ph[q - 2]), "v"(pm[q - 2]), "v"(ph[q - 3]), "v"(pm[q - 3])); }
                        else if (q == 3 || q == 7) {
                            const int dst = c * 2 * SLOT + PR * 2048 + (q == 7 ? 512 : 0);
                            CW_AT(u32x4, q_pub, dst) = u32x4{ph[q - 3], ph[q - 2], ph[q - 1], ph[q]};
                            CW_AT(u32x4, q_pub, dst + 1024) = u32x4{pm[q - 3], pm[q - 2], pm[q - 1], pm[q]};
                        }
                    }
                } else if (PL == 1 && PR == wave && h == 0) {
                    unsigned xh[4], xm[4];
                    split2h(__fmul_rn(e0.x, sc_run), __fmul_rn(e0.y, sc_run), xh[0], xm[0]);
                    split2h(__fmul_rn(e0.z, sc_run), __fmul_rn(e0.w, sc_run), xh[1], xm[1]);
                    split2h(__fmul_rn(e1.x, sc_run), __fmul_rn(e1.y, sc_run), xh[2], xm[2]);
                    split2h(__fmul_rn(e1.z, sc_run), 0.f, xh[3], xm[3]);
                    CW_AT(u32x4, q_ext, PR * 2048) = u32x4{xh[0], xh[1], xh[2], xh[3]};
                    CW_AT(u32x4, q_ext, PR * 2048 + 1024) = u32x4{xm[0], xm[1], xm[2], xm[3]};
                }
            }
            return;
        }
        // -- second half, layer 3: K-weighted sums (8 adjacent lanes: three DPP steps), X5 / sigma stores, next tile's layer-0 image of row tile PR.
        //    micro-stage 0: exchange read; 1 + i: value pair i (16 pairs; a float4 store after every second pair); 17: sigma; 18: image
        const float wq_e = PR == 3 ? wq_fin : wq[PR];
        // a tile of the second / third slot class (hnr_chain_plan): 8 samples of 4 row slots / 16 samples of 2 per row tile -- the sum stops after two
        // DPP steps / one
        const int kc_e = tile_e < 0 ? 0 : chain_tile_class(cls, tile_e);
        const int MS = 19, m0 = k2 * MS / N2, m1 = (k2 + 1) * MS / N2;
#pragma unroll
        for (int ms = m0; ms < m1; ++ms) {
            if (ms == 0) {
                if (PR == 3) ex4 = CW_AT(const float4, q_dsr, 0);
                // where this row tile's sums go (once per pass: per store it cost a 64-bit multiply, a three-way select and an exec-masked branch)
                const int row0 = tile_e < 0 ? n_valid : chain_tile_first(cls, tile_e, kc_e) + (4 << kc_e) * PR;
                const int ls = j >> (3 - kc_e);
                const bool st = (j & ((8 >> kc_e) - 1)) == 0 && row0 + ls < (tile_e < 0 ? n_valid : chain_class_end(cls, kc_e));
                x5_voff = st ? (ls * a.ld5 + col0) * 4 : 0x40000000;
                x5_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.X5) + (size_t)row0 * a.ld5 * 4, 0, 16 * a.ld5 * 4, 0x00020000);
                x5_flag = kc_e > 0 ? 0.f : 1.f;
                x5_flag2 = kc_e > 1 ? 0.f : 1.f;
            } else if (ms < 17) {
                const int pr = ms - 1, c = pr >> 3, e0i = 2 * (pr & 7);
                // the sum over a sample's 8 (4, 2) row slots = 8 (4, 2) adjacent lanes: pair swap, quad-pair swap, half-row mirror; the second and
                // third step as f += dpp(f) * flag (exactly f + dpp(f) or f).  One block: every DPP read sits two wait states after the write it reads
                // (the assembler does not pad inline asm), and the compiler cannot expand the steps into moves / selects / branches
                float f0, f1;
                asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4\n\ts_nop 0\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                             "v_fmac_f32_dpp %0, %0, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %1, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
                             "v_fmac_f32_dpp %0, %0, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %1, %5 row_half_mirror row_mask:0xf bank_mask:0xf"
                             : "=&v"(f0), "=&v"(f1) : "v"(acc[se][c][e0i]), "v"(acc[se][c][e0i + 1]), "v"(wq_e), "v"(x5_flag), "v"(x5_flag2));
                if ((pr & 1) == 0) { ks0 = f0; ks1 = f1; }
                else __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(ks0), __float_as_uint(ks1), __float_as_uint(f0), __float_as_uint(f1)}, x5_rs,
                                                            x5_voff + (32 * c + 2 * (pr & 6)) * 4, 0, 0);
            } else if (ms == 17) {
                // the tile's densities, once per tile and in all four waves at the same time (softplus is ~100 instructions that cannot be cut
                // into pieces; per row tile it stalled a different pass for each wave): wave w takes row tile w
                if (PR == 3) {
                    const int s_sig = tile_e < 0 ? n_valid : chain_tile_first(cls, tile_e, kc_e) + (4 << kc_e) * wave + (j >> (3 - kc_e));
                    const float d = __fadd_rn(__fadd_rn(ex4.x, ex4.y), __fadd_rn(ex4.z, ex4.w));
                    float sg = __fmul_rn(chain_softplus_m1(__fadd_rn(d, alpha_b)), wq_sig);
                    sg = __fadd_rn(sg, __builtin_amdgcn_update_dpp(0.f, sg, 0xB1, 0xf, 0xf, false));
                    { const float g = __builtin_amdgcn_update_dpp(0.f, sg, 0x4E, 0xf, 0xf, false); sg = __fadd_rn(sg, kc_e > 1 ? 0.f : g); }
                    { const float g = __builtin_amdgcn_update_dpp(0.f, sg, 0x141, 0xf, 0xf, false); sg = __fadd_rn(sg, kc_e > 0 ? 0.f : g); }
                    if (h == 0 && (j & ((8 >> kc_e) - 1)) == 0 && s_sig < (tile_e < 0 ? 0 : chain_class_end(cls, kc_e))) a.sigma[s_sig] = sg;
                }
            } else if (stage_next && tile_stage < t_end) {
#pragma unroll
                for (int i = 0; i < 2; ++i) CW_AT(u32x4, q_xp, i * 2 * SLOT + PR * 2048) = xv[i];
            }
        }
    };

    auto b_read = [&](int rt, int s, int p) __attribute__((always_inline)) {
        const int off = s * SLOT + (rt * 2 + p) * 1024;
        return off < 65536 ? CW_AT(const u32x4, q_b0, off) : off < 131072 ? CW_AT(const u32x4, q_b1, off - 65536) : CW_AT(const u32x4, q_b2, off - 131072);
    };

    for (int tile = t_first; tile < t_end; tile += t_step) {
        ++n_my;
        const int tile_next = tile + t_step;
        CW_REFRESH();
        // k steps 0, 1 of pass (0,0)
        bf[0][0] = b_read(0, 0, 0); bf[0][1] = b_read(0, 0, 1);
        bf[1][0] = b_read(0, 1, 0); bf[1][1] = b_read(0, 1, 1);
        cw_static_for<16>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value;
            constexpr int L = P >> 2, rt = P & 3, S = cw_steps(L), G0 = cw_goff(P);
            constexpr int PP = (P + 15) & 15, PL = PP >> 2, PR = PP & 3, sm = rt & 1;
            constexpr int T = 6 * S, H = T / 2;
            const int tile_e = P == 0 ? tile_fin : tile;                   // tile of the row tile whose epilogue runs here
            const int tile_stage = P == 0 ? tile : tile_next;              // tile whose layer-0 image a layer-3 epilogue stages
            if (DBG >= 2) { const long long t_ = clock64(); tm[(P + 15) & 15] += t_ - t_prev; t_prev = t_; }
            if (P != 0) CW_REFRESH();
            // ---- pass start: loads that ride ahead
            if (L == 2) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        wx[c][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + (c * 2 + p) * 1024, CH_W2 + 16 * CH_WSTEP, 0));
            }
            if (PL == 3 && tile_stage < t_end) {
#pragma unroll
                for (int i = 0; i < 2; ++i) xv[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xp_src(tile_stage, PR, i)));      // read once
            }
            if (P == 15 && tile_next < t_end) load_ids(tile_next, pid_n);
            if (P == 1) load_rest(tile);                                   // (pass (0, 0) still reads the previous tile's weight of row tile 3: wq_fin)
            __builtin_amdgcn_sched_barrier(0);
            cw_static_for<6 * S>([&](auto kc) __attribute__((always_inline)) {
                // one piece per MFMA: slot = 6 s + 2 g + c (k step s, term g, column tile c)
                constexpr int slot = decltype(kc)::value, s = slot / 6, g = (slot % 6) >> 1, c = slot & 1;
                if (slot == H) {
                    cw_lds_barrier();
                    // layer 0's table rows are asked for one and a half passes before their epilogue: row tile rt + 1 at the barrier of pass
                    // (0, rt) -- the set it goes into was consumed in this pass's first half --, the next tile's row tile 0 at the barrier of (3, 3)
                    if ((L == 0 && rt < 3) || (L == 3 && rt == 3)) {
                        const int pr = L == 0 ? pid[rt < 3 ? rt + 1 : 0] : pid_n[0];
                        if (L == 0 || tile_next < t_end) {
                            const float *trow = a.ptab + (size_t)(pr < 0 ? 0 : pr) * a.ldt + col0;
#pragma unroll
                            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                                for (int q = 0; q < 4; ++q) tv[L == 0 ? (rt + 1) & 1 : 0][cc][q] = *reinterpret_cast<const float4 *>(trow + 32 * cc + 4 * q);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int wp = g == 0 ? 1 : 0, xp_ = g == 1 ? 1 : 0;   // wm*xh, wh*xm, wh*xh: smallest terms first
                constexpr int ring = (G0 + s) % 3;
                // first pass of a layer: the k step's fragments were fetched during the previous layer's last pass; `after` = k steps fetched after it
                if constexpr (rt == 0 && g == 0 && c == 0 && s < 16) {
                    constexpr int after = L == 1 ? (s < 4 ? 3 - s : 15 - s + 4) : L == 0 ? 3 - s : 15 - s;
                    cw_wait_vm<(4 * after < 60 ? 4 * after : 60)>();
                }
                constexpr int A0 = 16 * (s < 16 ? s : 0) + 4 * wp + 8 * c;
                if constexpr (DBG == 4) { if (slot < 2) { acc[sm][c] = f32x16{} + bf[ring][xp_][0]; } }
                else if constexpr (L == 2 && s == 16) cw_mfma_vw(acc[sm][c], wx[c][wp], bf[ring][xp_]);
                else if constexpr (slot < 2) cw_mfma_first<A0>(acc[sm][c], bf[ring][xp_]);
                else cw_mfma<A0>(acc[sm][c], bf[ring][xp_]);
                __builtin_amdgcn_sched_barrier(0);                          // nothing of the epilogue is hoisted above the MFMA (an accumulator is read two MFMAs after its last writer at the earliest)
                // activation fragments two k steps ahead (possibly the next pass's)
                if constexpr (g < 2 && c == 0 && G0 + s + 2 < CW_GTOT) {
                    constexpr int s2 = s + 2 >= S ? s + 2 - S : s + 2, P2 = s + 2 >= S ? P + 1 : P;
                    if constexpr (P2 < 16 && s2 < cw_steps((P2 & 15) >> 2)) bf[(G0 + s + 2) % 3][g] = b_read(P2 & 3, s2, g);
                }
                // the next layer's weights move into a k step's registers once the layer's last pass has used them; during layer 0 (4 k steps
                // resident) k steps 4..15 of block1.2 are fetched early in the pass, before the pass's table rows are asked for
                if constexpr (g == 2 && c == 1 && rt == 3 && s < 16 && s < cw_steps((L + 1) & 3)) {
                    if constexpr (L == 3) { if (tile_next < t_end) CW_LOAD_W(0, s); }
                    else CW_LOAD_W(L + 1, s);
                }
                if constexpr (L == 0 && rt < 3 && slot >= 2 && slot <= 8 && (slot & 1) == 0) CW_LOAD_W(1, 4 + 4 * rt + (slot / 2 - 1));
                if (DBG != 3) epilogue_piece(PL, PR, S, slot, tile_e, true, tile_stage);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // tile switch
        wq_fin = wq[3]; tile_fin = tile;
        wq_sig = wave == 0 ? wq[0] : wave == 1 ? wq[1] : wave == 2 ? wq[2] : wq[3];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) pid[rt] = pid_n[rt];
    }
    // ---- drain: the last tile's layer-3 epilogue of row tile 3
    CW_REFRESH();
    cw_static_for<6 * CH_S0>([&](auto kc) __attribute__((always_inline)) {
        constexpr int slot = decltype(kc)::value;
        if (slot == (6 * CH_S0) / 2) cw_lds_barrier();
        epilogue_piece(3, 3, CH_S0, slot, tile_fin, false, 0);
    });
    if (DBG >= 2 && blockIdx.x == 0 && lane == 0 && a.dbg) {               // block 0: cycles per pass [wave][16]
        long long *o = reinterpret_cast<long long *>(a.dbg) + 4 * 1024 + wave * 16;
        for (int i = 0; i < 16; ++i) o[i] = tm[i];
    }
    if (DBG >= 2 && tid == 0 && a.dbg) {                                   // every block: {cycles, wall ticks, tiles}
        long long *o = reinterpret_cast<long long *>(a.dbg) + 4 * (size_t)blockIdx.x;
        o[0] = clock64() - t_start; o[1] = wall_clock64() - w_start; o[2] = n_my; o[3] = 0;
    }
}

int launch_chain_ws(const ChainArgs &a, int grid, hipStream_t st, int mode)
{
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
#ifdef HNR_CHAIN_WS_PROBES
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
#endif
    }
#ifdef HNR_CHAIN_WS_PROBES                                                       // timing probes (results are garbage): make EXTRA=-DHNR_CHAIN_WS_PROBES
    if (mode == 5) chain_ws_kernel<5><<<grid, 256, cw_lds_bytes(), st>>>(a);           // epilogue without its LDS reads
    else if (mode == 6) chain_ws_kernel<6><<<grid, 256, cw_lds_bytes(), st>>>(a);      // epilogue without its LDS writes
    else if (mode == 7) chain_ws_kernel<7><<<grid, 256, cw_lds_bytes(), st>>>(a);      // neither
    else if (mode == 3) chain_ws_kernel<3><<<grid, 256, cw_lds_bytes(), st>>>(a);      // no epilogue
    else if (mode == 4) chain_ws_kernel<4><<<grid, 256, cw_lds_bytes(), st>>>(a);      // no MFMAs
    else
#endif
    if (mode == 2) chain_ws_kernel<2><<<grid, 256, cw_lds_bytes(), st>>>(a);
    else if (mode == 1) chain_ws_kernel<1><<<grid, 256, cw_lds_bytes(), st>>>(a);
    else chain_ws_kernel<0><<<grid, 256, cw_lds_bytes(), st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

}  // namespace hnr
