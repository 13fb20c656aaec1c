// Weight-stationary, software-pipelined colour-feature MLP: color_feature_branch (280 -> 128 -> 128 -> 128, LeakyReLU) + the tail 128 -> 64 (the
// colour-feature columns of aux_merge_weight_block.0 with that layer's bias, no activation)          models/aggregators/point_aggregators.py:1028-1037, :1199
// -- the (18, 8, 8, 8) k-step instance of hnr_mlp3_forward, same packed weight image, same arithmetic (f16x2: hnr_h2.h), bit-identical rows.
//
// mlp3_kernel streams every layer's weight fragments L2 -> registers per 64-row tile (307 KiB per tile: the CU's vector-memory front end was 60 % busy
// with that alone) and runs the phases of a tile one after the other (row load + split, MFMAs, epilogue, barriers; the matrix pipe 31 % busy).  Here, as
// in csrc/chain_ws.hip:
//   * a wave keeps ALL FOUR layers' fragments of its 32 output columns in registers for the whole launch: 18 + 8 + 6 k steps x 2 planes in 256
//     self-numbered AGPRs, the last two k steps of layer 2 and the tail's eight in 80 VGPRs -- nothing but the rows themselves is fetched per tile;
//   * a 64-row tile is two 32-row tiles that go through the layers alternately (pass P = (layer P >> 1, row tile P & 1): one accumulator chain of
//     3 S MFMAs); the epilogue of pass P - 1 (bias, LeakyReLU, row maxima | barrier | row scale, fp16 split, operand planes of the next layer) is cut into
//     pieces issued between the MFMAs of pass P, and so is the conversion of the NEXT tile's input rows (row maximum, scale, fp16 split into the
//     layer-0 planes) in the two long layer-0 passes: row tile 0's during pass (0,1) -- whose MFMAs no longer read that half of the planes --, row
//     tile 1's during the next tile's pass (0,0).  The rows arrive long before: row tile 0's by DMA into a raw LDS buffer (asked for during pass (1,0) of
//     the tile BEFORE: a whole tile ahead, no registers), row tile 1's in 40 staging VGPRs (asked for at the end of pass (0,1), converted seven passes
//     later).  Two barriers per pass (a pass's output planes are read by the very next pass of the other row tile's successor layer); nothing
//     inside a pass waits for memory that was asked for less than a pass ago.
#include <utility>

#include "cf_ws.h"

namespace hnr {

constexpr int CF_S[4] = {18, 8, 8, 8};
constexpr int CF_WSTEP = 8192;                                              // weight image bytes per k step: [column tile 4][plane 2][64 lanes][16 B] (hnr_mlp3_pack)
constexpr int CF_SLOT = 2 * 2048 + 32;                                      // LDS bytes per k step of operand planes: [row tile 2][plane 2][64 lanes][16 B] + the pad of mlp3_kernel
constexpr int CF_H = 18 * CF_SLOT;                                          // hidden layers' operand planes (8 k steps) behind layer 0's (18)
constexpr int CF_EXCH = CF_H + 8 * CF_SLOT;                                 // float [64 rows][4 waves]: row maxima of a layer's four column tiles
constexpr int CF_RINV = CF_EXCH + 64 * 4 * 4;                               // float [64]: 2^-k of the input rows' scales
constexpr int CF_CST = CF_RINV + 64 * 4;                                    // bias[4][128], descale[4]
constexpr int CF_CST_FLOATS = 4 * 128 + 4;
constexpr int CF_RAW = CF_CST + ((CF_CST_FLOATS * 4 + 15) & ~15);           // raw fp32 rows of the NEXT tile's row tile 0 (DMA): [wave 4][batch 2][burst 5][64 lanes][16 B]
constexpr int CF_DUMMY = CF_RAW + 4 * 2 * 5 * 1024;                          // 2 x 1 KiB strip for the plane stores of the columns past 288 (never read)
constexpr int cf_lds_bytes() { return CF_DUMMY + 2048; }

// when the next rows are asked for, one burst at a time: row tile 1 of the next tile (-> staging registers, free once pass (0,1) has converted the raw
// buffer through them) behind MFMAs of passes (0,1) .. (1,1); row tile 0 of the tile after next (-> raw buffer) behind MFMAs of passes (2,0) .. (3,1)
constexpr int cf_ld_index(int P, int slot) { return P == 1 ? (slot == 49 ? 0 : slot == 52 ? 1 : -1) : (P == 2 || P == 3) ? ((slot == 3 || slot == 8 || slot == 13 || slot == 18) ? 2 + (P - 2) * 4 + (slot - 3) / 5 : -1) : -1; }
constexpr int cf_dma_index(int P, int slot) { return (P >= 4 && (slot == 2 || slot == 8 || slot == 14)) ? ((P - 4) * 3 + (slot - 2) / 6 < 10 ? (P - 4) * 3 + (slot - 2) / 6 : -1) : -1; }
constexpr int cf_goff(int P) { int g = 0; for (int p = 0; p < P; ++p) g += CF_S[p >> 1]; return g; }      // iterations of the tile before pass P
// resident fragment (layer L, k step s, plane p): AGPR a[N .. N + 3] with N = cf_areg, or -1: in VGPRs (layer 2's k steps 6, 7; the tail)
constexpr int cf_areg(int L, int s, int p) { return L == 0 ? 8 * s + 4 * p : L == 1 ? 144 + 8 * s + 4 * p : (L == 2 && s < 6) ? 208 + 8 * s + 4 * p : -1; }

template <int N, int IMM> __device__ __forceinline__ void cf_load_frag(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff)
{
    asm volatile("buffer_load_dwordx4 a[%2:%3], %0, %1, %4 offen offset:%5" :: "v"(voff), "s"(rsrc), "n"(N), "n"(N + 3), "s"(soff), "n"(IMM));
}
template <int N> __device__ __forceinline__ void cf_mfma_a(f32x16 &acc, const u32x4 &x) { asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, %0" : "+v"(acc) : "v"(x), "n"(N), "n"(N + 3)); }
template <int N> __device__ __forceinline__ void cf_mfma_a_first(f32x16 &acc, const u32x4 &x) { asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, 0" : "=&v"(acc) : "v"(x), "n"(N), "n"(N + 3)); }
__device__ __forceinline__ void cf_mfma_v(f32x16 &acc, const u32x4 &w, const u32x4 &x) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x)); }
__device__ __forceinline__ void cf_mfma_v_first(f32x16 &acc, const u32x4 &w, const u32x4 &x) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x)); }
__device__ __forceinline__ void cf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define CF_KEEP(x_) ({ int k_ = (x_); asm volatile("" : "+v"(k_)); k_; })
template <class F, int... Is> __device__ __forceinline__ void cf_static_seq(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void cf_static_for(F &&f) { cf_static_seq(f, std::make_integer_sequence<int, N>{}); }

#ifdef HNR_CF_PROBE                                                           // probe build (make EXTRA=-DHNR_CF_PROBE): cycles per pass of workgroup 0, wave 0
__device__ long long g_cf_probe[12];
#endif
__global__ __launch_bounds__(256, 1) void cf_ws_kernel(CfWsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, j = lane & 31;
    long long M = a.M_cap;
    if (a.counts) { const long long c = (long long)a.counts[a.count_index]; if (c < M) M = c; }
    const int n_tiles = (int)((M + 63) / 64);
    if ((int)blockIdx.x >= n_tiles) return;
    const float *meta = reinterpret_cast<const float *>(a.wimg + (size_t)(18 + 8 + 8 + 8) * CF_WSTEP);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, (18 + 8 + 8 + 8) * CF_WSTEP, 0x00020000);
    const int col0 = 32 * wave + 16 * h;                                    // this lane's 16 output columns: col0 + r
    const unsigned woff = (unsigned)wave * 2048u + (unsigned)lane * 16u;    // column tile `wave` of every k step of the image
    for (int i = tid; i < CF_CST_FLOATS; i += 256) *reinterpret_cast<float *>(lds + CF_CST + 4 * i) = meta[i];

    // ---- resident weights
    asm volatile("" ::: "a255");                                            // the kernel owns all 256 AGPRs: the compiler must not use them (checked on the generated code)
    const int wb0 = a.wbase[0], wb1 = a.wbase[1], wb2 = a.wbase[2];
    cf_static_for<18>([&](auto sc) __attribute__((always_inline)) { constexpr int s = decltype(sc)::value; cf_load_frag<cf_areg(0, s, 0), 0>(wsrd, woff, wb0 + s * CF_WSTEP); cf_load_frag<cf_areg(0, s, 1), 1024>(wsrd, woff, wb0 + s * CF_WSTEP); });
    cf_static_for<8>([&](auto sc) __attribute__((always_inline)) { constexpr int s = decltype(sc)::value; cf_load_frag<cf_areg(1, s, 0), 0>(wsrd, woff, wb1 + s * CF_WSTEP); cf_load_frag<cf_areg(1, s, 1), 1024>(wsrd, woff, wb1 + s * CF_WSTEP); });
    cf_static_for<6>([&](auto sc) __attribute__((always_inline)) { constexpr int s = decltype(sc)::value; cf_load_frag<cf_areg(2, s, 0), 0>(wsrd, woff, wb2 + s * CF_WSTEP); cf_load_frag<cf_areg(2, s, 1), 1024>(wsrd, woff, wb2 + s * CF_WSTEP); });
    u32x4 w2v[2][2], wtv[8][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w2v[s][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + p * 1024, a.wbase[2] + (6 + s) * CF_WSTEP, 0));
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) wtv[s][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + p * 1024, a.wbase[3] + s * CF_WSTEP, 0));

    // LDS addressing: per-lane byte offsets laundered where they are used (csrc/chain_ws.hip: otherwise one hoisted address register per use)
    const int o_b = lane * 16;                                              // fragment (region, s, rt, p) at region + s * SLOT + (rt * 2 + p) * 1024
    const int o_pub = CF_H + (2 * wave + h) * CF_SLOT + j * 16;             // publish: + rt * 2048 + {0, 512, 1024, 1536}
    const int o_exw = CF_EXCH + (j * 4 + wave) * 4, o_exr = CF_EXCH + j * 16;      // + rt * 512
    const int o_cst = CF_CST + col0 * 4;                                    // + layer * 512 + q4 * 16
#define CF_AT(T_, ptr_, off_) (*reinterpret_cast<T_ *>((ptr_) + (off_)))
    char *q_b = lds, *q_pub = lds, *q_exw = lds, *q_exr = lds, *q_cst = lds;
#define CF_REFRESH() do { q_b = lds + CF_KEEP(o_b); q_pub = lds + CF_KEEP(o_pub); q_exw = lds + CF_KEEP(o_exw); q_exr = lds + CF_KEEP(o_exr); q_cst = lds + CF_KEEP(o_cst); } while (0)

    // ---- input rows.  A row tile's 32 rows are converted by the four waves, 8 rows each, two batches of 4 rows (16 lanes per row, five 16-B bursts:
    //      the prologue of mlp3_kernel): row maximum over its 16 lanes by DPP, power-of-two scale, fp16 split, 8-byte plane stores.
    const int lr0 = lane & 15, sub0 = lane >> 4;
    float4 stg[2][5];                                                       // the staged rows of a row tile 1
    auto row_ptr = [&](int tile, int rt, int b) __attribute__((always_inline)) {
        const int sub = CF_KEEP(sub0);
        long long row = (long long)tile * 64 + 32 * rt + 8 * wave + 4 * b + sub;
        if (row >= M) row = M - 1;
        return a.A + (size_t)row * a.lda;
    };
    // one 16-B burst (batch b, burst nb) of a row tile: into stg (row tile 1) / by DMA into the raw LDS buffer (row tile 0: every lane's 16 B land
    // where it will read them back).  Issued ONE at a time between MFMAs: ten vector-memory instructions back to back cost the wave ~150 cycles each.
    auto load_burst = [&](int tile, int b, int nb) __attribute__((always_inline)) {
        const int lr = CF_KEEP(lr0), c = 4 * (nb * 16 + lr);
        stg[b][nb] = *reinterpret_cast<const float4 *>(row_ptr(tile, 1, b) + (c + 4 <= a.lda ? c : 0));
    };
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>(lds);
    auto dma_burst = [&](int tile, int b, int nb) __attribute__((always_inline)) {
        const int lr = CF_KEEP(lr0), c = 4 * (nb * 16 + lr);
        const float *p = row_ptr(tile, 0, b) + (c + 4 <= a.lda ? c : 0);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(lds_base + (unsigned)(CF_RAW + ((wave * 2 + b) * 5 + nb) * 1024)) : "memory", "m0");
    };
    auto load_rows = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i) load_burst(tile, i / 5, i % 5);
    };
    auto dma_rows = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i) dma_burst(tile, i / 5, i % 5);
    };
    float cv_sc[2] = {1.f, 1.f};
    int cv_base = 0, cv_base4 = 0;                                          // plane-store addresses of the running conversion (batch 0; batch 1: + 64)
    // conversion pieces of row tile rt (rt = 1: from stg; rt = 0: from the raw LDS buffer): piece 2 b (b = 0, 1): maximum + scale of batch b;
    // pieces 4 + 5 b + nb: split + store of burst nb.  Same values as mlp3_kernel's prologue (the launcher guarantees 256 < K0 <= 288: only the last
    // burst holds columns past K0; the scale multiplication rides in the fused multiply-convert: exact, a power of two).
    auto convert_piece = [&](int rt, int piece) __attribute__((always_inline)) {
        const int lr = CF_KEEP(lr0), sub = CF_KEEP(sub0);                   // laundered: their multiples / LDS addresses are recomputed here, not kept in registers across the tile
        if (piece < 4) {
            if (piece & 1) return;
            const int b = piece >> 1;
            if (rt == 0) {
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) stg[b][nb] = *reinterpret_cast<const float4 *>(lds + CF_RAW + ((wave * 2 + b) * 5 + nb) * 1024 + lane * 16);
            }
            {   // columns 256 + 4 lr + e of the last burst: zero from K0 on
                float *t = reinterpret_cast<float *>(&stg[b][4]);
                const int left = a.K0 - 256 - 4 * lr;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = e < left ? t[e] : 0.f;
            }
            float m = 0.f;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) m = fmaxf(fmaxf(fmaxf(m, fabsf(stg[b][nb].x)), fmaxf(fabsf(stg[b][nb].y), fabsf(stg[b][nb].z))), fabsf(stg[b][nb].w));
            m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xf, 0xf, false));
            m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xf, 0xf, false));
            m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x141, 0xf, 0xf, false));
            m = fmaxf(m, __builtin_amdgcn_update_dpp(m, m, 0x140, 0xf, 0xf, false));
            const int k = row_scale_exp(m);
            cv_sc[b] = pow2f(k);
            if (lr == 0) *reinterpret_cast<float *>(lds + CF_RINV + (32 * rt + 8 * wave + 4 * b + sub) * 4) = pow2f(-k);
            if (b == 0) {
                // burst nb, column c = 4 (16 nb + lr): k step 4 nb + (lr >> 2), lane half (lr >> 1) & 1, element 4 (lr & 1); row 8 wave + 4 b + sub of the row tile
                cv_base = (lr >> 2) * CF_SLOT + (rt * 2) * 1024 + (((lr >> 1) & 1) * 32 + 8 * wave + sub) * 16 + (lr & 1) * 8;
                cv_base4 = lr < 8 ? cv_base + 16 * CF_SLOT : CF_DUMMY + lane * 8;          // columns 288 .. 319 do not exist: their lanes store into a scratch strip
            }
        } else {
            const int b = (piece - 4) / 5, nb = (piece - 4) % 5;
            unsigned ph0, ph1, pm0, pm1;
            asm volatile("v_fma_mixlo_f16 %0, %4, %8, 0\n\tv_fma_mixlo_f16 %1, %6, %8, 0\n\tv_fma_mixhi_f16 %0, %5, %8, 0\n\tv_fma_mixhi_f16 %1, %7, %8, 0\n\t"
                         "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
                         "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                         : "=&v"(ph0), "=&v"(ph1), "=&v"(pm0), "=&v"(pm1)
                         : "v"(stg[b][nb].x), "v"(stg[b][nb].y), "v"(stg[b][nb].z), "v"(stg[b][nb].w), "v"(cv_sc[b]));
            char *dst = lds + (nb < 4 ? cv_base : cv_base4) + b * 64;
            *reinterpret_cast<uint2 *>(dst + (nb < 4 ? nb * 4 * CF_SLOT : 0)) = make_uint2(ph0, ph1);
            *reinterpret_cast<uint2 *>(dst + (nb < 4 ? nb * 4 * CF_SLOT : 0) + 1024) = make_uint2(pm0, pm1);
        }
    };
    constexpr int CV_PIECES = 14;

    const float dw0 = meta[4 * 128 + 0], dw1 = meta[4 * 128 + 1], dw2 = meta[4 * 128 + 2], dw3 = meta[4 * 128 + 3];
    f32x16 acc[2];
    u32x4 bf[3][2];
    float inv[2] = {0.f, 0.f};
    float4 bq[2];                                                           // bias chunks (four columns each), two in flight
    float amax = 0.f, sc_run = 1.f;
    float s1x = 0.f, s1y = 0.f, s1a = 0.f, s1b = 0.f;
    unsigned ph[4], pm[4];                                                  // rings of four (a plane store leaves after every fourth item)
    float4 ex4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // output rows: buffer descriptors over C / C2 + per-lane byte offsets (a row past M: out of range, the store is dropped)
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.C), 0, (int)(M * a.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t c2_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.C2), 0, (int)(M * a.ldc2 * 4), 0x00020000);

    // ---- epilogue of pass PP = (PL, PR) of the tile whose first row is row_e, cut into micro-stages; `slot` counts the 3 S MFMAs of the RUNNING pass,
    //      first half in slots [0, H1), second half in [H1, H2) (the barriers sit in front of MFMA H1 and MFMA H2)
    auto epilogue_piece = [&](int PL, int PR, int S, int slot, long long row_e) __attribute__((always_inline)) {
        const int T = 3 * S, H1 = T >= 54 ? 18 : 9, H2 = T - 6, se = PR;
        const float inv_l = inv[PR];
        if (slot < H1) {
            // first half.  stage 0: constants; 1 + 2 q, 2 + 2 q: item q (two values); 17: row maximum -> exchange; 18, 19: the rows' stores (layer 2, tail)
            const int MS = 20, m0 = slot * MS / H1, m1 = (slot + 1) * MS / H1;
#pragma unroll
            for (int ms = m0; ms < m1; ++ms) {
                if (ms == 0) {
                    bq[0] = CF_AT(const float4, q_cst, PL * 512); bq[1] = CF_AT(const float4, q_cst, PL * 512 + 16);
                    if (PL == 0) inv[PR] = __fmul_rn(*reinterpret_cast<const float *>(lds + CF_RINV + (32 * PR + j) * 4), dw0);
                    amax = 0.f;
                } else if (ms < 17) {
                    const int q = (ms - 1) >> 1, st = (ms - 1) & 1;
                    if (st == 0) {
                        const float4 b4 = bq[(q >> 1) & 1];
                        const float bx = (q & 1) ? b4.z : b4.x, by = (q & 1) ? b4.w : b4.y;
                        if ((q & 1) == 0 && q >= 2 && (q >> 1) + 1 < 4) bq[((q >> 1) + 1) & 1] = CF_AT(const float4, q_cst, PL * 512 + ((q >> 1) + 1) * 16);       // chunk k + 1 at the start of chunk k (k >= 1)
                        s1x = fmaf(acc[se][2 * q], PL == 0 ? inv[PR] : inv_l, bx); s1y = fmaf(acc[se][2 * q + 1], PL == 0 ? inv[PR] : inv_l, by);
                        if (PL < 3) { s1a = __fmul_rn(s1x, a.slope); s1b = __fmul_rn(s1y, a.slope); }
                    } else {
                        const float vx = PL < 3 ? fmaxf(s1x, s1a) : s1x, vy = PL < 3 ? fmaxf(s1y, s1b) : s1y;
                        acc[se][2 * q] = vx; acc[se][2 * q + 1] = vy;
                        if (PL < 3) amax = fmaxf(fmaxf(amax, fabsf(vx)), fabsf(vy));
                    }
                } else if (ms == 17) {
                    if (PL < 3) {
                        const float m = fmaxf(amax, __shfl_xor(amax, 32));
                        if (h == 0) CF_AT(float, q_exw, PR * 512) = m;
                    }
                } else if (PL >= 2) {
                    // layer 2: the colour-feature rows; tail: its 64 columns (column tiles 0, 1: the other waves' offsets are out of range)
                    const long long row = row_e + 32 * PR + j;
                    const int half = ms - 18;
                    if (PL == 2) {
                        const int off = (row < M) ? (int)((row * a.ldc + col0) * 4) : 0x7fffff00;
#pragma unroll
                        for (int q4 = 2 * half; q4 < 2 * half + 2; ++q4)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[se][4 * q4]), __float_as_uint(acc[se][4 * q4 + 1]), __float_as_uint(acc[se][4 * q4 + 2]), __float_as_uint(acc[se][4 * q4 + 3])},
                                                                   c_rs, off + q4 * 16, 0, 0);
                    } else {
                        const int off = (row < M && col0 < 64) ? (int)((row * a.ldc2 + col0) * 4) : 0x7fffff00;
#pragma unroll
                        for (int q4 = 2 * half; q4 < 2 * half + 2; ++q4)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(acc[se][4 * q4]), __float_as_uint(acc[se][4 * q4 + 1]), __float_as_uint(acc[se][4 * q4 + 2]), __float_as_uint(acc[se][4 * q4 + 3])},
                                                                   c2_rs, off + q4 * 16, 0, 0);
                    }
                }
            }
            return;
        }
        if (slot >= H2 || PL == 3) return;
        // second half: stage 0: exchange read; 1: row scale; 2 + i (i = 0..8): fp16 high parts of item i and low parts of item i - 1, interleaved (chain_ws);
        // the four operand-plane stores behind the stages that complete them
        const int k2 = slot - H1, N2 = H2 - H1;
        const int MS = 11, m0 = k2 * MS / N2, m1 = (k2 + 1) * MS / N2;
#pragma unroll
        for (int ms = m0; ms < m1; ++ms) {
            if (ms == 0) ex4 = CF_AT(const float4, q_exr, PR * 512);
            else if (ms == 1) {
                const int k = row_scale_exp(fmaxf(fmaxf(ex4.x, ex4.y), fmaxf(ex4.z, ex4.w)));
                sc_run = pow2f(k);
                inv[PR] = __fmul_rn(pow2f(-k), PL == 0 ? dw1 : PL == 1 ? dw2 : dw3);
            } else {
                const int it = ms - 2, ip = it - 1;
                if (it == 0) {
                    asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(ph[0]) : "v"(acc[se][0]), "v"(acc[se][1]), "v"(sc_run));
                } else if (it < 8) {
                    asm volatile("v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
                                 "v_fma_mixlo_f16 %1, %4, %6, -%7 op_sel_hi:[0,0,1]\n\t"
                                 "v_fma_mixhi_f16 %0, %3, %6, 0\n\t"
                                 "v_fma_mixhi_f16 %1, %5, %6, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                 : "=&v"(ph[it & 3]), "=&v"(pm[ip & 3])
                                 : "v"(acc[se][2 * it]), "v"(acc[se][2 * it + 1]), "v"(acc[se][2 * ip]), "v"(acc[se][2 * ip + 1]), "v"(sc_run), "v"(ph[ip & 3]));
                } else {
                    asm volatile("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                 : "=&v"(pm[3]) : "v"(acc[se][14]), "v"(acc[se][15]), "v"(sc_run), "v"(ph[3]));
                }
                if (it == 3 || it == 7) CF_AT(u32x4, q_pub, PR * 2048 + (it == 7 ? 512 : 0)) = u32x4{ph[0], ph[1], ph[2], ph[3]};
                if (ip == 3 || ip == 7) CF_AT(u32x4, q_pub, PR * 2048 + 1024 + (ip == 7 ? 512 : 0)) = u32x4{pm[0], pm[1], pm[2], pm[3]};
            }
        }
    };

    auto b_read = [&](int L, int rt, int s, int p) __attribute__((always_inline)) {
        return CF_AT(const u32x4, q_b, (L == 0 ? 0 : CF_H) + s * CF_SLOT + (rt * 2 + p) * 1024);
    };

    // ---- prologue: constants visible, the first tile's rows converted (both row tiles), resident weights landed
    const int t_first = blockIdx.x, t_step = gridDim.x;
    __syncthreads();
    {
        const int t2 = t_first + t_step < n_tiles ? t_first + t_step : t_first;
        dma_rows(t_first);                                                  // this tile's row tile 0 (converted right here), then the next tile's (converted in this tile's pass (0,1))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int pc = 0; pc < CV_PIECES; ++pc) convert_piece(0, pc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (this wave's reads of the raw buffer are done before its next DMA overwrites it)
        dma_rows(t2);
        load_rows(t_first);                                                 // row tile 1: converted in pass (0,0)
    }
    __syncthreads();
    CF_REFRESH();
    bf[0][0] = b_read(0, 0, 0, 0); bf[0][1] = b_read(0, 0, 0, 1);
    bf[1][0] = b_read(0, 0, 1, 0); bf[1][1] = b_read(0, 0, 1, 1);
    long long row_prev = M;                                                 // first row of the tile whose last epilogue (tail, row tile 1) is still due: none yet
#ifdef HNR_CF_PROBE
    long long tm_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = clock64(), nt_ = 0;
#endif
    for (int tile = t_first; tile < n_tiles; tile += t_step) {
        const int tile_nx = tile + t_step < n_tiles ? tile + t_step : tile; // (past the end: the last tile's rows again -- no branch in the passes)
        const int tile_nx2 = tile + 2 * t_step < n_tiles ? tile + 2 * t_step : tile_nx;
        const long long row_cur = (long long)tile * 64;
        cf_static_for<8>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value, L = P >> 1, rt = P & 1, S = CF_S[L], G0 = cf_goff(P), T = 3 * S;
            constexpr int PP = (P + 7) & 7, PL = PP >> 1, PR = PP & 1;
            constexpr int H1 = T >= 54 ? 18 : 9, H2 = T - 6;
            const long long row_e = P == 0 ? row_prev : row_cur;
#ifdef HNR_CF_PROBE
            { const long long t_ = clock64(); tm_[(P + 7) & 7] += t_ - tp_; tp_ = t_; if (P == 0) ++nt_; }
#endif
            CF_REFRESH();
            __builtin_amdgcn_sched_barrier(0);
            cf_static_for<T>([&](auto kc) __attribute__((always_inline)) {
                constexpr int slot = decltype(kc)::value, it = slot / 3, g = slot % 3;
                if (slot == H1 || slot == H2) { cf_lds_barrier(); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (P == 1 && slot == H1 + 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }     // the DMA of a tile ago (nothing younger is in flight here)
                constexpr int wp = g == 0 ? 1 : 0, xp = g == 1 ? 1 : 0;     // wm*xh, wh*xm, wh*xh: the order of h2_mfma_layer
                constexpr int ring = (G0 + it) % 3;
                constexpr int areg = cf_areg(L, it, wp);
                if constexpr (areg >= 0) { if constexpr (slot == 0) cf_mfma_a_first<areg>(acc[rt], bf[ring][xp]); else cf_mfma_a<areg>(acc[rt], bf[ring][xp]); }
                else if constexpr (L == 2) cf_mfma_v(acc[rt], w2v[it - 6][wp], bf[ring][xp]);
                else { if constexpr (slot == 0) cf_mfma_v_first(acc[rt], wtv[it][wp], bf[ring][xp]); else cf_mfma_v(acc[rt], wtv[it][wp], bf[ring][xp]); }
                __builtin_amdgcn_sched_barrier(0);
                // activation fragments two iterations ahead (the next pass's first two: behind this pass's second barrier)
                if constexpr (g < 2) {
                    constexpr int i2 = it + 2 >= S ? it + 2 - S : it + 2, P2 = it + 2 >= S ? (P + 1) & 7 : P;
                    bf[(G0 + it + 2) % 3][g] = b_read(P2 >> 1, P2 & 1, i2, g);
                }
                { constexpr int li = cf_ld_index(P, slot); if constexpr (li >= 0) load_burst(tile_nx, li / 5, li % 5); }
                { constexpr int di = cf_dma_index(P, slot); if constexpr (di >= 0) dma_burst(tile_nx2, di / 5, di % 5); }
                epilogue_piece(PL, PR, S, slot, row_e);
                // the next rows' conversion: this tile's row tile 1 during pass (0,0) (its planes were last read a tile ago), the next tile's row tile 0 during
                // pass (0,1) behind the first barrier (every wave is past pass (0,0), the last reader of those planes)
                if constexpr (P == 0 && slot >= 3 && slot < 45 && (slot % 3) == 0) convert_piece(1, (slot - 3) / 3);
                if constexpr (P == 1 && slot >= 20 && slot < 48 && ((slot - 20) & 1) == 0) convert_piece(0, (slot - 20) / 2);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        row_prev = row_cur;
    }
#ifdef HNR_CF_PROBE
    if (blockIdx.x == 0 && tid == 0) { for (int i = 0; i < 8; ++i) g_cf_probe[i] = tm_[i]; g_cf_probe[8] = nt_; }
#endif
    // ---- drain: the last tile's tail epilogue of row tile 1 (first half only: bias + stores)
    CF_REFRESH();
    cf_static_for<18>([&](auto kc) __attribute__((always_inline)) { epilogue_piece(3, 1, 18, decltype(kc)::value, row_prev); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the rows asked for ahead (DMA into this workgroup's LDS) must not outlive the workgroup
}

int launch_cf_ws(const CfWsArgs &a, hipStream_t st)
{
    static PerDeviceOnce attr;
    if (attr.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(cf_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, cf_lds_bytes()));
    const int n_cu = device_num_cus();
    const long long tiles = (a.M_cap + 63) / 64;
    const int grid = (int)(tiles < n_cu ? tiles : n_cu);
    cf_ws_kernel<<<grid, 256, cf_lds_bytes(), st>>>(a);
    HNR_LAUNCH_CHECK();
#ifdef HNR_CF_PROBE
    {
        long long h[12];
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cf_probe), sizeof(h)) == hipSuccess && h[8] > 0) {
            fprintf(stderr, "[cf_ws probe] %lld tiles, cycles per pass (pass P ends at index P; index 7 = pass (3,1) + tile switch):", h[8]);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", h[i] / h[8]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return HNR_OK;
}

}  // namespace hnr
