// Weight-stationary input gradient of a 256 x 256 per-neighbour layer of the TRAINING step: dX = (dZ W) * LeakyReLU'(forward activation), the three
// largest launches of hnr_render_train_backward (what torch autograd derives for block3.2 / block3.0 / block1.2 of PointAggregator.viewmlp,
// /root/reference/models/aggregators/point_aggregators.py:948-975).  Same arithmetic as h2lin_kernel<16> (csrc/h2gemm.hip: per-row power-of-two scale, fp16
// (h, m) planes, three v_mfma_f32_32x32x16_f16 per product in the order w_m x_h, w_h x_m, w_h x_h, k steps ascending, the same epilogue operations):
// bit-identical results.  What differs is where the operands live and when things are issued:
//   * h2lin_kernel streams the layer's 256 KiB of weight fragments from L2 for every 64-row tile -- its address unit is busy 0.61 of the time
//     (profiles/README.md, round 5) and its phases (row loads + split | MFMA loop | epilogue) are serial per workgroup.  Here a wave loads its 64 output
//     columns' fragments ONCE into 256 self-numbered AGPRs (as csrc/chain_ws.hip does per layer) and streams 32-row tiles past them;
//   * the loop is software-pipelined by hand: iteration i multiplies tile i (96 MFMAs per wave) and, between those MFMAs, converts the rows of tile
//     i + 1 (loaded two iterations earlier into one of two register sets) into the next of THREE LDS plane buffers, reloads that set with tile i + 3, and
//     runs the epilogue of tile i - 1 out of the other accumulator set (values in place, then quad-transposed 64-byte stores).  One workgroup barrier per
//     tile, in mid-iteration; every piece behind an MFMA is 4 - 7 instructions (the slot tables below).  Measured: profiles/r06_train_gemms.txt.
// The compiler must not touch AGPRs or scratch in this file (Makefile: build/h2lin_ws.checked).
#include <stdio.h>
#include <stdlib.h>
#include <utility>

#include "chain_defs.h"

namespace hnr {

constexpr int HW_S = 16;                                   // k steps (K = 256)
constexpr int HW_SLOT = 2048 + 32;                         // LDS bytes of one k step of a tile's operand planes: [plane 2][64 lanes][16 B] + pad (as h2lin_kernel)
constexpr int HW_BUF = HW_S * HW_SLOT;
constexpr int HW_RINV = 3 * HW_BUF;                        // (three plane buffers) float [4][32]: 2^-k of the rows of tiles i - 1 .. i + 1 (index tile & 3)
constexpr int HW_DUMP = HW_RINV + 4 * 32 * 4;              // float [4 waves][64]: where lanes 1..63 put the value lane 0 publishes
constexpr int HW_LDS = HW_DUMP + 4 * 64 * 4 + 16;
constexpr int HW_DESC = 256;                               // meta float: the layer's descale (HL_DESC of h2gemm.hip)

// Which pieces run behind MFMA `slot` = 6 (k step) + (MFMA of the k step), 96 slots per tile; every piece is 4 - 7 instructions, so that the wave's VALU
// work is spread evenly under its MFMAs (32 cycles of matrix pipe each: room for ~6 VALU instructions).
//   conversion: 36 ops (2 groups x 18, alternating) on the even slots before the barrier (end of k step 11);
//   epilogue: 52 ops -- column tile 0's sixteen values, its transpose + stores (9), column tile 1's values and stores, 2 ops that prepare the next
//   iteration's epilogue -- on the odd slots before the barrier and two of every three slots after it.
constexpr int HW_BAR_STEP = 11;
constexpr int hw_conv_at(int slot) { return (slot < 72 && !(slot & 1)) ? slot / 2 : -1; }
constexpr int hw_epi_at(int slot)
{
    if (slot < 72) return (slot & 1) ? (slot - 1) / 2 : -1;
    for (int k = 0; k < 16; ++k) if (72 + (k * 24) / 16 == slot) return 36 + k;
    return -1;
}

struct H2LinWsArgs {
    const float *A; int lda;
    const long long *d_m; long long M_cap;
    const char *wimg;
    float slope;
    const uint32_t *side_bits;
    float *C; int ldc;
    unsigned *absmax;
};

#ifdef HNR_WS_PROBE
__device__ long long g_hw_probe[40];      // probe build (make EXTRA=-DHNR_WS_PROBE): cycles per k step of wave 0 of workgroup 0, [17] = iterations
#define HW_STAMP(k_) do { const long long t_ = clock64(); tm_[k_] += t_ - tp_; tp_ = t_; } while (0)
#else
#define HW_STAMP(k_) do { } while (0)
#endif

namespace {

#define HW_LOAD_FRAG(N_, rsrc_, voff_, soff_, IMM_) asm volatile("buffer_load_dwordx4 a[%2:%3], %0, %1, %4 offen offset:%5" :: "v"(voff_), "s"(rsrc_), "n"(N_), "n"((N_) + 3), "s"(soff_), "n"(IMM_))
template <int N> __device__ __forceinline__ void hw_mfma(f32x16 &acc, const u32x4 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, %0" : "+v"(acc) : "v"(x), "n"(N), "n"(N + 3));
}
template <int N> __device__ __forceinline__ void hw_mfma_first(f32x16 &acc, const u32x4 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%2:%3], %1, 0" : "=&v"(acc) : "v"(x), "n"(N), "n"(N + 3));
}
template <int s> __device__ __forceinline__ void hw_load_step(__amdgpu_buffer_rsrc_t wsrd, int voff)
{
    int soff = s * CH_WSTEP;
    HW_LOAD_FRAG(16 * s + 0, wsrd, voff, soff, 0);
    HW_LOAD_FRAG(16 * s + 4, wsrd, voff, soff, 1024);
    HW_LOAD_FRAG(16 * s + 8, wsrd, voff, soff, 2048);
    HW_LOAD_FRAG(16 * s + 12, wsrd, voff, soff, 3072);
}
template <int... Ss> __device__ __forceinline__ void hw_load_all(__amdgpu_buffer_rsrc_t wsrd, int voff, std::integer_sequence<int, Ss...>) { (hw_load_step<Ss>(wsrd, voff), ...); }
// In-lane half of the quad's 4 x 4 transpose of 16-byte pieces (cw_table_swap of csrc/chain_ws.hip): slot d <- register t ^ d, t = lane & 3 --
// two rounds of exec-masked v_swap_b32 (a: lanes with t & 1 swap slots 0 <-> 1, 2 <-> 3; b: lanes with t & 2 swap 0 <-> 2, 1 <-> 3).
__device__ __forceinline__ void hw_quad_swap_a(float4 &r0, float4 &r1, float4 &r2, float4 &r3)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %16, exec\n\ts_mov_b32 exec_lo, 0xaaaaaaaa\n\ts_mov_b32 exec_hi, 0xaaaaaaaa\n\t"
                 "v_swap_b32 %0, %4\n\tv_swap_b32 %1, %5\n\tv_swap_b32 %2, %6\n\tv_swap_b32 %3, %7\n\t"
                 "v_swap_b32 %8, %12\n\tv_swap_b32 %9, %13\n\tv_swap_b32 %10, %14\n\tv_swap_b32 %11, %15\n\t"
                 "s_mov_b64 exec, %16\n\ts_nop 1"
                 : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w),
                   "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w), "+v"(r3.x), "+v"(r3.y), "+v"(r3.z), "+v"(r3.w), "=&s"(keep));
}
__device__ __forceinline__ void hw_quad_swap_b(float4 &r0, float4 &r1, float4 &r2, float4 &r3)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %16, exec\n\ts_mov_b32 exec_lo, 0xcccccccc\n\ts_mov_b32 exec_hi, 0xcccccccc\n\t"
                 "v_swap_b32 %0, %8\n\tv_swap_b32 %1, %9\n\tv_swap_b32 %2, %10\n\tv_swap_b32 %3, %11\n\t"
                 "v_swap_b32 %4, %12\n\tv_swap_b32 %5, %13\n\tv_swap_b32 %6, %14\n\tv_swap_b32 %7, %15\n\t"
                 "s_mov_b64 exec, %16\n\ts_nop 1"
                 : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w),
                   "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w), "+v"(r3.x), "+v"(r3.y), "+v"(r3.z), "+v"(r3.w), "=&s"(keep));
}
template <class F, int... Is> __device__ __forceinline__ void hw_static_seq(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void hw_static_for(F &&f) { hw_static_seq(f, std::make_integer_sequence<int, N>{}); }

}  // namespace

__global__ __launch_bounds__(256, 1) void h2lin_ws_kernel(H2LinWsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, j = lane & 31;
    long long M = a.M_cap;
    if (a.d_m) { const long long c = *a.d_m; if (c < M) M = c; }
    const int n_tiles = (int)((M + 31) >> 5);
    if ((int)blockIdx.x >= n_tiles) return;
    const int n_my = (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int last_tile = (int)blockIdx.x + (n_my - 1) * (int)gridDim.x;
    asm volatile("" ::: "a255");                                           // the kernel owns all 256 AGPRs
    // ---- this wave's weight fragments: (k step s, column tile c of the wave, plane p) -> a[16 s + 8 c + 4 p .. + 3]
    {
        const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.wimg), 0, HW_S * CH_WSTEP, 0x00020000);
        hw_load_all(wsrd, wave * 4096 + lane * 16, std::make_integer_sequence<int, HW_S>{});
    }
    const float dw = reinterpret_cast<const float *>(a.wimg + (size_t)HW_S * CH_WSTEP)[HW_DESC];
    float *rinv_all = reinterpret_cast<float *>(lds + HW_RINV), *rinv_dump = reinterpret_cast<float *>(lds + HW_DUMP) + 64 * wave;
    // tile of this workgroup's i-th iteration; past the end: the last one again (its results are not stored a second time)
    auto tile_of = [&](int i) { const int t = (int)blockIdx.x + i * (int)gridDim.x; return t < n_tiles ? t : last_tile; };
    // Rows 8 wave .. + 7 of a tile as two groups g of four rows: lane (sub = lane >> 4, lr = lane & 15) holds columns 4 (lr + 16 nb) .. + 3, nb = 0..3, of row
    // 8 wave + 4 g + sub -- sixteen lanes per row, so that a row's maximum is 4 DPP steps for FOUR rows at once (one lane group per row: 6 steps per row).
    // Rows past M inside the last tile are read as they are (the operand has M_cap rows, a multiple of 32): every row of this product is independent of
    // the others, theirs are neither stored nor counted.
    const int sub = lane >> 4, lr = lane & 15;
    const int ld_off = sub * a.lda + 4 * lr;
    auto load_q = [&](int i, int g, int nb) -> float4 {
        const float *base = a.A + ((size_t)tile_of(i) * 32 + 8 * wave + 4 * g) * (size_t)a.lda;      // uniform: scalar arithmetic
        return *reinterpret_cast<const float4 *>(base + ld_off + 64 * nb);
    };
    // lane-constant part of the plane addresses of the lane's values: k step (lr >> 2) + 4 nb, lane half (lr >> 1) & 1, elements 4 (lr & 1) .., row 8 wave + 4 g + sub
    char *cv_base = lds + (lr >> 2) * HW_SLOT + ((lr >> 1) & 1) * 512 + (lr & 1) * 8 + (8 * wave + sub) * 16;
    // The conversion of a group in eighteen OPS (an op = what is issued behind one MFMA): 0, 1: the lane's maximum; 2, 3: the row's (DPP butterfly over its
    // sixteen lanes); 4: power-of-two scale; 5: lane lr = 0 publishes 2^-k; 6 + 3 nb + {0, 1, 2}: column block nb: scale + fp16 split of its first / second
    // pair, then the plane stores and the reload of the register with the block of the tile two further on.  The two groups' ops alternate: two
    // independent dependency chains.
    float cm[2] = {0.f, 0.f}, cmb[2] = {0.f, 0.f}, csc[2] = {0.f, 0.f};
    unsigned cph[2][2] = {{0u, 0u}, {0u, 0u}}, cpm[2][2] = {{0u, 0u}, {0u, 0u}};
    auto conv_op = [&](float4 (&v)[4], int g, int op, int tile_i, int wb, int i_next) __attribute__((always_inline)) {
        float &m = cm[g];
        // m = max(m, m of the DPP source lane) as ONE instruction (the builtin form compiles to mov_dpp + canonicalise + max + a copy).
        // s_nop 1: a DPP operand written by the previous VALU instruction needs two wait states.
#define HW_MAX_DPP(ctrl_) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl_ : "+v"(m))
        // (v_max3_f32 with |.| on its sources: two values per instruction, two independent chains; written as fmaxf(fabsf()) the compiler canonicalises every
        // operand first -- 16 instructions for 16 values)
#define HW_MAX3(d_, a_, b_, c_) asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d_) : "v"(a_), "v"(b_), "v"(c_))
        if (op == 0) {
            float ma, mb;
            HW_MAX3(ma, v[0].x, v[0].y, v[0].z); HW_MAX3(mb, v[0].w, v[1].x, v[1].y); HW_MAX3(ma, v[1].z, v[1].w, ma); HW_MAX3(mb, v[2].x, v[2].y, mb);
            m = ma; cmb[g] = mb;
        } else if (op == 1) {
            float ma = m, mb = cmb[g];
            HW_MAX3(ma, v[2].z, v[2].w, ma); HW_MAX3(mb, v[3].x, v[3].y, mb); HW_MAX3(ma, v[3].z, v[3].w, ma);
            m = fmaxf(ma, mb);
        }
#undef HW_MAX3
        else if (op == 2) { HW_MAX_DPP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); HW_MAX_DPP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"); }
        else if (op == 3) { HW_MAX_DPP("row_half_mirror row_mask:0xf bank_mask:0xf"); HW_MAX_DPP("row_mirror row_mask:0xf bank_mask:0xf"); }
#undef HW_MAX_DPP
        else if (op == 4) {
            const int k = row_scale_exp(m);
            csc[g] = pow2f(k);
            m = pow2f(-k);
        } else if (op == 5) {
            // (the other lanes write a dump word each: no exec-masked branch in the MFMA stream)
            float *dst = lr == 0 ? rinv_all + (tile_i & 3) * 32 + 8 * wave + 4 * g + sub : rinv_dump + lane;
            *dst = m;
        } else {
            const int nb = (op - 6) / 3, part = (op - 6) % 3;
            const float sc = csc[g];
            if (part == 0) split2h(__fmul_rn(v[nb].x, sc), __fmul_rn(v[nb].y, sc), cph[g][0], cpm[g][0]);
            else if (part == 1) split2h(__fmul_rn(v[nb].z, sc), __fmul_rn(v[nb].w, sc), cph[g][1], cpm[g][1]);
            else {
                char *dst = cv_base + wb + g * 64 + nb * 4 * HW_SLOT;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(cph[g][0], cph[g][1]);
                *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(cpm[g][0], cpm[g][1]);
                v[nb] = load_q(i_next, g, nb);
            }
        }
    };
    float4 rows[2][2][4];
    f32x16 acc[2][2];
    float gmax = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { rows[0][g][nb] = load_q(0, g, nb); rows[1][g][nb] = load_q(1, g, nb); }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][c][r] = 0.f;                    // (iteration 0 runs an epilogue over these)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the weight fragments (loads the compiler does not know of) and the first rows
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int op = 0; op < 18; ++op) conv_op(rows[0][g], g, op, 0, 0, 2);
    __syncthreads();

    const int colb = 64 * wave + 16 * h;
    const char *fr_base = lds + lane * 16;
#ifdef HNR_WS_PROBE
    long long tm_[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = clock64();
#endif
    unsigned bits_carry = 0u;
    float rinv_carry = 0.f, inv = 0.f;                                      // inv: 2^-k of row j of the tile whose epilogue runs x the layer's descale (0: a dead row)
    int st_base = (int)0x80000000;                                          // byte offset of (row (j & ~3) of that tile, this lane half's 64 bytes of the wave's first column tile); iteration 0: out of range
    float4 tb[4];
    u32x4 xf[2][2];
    xf[0][0] = *reinterpret_cast<const u32x4 *>(fr_base);
    xf[0][1] = *reinterpret_cast<const u32x4 *>(fr_base + 1024);
    int rb = 0, wb = HW_BUF;                                                // byte offsets of the buffer read by this iteration's MFMAs / written by its conversion
    const int t16 = (j & 3) * 16, ld4 = a.ldc * 4;
    const unsigned slope_bits = __float_as_uint(a.slope);
    // the output rows through a buffer descriptor that ends with row M - 1: rows past M (and everything in iteration 0) fall outside and are dropped
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)(M * a.ldc * 4), 0x00020000);
    // Iteration i (PAR = i & 1): MFMAs of tile i from buffer i % 3 into acc[PAR]; behind them, one small piece per MFMA ("slot" 6 s + pc):
    //   * k steps 0..11: rows of tile i + 1 (register set PAR ^ 1) -> buffer (i + 1) % 3, the set reloaded with tile i + 3; then the iteration's ONE barrier.
    //     Three buffers: the buffer written here was last read in iteration i - 1, which every wave has left before it passes iteration i's barrier, and the
    //     next iteration's first fragments can be fetched at the end of this one (no LDS latency at the head of an iteration);
    //   * all k steps: epilogue of tile i - 1 out of acc[PAR ^ 1], in place, then the transposed stores; last, what the next iteration's epilogue needs.
    // i = n_my multiplies the last tile once more (nothing of it is stored).
    auto body = [&](auto par_c, int i) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value, OTH = PAR ^ 1;
        const char *fb = fr_base + rb;
        const unsigned bits = bits_carry;                                   // sign words of tile i - 1 (loaded one iteration ago)
        bits_carry = a.side_bits[((size_t)tile_of(i) * 4 + wave) * 64 + lane];
        auto value = [&](int c, int vi) __attribute__((always_inline)) {
            float v = fmaf(acc[OTH][c][vi], inv, 0.f);
            // x (bit ? 1 : slope) without a compare: the bit spread over a word (v_bfe_i32), then a bit-field select between the two constants -- no SGPR
            // pair written by a VALU instruction and read by the next (a hazard nop behind every v_cmp / v_cndmask pair)
            unsigned fac;                                                  // (one asm statement: the compiler pads every asm whose result is used next with a nop)
            asm("v_bfe_i32 %0, %1, %2, 1\n\tv_bfi_b32 %0, %0, %3, %4" : "=&v"(fac) : "v"(bits), "n"(31 - (16 * c + vi)), "v"(0x3f800000u), "v"(slope_bits));
            v = __fmul_rn(v, __uint_as_float(fac));
            acc[OTH][c][vi] = v;
            gmax = fmaxf(gmax, fabsf(v));
        };
        // the sixteen values of (row j, column tile c) leave as four 16-B stores of 64 consecutive bytes per quad (csrc/chain_ws.hip, tr_group): straight
        // from the accumulator layout a quad's lanes write four different rows -- one address-unit cycle per (quad, cache line), 64 per instruction
        auto dppx = [&](float v, int x) __attribute__((always_inline)) -> float {      // the value of lane t ^ x of the quad (bit pattern moved: the builtin on ints)
            const int b = __builtin_bit_cast(int, v);
            const int r = x == 1 ? __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xf, 0xf, false) : x == 2 ? __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xf, 0xf, false)
                                                                                             : __builtin_amdgcn_update_dpp(0, b, 0x1B, 0xf, 0xf, false);
            return __builtin_bit_cast(float, r);
        };
        auto transpose = [&](int c, int sub_) __attribute__((always_inline)) {
            const f32x16 &o = acc[OTH][c];
            if (sub_ == 0) { tb[0] = make_float4(o[0], o[1], o[2], o[3]); tb[1] = make_float4(dppx(o[4], 1), dppx(o[5], 1), dppx(o[6], 1), dppx(o[7], 1)); }
            else if (sub_ == 1) tb[2] = make_float4(dppx(o[8], 2), dppx(o[9], 2), dppx(o[10], 2), dppx(o[11], 2));
            else if (sub_ == 2) tb[3] = make_float4(dppx(o[12], 3), dppx(o[13], 3), dppx(o[14], 3), dppx(o[15], 3));
            else if (sub_ == 3) hw_quad_swap_a(tb[0], tb[1], tb[2], tb[3]);
            else if (sub_ == 4) hw_quad_swap_b(tb[0], tb[1], tb[2], tb[3]);
            else {
                const int r = sub_ - 5;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(tb[r].x), __float_as_uint(tb[r].y), __float_as_uint(tb[r].z), __float_as_uint(tb[r].w)}, crs,
                                                       st_base + r * ld4 + 128 * c + ((16 * r) ^ t16), 0, 0);
            }
        };
        // what the NEXT iteration's epilogue (tile i) needs; the stores of this one have been issued
        auto prepare = [&](int part) __attribute__((always_inline)) {
            const int tn = tile_of(i);
            if (part == 0) st_base = ((tn * 32 + (j & ~3)) * a.ldc + colb) * 4;
            else inv = ((long long)tn * 32 + j < M) ? __fmul_rn(rinv_carry, dw) : 0.f;      // rows past M: scaled by 0 -- never the maximum -- and out of the store range
        };
        hw_static_for<HW_S>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value, cur = s & 1, nxt = cur ^ 1;
            if constexpr (s + 1 < HW_S) {
                xf[nxt][0] = *reinterpret_cast<const u32x4 *>(fb + (s + 1) * HW_SLOT);
                xf[nxt][1] = *reinterpret_cast<const u32x4 *>(fb + (s + 1) * HW_SLOT + 1024);
            } else {
                // the next iteration's first fragments (its buffer is complete since this iteration's barrier) and its epilogue's row scales
                xf[nxt][0] = *reinterpret_cast<const u32x4 *>(fr_base + wb);
                xf[nxt][1] = *reinterpret_cast<const u32x4 *>(fr_base + wb + 1024);
                rinv_carry = rinv_all[(i & 3) * 32 + j];
            }
            hw_static_for<6>([&](auto pc_c) {
                constexpr int pc = decltype(pc_c)::value, slot = 6 * s + pc;
                // smallest terms first (w_m x_h, w_h x_m, w_h x_h), the two accumulators alternating
                constexpr int cc = pc & 1, term = pc >> 1, wp = term == 0 ? 1 : 0, xp = term == 1 ? 1 : 0;
                if constexpr (s == 0 && term == 0) hw_mfma_first<16 * s + 8 * cc + 4 * wp>(acc[PAR][cc], xf[cur][xp]);
                else hw_mfma<16 * s + 8 * cc + 4 * wp>(acc[PAR][cc], xf[cur][xp]);
                // the pieces behind this MFMA
                constexpr int cn = hw_conv_at(slot);
                if constexpr (cn >= 0) conv_op(rows[OTH][cn & 1], cn & 1, cn >> 1, i + 1, wb, i + 3);
                constexpr int en = hw_epi_at(slot);
                if constexpr (en >= 0 && en < 16) value(0, en);
                else if constexpr (en >= 16 && en < 25) transpose(0, en - 16);
                else if constexpr (en >= 25 && en < 41) value(1, en - 25);
                else if constexpr (en >= 41 && en < 50) transpose(1, en - 41);
                else if constexpr (en >= 50) prepare(en - 50);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (s == HW_BAR_STEP) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
            HW_STAMP(s);
        });
        rb = wb;
        wb = wb + HW_BUF < 3 * HW_BUF ? wb + HW_BUF : 0;
#ifdef HNR_WS_PROBE
        tm_[17] += 1;
#endif
    };
    {
        int i = 0;
        for (; i + 1 <= n_my; i += 2) { body(std::integral_constant<int, 0>{}, i); body(std::integral_constant<int, 1>{}, i + 1); }
        if (i <= n_my) body(std::integral_constant<int, 0>{}, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef HNR_WS_PROBE
    if (blockIdx.x == 0 && tid == 0) for (int k = 0; k < 18; ++k) g_hw_probe[k] = tm_[k];
#endif
    if (a.absmax) {
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
        float *s_m = reinterpret_cast<float *>(lds + HW_DUMP + 4 * 64 * 4);
        if (lane == 0) s_m[wave] = gmax;
        __syncthreads();
        if (tid == 0) { gmax = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])); if (gmax > 0.f) atomicMax(a.absmax, __float_as_uint(gmax)); }
    }
}

// csrc/h2gemm.hip (h2lin_dgrad_bits) calls this for N = K = 256 when HNR_H2LIN_WS != 0
int launch_h2lin_ws(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, float slope, const uint32_t *d_side_bits, float *d_C, int ldc,
                    uint32_t *d_absmax, void *stream)
{
    if (M_cap <= 0) return HNR_OK;
    if ((long long)M_cap * ldc * 4 >= 0x7fffffffLL) { set_error("h2lin_ws: more than 2 GiB of output rows (32-bit store offsets)"); return HNR_ERR_BADARG; }
    H2LinWsArgs a;
    a.A = d_dZ; a.lda = ldz; a.d_m = reinterpret_cast<const long long *>(d_m); a.M_cap = M_cap; a.wimg = (const char *)d_packed; a.slope = slope;
    a.side_bits = d_side_bits; a.C = d_C; a.ldc = ldc; a.absmax = d_absmax;
    static PerDeviceOnce once;
    if (once.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(h2lin_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HW_LDS));
    const int64_t tiles = (M_cap + 31) / 32;
    const int n_cu = device_num_cus(), grid = (int)(tiles < n_cu ? tiles : n_cu);
    h2lin_ws_kernel<<<grid, 256, HW_LDS, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
#ifdef HNR_WS_PROBE
    if (getenv("HNR_WS_PROBE_PRINT")) {
        long long hp[40];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_hw_probe), sizeof(hp)) == hipSuccess && hp[17] > 0) {
            fprintf(stderr, "h2lin_ws probe: %lld iterations; cycles per k step:", hp[17]);
            long long tot = 0;
            for (int k = 0; k < 17; ++k) { fprintf(stderr, " %lld", hp[k] / hp[17]); tot += hp[k]; }
            fprintf(stderr, "  = %lld per tile\n", tot / hp[17]);
        }
    }
#endif
    return HNR_OK;
}

}  // namespace hnr
