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
//
// Load schedule (round 4).  `s_waitcnt vmcnt(N)` counts in issue order, and the compiler's waits for its own loads know nothing of the asm
// loads queued between them: every wait that lands behind a FRESH load stalls the wave (and with it the matrix pipe) for a full L2 / HBM
// latency.  The per-pass clocks showed exactly that (pass (0,0) 7.0 k cycles for 24 MFMAs, the other layer-0 passes 3.5 k): so
//   * block1.2's k steps 4..15 are fetched during pass (3,3) (layer 0 only occupies register blocks 0..3), not during the short layer-0
//     passes whose epilogues consume the table rows; block1.2 runs its k steps in the order 8..15, 0..7, so that the four blocks that can
//     only be fetched during pass (0,3) are used half a pass later (every row tile uses the same order: results do not depend on the slot);
//   * the next tile's layer-0 operand image goes global -> LDS by DMA (no registers, no wait inside a pass: issued after the barrier of the
//     pass that read the region last, waited for -- by count -- before the barrier two or three passes later);
//   * block3.0's 17th k step (VGPRs) is fetched once per tile, the next tile's point ids two passes before its table rows are asked for;
//   * nothing inside a pass is conditional on "is there a next tile" (the last tile re-fetches its own data): no branches, so the
//     compiler cannot sink an epilogue piece past the MFMAs it was placed between.
#include <utility>

#include "chain_defs.h"

namespace hnr {

constexpr int CW_CST = ch_lds_exch(4) + 2048;          // LDS copy of bias[4][256], alpha_w[256], alpha_b, descale[4] (meta floats 0..1284)
constexpr int CW_CST_FLOATS = CH_META_DESCALE + 4;
constexpr int CW_DSUM = CW_CST + ((CW_CST_FLOATS * 4 + 15) & ~15);   // alpha-branch partial dot products of a tile: [4 row tiles][32 rows][4 waves]
constexpr int CW_HMAX = CW_DSUM + 4 * 32 * 4 * 4;        // training form: running maxima [4 layers + the stored sums][32 rows][4 waves] (ds_max_f32: no register carried through the tile loop)
constexpr int cw_lds_bytes() { return CW_HMAX + 5 * 512; }

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

// the layer-0 operand image of one row tile: two 1-KiB chunks per wave, global -> LDS (lane l's 16 B land at M0 + 16 l)
__device__ __forceinline__ void cw_dma_image(const char *g0, const char *g1, int voff, unsigned lds0, unsigned lds1)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt" :: "v"(voff), "s"(g0), "s"(g1), "s"(lds0), "s"(lds1) : "memory", "m0");
}

// Per-point table rows (layer 0's addend).  A lane (h, j) needs 16 consecutive floats of row j per column tile; fetched that way (four 16-B loads
// per lane and column tile) the four lanes of a quad hit four different rows in every instruction, and the CU's vector-memory front end spends one
// cycle per (quad, cache line): 64 per instruction, 2.1 k cycles per row tile for the four waves -- what the layer-0 passes were bound by
// (tools/gather_probe.hip: 2146 cycles per row tile that way against 696 when a quad reads 64 contiguous bytes).  So instruction r of a column tile
// reads row (j & ~3) + r for the whole quad, lane t = j & 3 taking piece r ^ t of its 64 bytes, and the 4 x 4 transpose happens in registers: first
// inside each lane (slot d <- register t ^ d: two exec-masked rounds of v_swap_b32), then across the lanes for free -- piece p of the lane's own row
// now sits in slot p of lane t ^ p, and the epilogue's `bias + table` add reads it through DPP (quad_perm = xor p).
__device__ __forceinline__ void cw_table_swap(float4 &r0, float4 &r1, float4 &r2, float4 &r3)
{
    unsigned long long keep;
    asm volatile("s_mov_b64 %16, exec\n\ts_mov_b32 exec_lo, 0xaaaaaaaa\n\ts_mov_b32 exec_hi, 0xaaaaaaaa\n\t"
                 "v_swap_b32 %0, %4\n\tv_swap_b32 %1, %5\n\tv_swap_b32 %2, %6\n\tv_swap_b32 %3, %7\n\t"
                 "v_swap_b32 %8, %12\n\tv_swap_b32 %9, %13\n\tv_swap_b32 %10, %14\n\tv_swap_b32 %11, %15\n\t"
                 "s_mov_b32 exec_lo, 0xcccccccc\n\ts_mov_b32 exec_hi, 0xcccccccc\n\t"
                 "v_swap_b32 %0, %8\n\tv_swap_b32 %1, %9\n\tv_swap_b32 %2, %10\n\tv_swap_b32 %3, %11\n\t"
                 "v_swap_b32 %4, %12\n\tv_swap_b32 %5, %13\n\tv_swap_b32 %6, %14\n\tv_swap_b32 %7, %15\n\t"
                 "s_mov_b64 exec, %16\n\ts_nop 1"
                 : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w),
                   "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w), "+v"(r3.x), "+v"(r3.y), "+v"(r3.z), "+v"(r3.w), "=&s"(keep));
}
template <int P> __device__ __forceinline__ float cw_table_add(float t, float b)      // b + (slot-P value of lane t ^ P): fp32 add, either operand order
{
    float r;
    if constexpr (P == 0) r = __fadd_rn(b, t);
    else if constexpr (P == 1) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(t), "v"(b));
    else if constexpr (P == 2) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(t), "v"(b));
    else asm("v_add_f32_dpp %0, %1, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(t), "v"(b));
    return r;
}

// compile-time loops: the pass / k step / piece indices must be constants at every use (register arrays, asm operands), and the body is too
// large for `#pragma unroll` to accept
template <class F, int... Is> __device__ __forceinline__ void cw_static_seq(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void cw_static_for(F &&f) { cw_static_seq(f, std::make_integer_sequence<int, N>{}); }

constexpr int cw_steps(int L) { return L == 0 ? CH_S0 : L == 1 ? CH_S1 : L == 2 ? CH_S2 : CH_S3; }
constexpr int cw_korder(int L, int i) { return L == 1 ? ((i + 8) & 15) : i; }     // k step of a pass's i-th iteration (see "Load schedule")
constexpr int cw_wbase(int L) { return L == 0 ? CH_W0 : L == 1 ? CH_W1 : L == 2 ? CH_W2 : CH_W3; }
constexpr int cw_goff(int P) { int g = 0; for (int p = 0; p < P; ++p) g += cw_steps(p >> 2); return g; }     // k steps of the tile before pass P
constexpr int CW_GTOT = cw_goff(16);

// ---- the weight stream's static schedule.  A layer's 256 KiB of fragments pass the CU's vector-memory front end at 64 B / clock: 4.1 k
// cycles -- more than a pass's 3.1 k cycles of MFMAs, so a pass that fetches a whole layer (as passes (1,3), (2,3), (3,3) did) is bound by
// the fetch (4.1 - 4.3 k cycles even without any epilogue).  A register block can be refilled any time between its last use by the layer's last
// pass and its first use one pass later, so half of each layer is fetched during the NEXT layer's first pass, and block1.2's twelve blocks
// that layer 0 does not occupy during the four short layer-0 passes (second halves: the first halves consume the table rows, and a wait for
// those would also wait for any younger load).  cw_issue(P, slot): the block (layer * 32 + k step) fetched behind MFMA `slot` of pass P, or -1.
constexpr int cw_issue(int P, int slot)
{
    const int L = P >> 2, rt = P & 3;
    // block1.2 (runs its k steps in the order 8..15, 0..7): blocks 8..11 behind iterations 8..11 of pass (3,3) (layer 0 does not use them), 12..15 one or
    // two per short layer-0 pass (second halves -- every vector-memory instruction costs the wave ~40 cycles of issue there: 16 a pass made the
    // layer-0 passes fetch-bound), 0..7 during the first half of pass (1,0) itself, which uses them from iteration 8 on
    if (L == 0) {
        if (rt == 0) return slot == 14 ? 32 + 12 : slot == 20 ? 32 + 13 : -1;
        if (rt == 1) return slot == 17 ? 32 + 14 : -1;
        if (rt == 2) return slot == 17 ? 32 + 15 : -1;
        return -1;
    }
    if (slot % 6 != 5) return -1;
    const int it = slot / 6;
    if (P == 4) return it < 8 ? 32 + it : -1;                              // behind iteration it (k step 8 + it): block `it`, used from iteration 8 + it on
    if (P == 7) {                                                          // block3.0 (block1.2 runs k steps 8..15, 0..7: block b is free after iteration (b + 8) & 15)
        const int t[16] = {-1, -1, 8, -1, -1, 9, -1, -1, 0, 1, -1, 2, 3, -1, 4, 5};
        return t[it] < 0 ? -1 : 64 + t[it];
    }
    if (P == 8) { const int t[17] = {6, 7, 10, 11, -1, 12, -1, 13, -1, 14, 15, -1, -1, -1, -1, -1, -1}; return t[it] < 0 ? -1 : 64 + t[it]; }
    if (P == 11) return (it & 1) && it < 16 ? 96 + (it >> 1) : -1;          // block3.2: 0..7 behind iterations 1, 3, .., 15
    if (P == 12) { const int t[16] = {8, 9, 10, -1, 11, 12, -1, 13, 14, -1, 15, -1, -1, -1, -1, -1}; return t[it] < 0 ? -1 : 96 + t[it]; }
    if (P == 15) return it < 4 ? it : (it >= 8 && it < 12) ? 32 + it : -1;  // layer 0's four k steps; block1.2's 8..11
    return -1;
}
// time stamps in program order: (pass, slot, phase) with phase 0 = the barrier's DMA, 1 = the wait in front of the MFMA, 2 = the fetch behind it
constexpr int cw_stamp(int P, int slot, int ph) { return (P * 128 + slot) * 4 + ph; }
constexpr int CW_TILE_STAMPS = 16 * 128 * 4;
// asm loads (weight fetches: 4 each; image DMAs behind the barriers of passes (3, 0..3): 2 each) with a time stamp in (t0, t1); stamps of the previous
// tile are the same minus CW_TILE_STAMPS.  Only asm loads count: the compiler's loads and stores between them make the true number of operations
// in flight larger, so a wait derived from this count is conservative.
// TRAINING form (chain_ws_kernel<8>): every epilogue stores the row tile's 16 x 2 post-activation values per lane as eight 16-B buffer stores, four
// per column tile once the tile's sixteen values are final (tr_group in the kernel: a 4 x 4 transpose inside each quad first, so that a store
// instruction writes 64 consecutive bytes per quad).  They are counted by vmcnt like the loads (gfx9: one in-order counter for both), so the waits
// below must allow for them exactly -- a wait that did not would drain the stores just issued.  cw_train_stores(P, sl): how many of them the
// epilogue piece behind MFMA `sl` of pass P issues (the micro-stage arithmetic of epilogue_piece).
constexpr int cw_train_stores(int P, int sl)
{
    const int S = cw_steps(P >> 2), T = 6 * S, H = T / 2;
    const bool wide = S == 4;
    int n = 0;
    if (sl < H) {                                                             // column tile 0's group: sub-steps 3, 4 (two stores each)
        const int MS = wide ? 17 : 33, PLe = ((P + 15) & 15) >> 2;
        for (int ms = sl * MS / H; ms < (sl + 1) * MS / H; ++ms) {
            if (wide) { if (ms == 15) n += 4; }
            else if (ms == 23 || ms == 25) n += 2;
            if (ms == MS - 1 && PLe != 3) n += 1;                             // the sign words of H[0..2]
        }
    } else {                                                                  // column tile 1's group: micro-stages 1..5 of the second half
        const int PLe = ((P + 15) & 15) >> 2, MS2 = PLe != 3 ? 21 : 35, k2 = sl - H, N2 = T - H;
        for (int ms = k2 * MS2 / N2; ms < (k2 + 1) * MS2 / N2; ++ms)
            if (ms == 4 || ms == 5) n += 2;
    }
    return n;
}
constexpr int cw_train_stores_pass(int P) { int n = 0; for (int sl = 0; sl < 6 * cw_steps(P >> 2); ++sl) n += cw_train_stores(P, sl); return n; }
static_assert(cw_train_stores_pass(0) == 8 && cw_train_stores_pass(4) == 9 && cw_train_stores_pass(5) == 9 && cw_train_stores_pass(9) == 9 && cw_train_stores_pass(15) == 8, "eight activation stores per pass (+ the sign word of a hidden layer's row tile)");
constexpr int cw_asm_between(int t0, int t1, bool trn = false)
{
    int n = 0;
    for (int wrap = -1; wrap <= 0; ++wrap)
        for (int Q = 0; Q < 16; ++Q)
            for (int sl = 0; sl < 6 * cw_steps(Q >> 2); ++sl) {
                if (cw_issue(Q, sl) >= 0) { const int t = cw_stamp(Q, sl, 2) + wrap * CW_TILE_STAMPS; if (t > t0 && t < t1) n += 4; }
                if ((Q >> 2) == 3 && sl == 3 * cw_steps(3)) { const int t = cw_stamp(Q, sl, 0) + wrap * CW_TILE_STAMPS; if (t > t0 && t < t1) n += 2; }
                if (trn && cw_train_stores(Q, sl) > 0) { const int t = cw_stamp(Q, sl, 3) + wrap * CW_TILE_STAMPS; if (t > t0 && t < t1) n += cw_train_stores(Q, sl); }
            }
    return n;
}
// what `s_waitcnt vmcnt` may leave in flight in front of the MFMA (P, slot) that needs block `blk`: the asm loads issued after the block's own
constexpr int cw_need_count(int blk, int P, int slot, bool trn = false)
{
    const int t_need = cw_stamp(P, slot, 1);
    int t_blk = -2 * CW_TILE_STAMPS;
    for (int wrap = -1; wrap <= 0; ++wrap)
        for (int Q = 0; Q < 16; ++Q)
            for (int sl = 0; sl < 6 * cw_steps(Q >> 2); ++sl)
                if (cw_issue(Q, sl) == blk) { const int t = cw_stamp(Q, sl, 2) + wrap * CW_TILE_STAMPS; if (t < t_need && t > t_blk) t_blk = t; }
    return cw_asm_between(t_blk, t_need, trn);
}
// ... in front of the barrier of pass Pw, for the image DMA issued behind the barrier of pass (3, rt)
constexpr int cw_dma_count(int rt, int Pw, bool trn = false)
{
    const int t1 = cw_stamp(Pw, 3 * cw_steps(Pw >> 2), 0);
    int t0 = cw_stamp(12 + rt, 48, 0);
    if (t0 >= t1) t0 -= CW_TILE_STAMPS;
    return cw_asm_between(t0, t1, trn);
}
static_assert(cw_need_count(32 + 0, 4, 48) == 28 && cw_need_count(32 + 7, 4, 90) == 0, "block1.2's k steps 0..7 are fetched behind the first eight iterations of pass (1,0)");
static_assert(cw_need_count(3, 0, 18) == 22 && cw_need_count(0, 0, 0) == 30, "layer 0 fetches: followed by the image DMA of row tile 3, block1.2's 8..11, then by the first fetch of pass (0,0)");
static_assert(cw_dma_count(3, 2) == 28 && cw_dma_count(0, 15) >= 16, "image DMA waits");

// lanes l and l ^ 32 exchange a value: v_permlane32_swap_b32 on two copies leaves the low half's value (lo) and the high half's (hi) in both lanes
__device__ __forceinline__ void cw_halves(float x, float &lo, float &hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}
typedef float cw_f32x4 __attribute__((ext_vector_type(4)));
template <class T> using cw_lptr = __attribute__((address_space(3))) T *;      // LDS pointer (32 bits): what the laundered bases are
template <int DBG>
__global__ __launch_bounds__(256, 1) void chain_ws_kernel(ChainArgs a)
{
    constexpr bool TR = DBG == 8;                                          // training form: every layer's output rows kept (a.H), maxima (a.hmax, a.x5max), table rows through a.row_u
    constexpr int DB = TR ? 0 : DBG;                                      // 1: layer dump, 2: phase clocks, 3..7: timing probes
    constexpr int SLOT = ch_slot(4);
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
#define CW_KEEP(x_) ({ int k_ = (x_); asm volatile("" : "+v"(k_)); k_; })
#define CW_LDS(T_, off_) (*reinterpret_cast<T_ *>(lds + (off_)))
#define CW_AT(T_, ptr_, off_) (*reinterpret_cast<cw_lptr<T_>>((ptr_) + (off_)))
#define CW_AT_F4(ptr_, off_) ({ const cw_f32x4 v_ = CW_AT(const cw_f32x4, ptr_, off_); make_float4(v_[0], v_[1], v_[2], v_[3]); })
    // The laundered values are 32-bit LDS POINTERS (copies of p_*, one v_mov each), refreshed at the start of a pass for the bases that pass uses
    // (CW_REFRESH(P): the running pass's MFMA operands + the epilogue of pass P - 1); a base a pass does not use keeps its older copy.  (Laundered
    // offsets + `lds +` cost a v_mov, a v_add of the zero LDS base and, for the bases a pass did not use, dead copies with hazard pads: ~30
    // instructions at the head of every pass.)
    cw_lptr<char> const lds3 = (cw_lptr<char>)lds;
    cw_lptr<char> const p_b0 = lds3 + o_b0, p_b1 = lds3 + o_b1, p_b2 = lds3 + o_b2, p_pub = lds3 + o_pub, p_ext = lds3 + o_ext, p_exw = lds3 + o_exw, p_exr = lds3 + o_exr,
                        p_cst = lds3 + o_cst, p_dsw = lds3 + o_dsw, p_dsr = lds3 + o_dsr;
    cw_lptr<char> q_b0 = p_b0, q_b1 = p_b1, q_b2 = p_b2, q_pub = p_pub, q_ext = p_ext, q_exw = p_exw, q_exr = p_exr, q_cst = p_cst, q_dsw = p_dsw, q_dsr = p_dsr;
#define CW_KEEPP(x_) ({ cw_lptr<char> k_ = (x_); asm volatile("" : "+v"(k_)); k_; })
#define CW_REFRESH(P_) do { constexpr int pe_ = ((((P_) + 15) & 15) >> 2); \
        q_b0 = CW_KEEPP(p_b0); q_cst = CW_KEEPP(p_cst); \
        if constexpr ((P_) >= 3) { q_b1 = CW_KEEPP(p_b1); q_b2 = CW_KEEPP(p_b2); } \
        if constexpr (pe_ != 3) { q_pub = CW_KEEPP(p_pub); q_exw = CW_KEEPP(p_exw); q_exr = CW_KEEPP(p_exr); } \
        if constexpr (pe_ == 1) q_ext = CW_KEEPP(p_ext); \
        if constexpr (pe_ == 3) { q_dsw = CW_KEEPP(p_dsw); q_dsr = CW_KEEPP(p_dsr); } } while (0)

    const int xcd = blockIdx.x & 7, nb = (gridDim.x + 7 - xcd) / 8, bi = blockIdx.x >> 3;
    const int per = (n_tiles + 7) / 8, t_lo = xcd * per, t_hi = (t_lo + per < n_tiles) ? t_lo + per : n_tiles;
    const bool xcd_order = gridDim.x >= 8;
    const int t_first = xcd_order ? t_lo + bi : (int)blockIdx.x, t_end = xcd_order ? t_hi : n_tiles, t_step = xcd_order ? nb : (int)gridDim.x;
    if (t_first >= t_end) return;

    for (int i = tid; i < 2 * 4 * 32; i += 256)                            // extras k step: its k = 8..15 half stays zero
        CW_LDS(u32x4, 16 * SLOT + (i >> 5) * 1024 + (32 + (i & 31)) * 16) = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < CW_CST_FLOATS; i += 256) CW_LDS(float, CW_CST + 4 * i) = meta[i];
    if constexpr (TR) { for (int i = tid; i < 5 * 128; i += 256) CW_LDS(float, CW_HMAX + 4 * i) = 0.f; }

    asm volatile("" ::: "a255");                                          // the kernel owns all 256 AGPRs (see above)
    u32x4 wx[2][2];                                                        // block3.0's 17th k step (VGPRs, once per tile)
    f32x16 acc[2][2];                                                      // [row-tile parity][column tile]
    u32x4 bf[3][2];                                                        // activation fragment ring [iteration % 3][plane]
    float inv[4] = {0.f, 0.f, 0.f, 0.f};
    int pid[4], pid_n[4] = {0, 0, 0, 0};
    float wq[4], wq_fin = 0.f;
    float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
    // layer 0: rows of the per-point table, two row tiles in flight ([row tile & 1][column tile][16 B]).  Training form: ONE set (32 registers that its kept
    // maxima, offsets and store operands need): a row tile's rows are asked for at the barrier of its own pass, half a pass before their epilogue
    float4 tv[TR ? 1 : 2][2][4];
#define CW_TV(i_) tv[TR ? 0 : ((i_) & 1)]

    // weight fragments of (layer L, k step s) -> a[16 s ..]: four loads  (probe build -DHNR_CHAIN_WS_SAME_W=1: every k step reads the layer's
    // first one -- 64 KiB instead of 848 KiB of weights per tile from L2; results are garbage, the time shows what the weight stream costs)
#ifndef HNR_CHAIN_WS_SAME_W
#define HNR_CHAIN_WS_SAME_W 0
#endif
#define CW_LOAD_W(L_, s_) do { \
        const int so_ = cw_wbase(L_) + (HNR_CHAIN_WS_SAME_W ? 0 : (s_)) * CH_WSTEP; \
        CW_LOAD_FRAG(16 * (s_) + 0, wsrd, woff, so_, 0); CW_LOAD_FRAG(16 * (s_) + 4, wsrd, woff, so_, 1024); \
        CW_LOAD_FRAG(16 * (s_) + 8, wsrd, woff, so_, 2048); CW_LOAD_FRAG(16 * (s_) + 12, wsrd, woff, so_, 3072); } while (0)
    // per-row scalars of a tile: the point ids are needed first (table rows of layer 0: fetched into pid_n two passes before the next tile's
    // first table rows are asked for), the aggregation weights (layer 3) and the extras (layer 1) only later: fetched in pass (0, 1) of their own tile
    auto load_ids = [&](int tile, int (&pd)[4]) __attribute__((always_inline)) {
        const char *aux = a.aux + (size_t)tile * 4 * CH_AUX_GROUP;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if constexpr (TR) { const int u = a.row_u[(size_t)tile * 128 + 32 * rt + j]; pd[rt] = (unsigned)u >= (unsigned)a.ucap ? 0 : u; }     // empty slots / rows past the end carry the sentinel: row 0, like chain_kernel<4, 3>
            else pd[rt] = reinterpret_cast<const int32_t *>(aux + rt * CH_AUX_GROUP)[j];
        }
    };
    auto load_rest = [&](int tile) __attribute__((always_inline)) {
        const char *aux = a.aux + (size_t)tile * 4 * CH_AUX_GROUP;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) wq[rt] = reinterpret_cast<const float *>(aux + rt * CH_AUX_GROUP + 128)[j];
        e0 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32);     // extras of the rows this wave publishes (row tile = wave)
        e1 = *reinterpret_cast<const float4 *>(aux + wave * CH_AUX_GROUP + 256 + j * 32 + 16);
    };
    auto load_table = [&](int pr, float4 (&t)[2][4]) __attribute__((always_inline)) {        // see cw_table_swap: t[column tile][r] = piece r ^ (j & 3) of row (j & ~3) + r
        const int tq = j & 3;
        const int p0 = __builtin_amdgcn_update_dpp(0, pr, 0x00, 0xf, 0xf, false), p1 = __builtin_amdgcn_update_dpp(0, pr, 0x55, 0xf, 0xf, false),
                  p2 = __builtin_amdgcn_update_dpp(0, pr, 0xAA, 0xf, 0xf, false), p3 = __builtin_amdgcn_update_dpp(0, pr, 0xFF, 0xf, 0xf, false);
        const int pp[4] = {p0, p1, p2, p3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *trow = a.ptab + (size_t)(pp[r] < 0 ? 0 : pp[r]) * a.ldt + col0 + 4 * (r ^ tq);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) t[cc][r] = *reinterpret_cast<const float4 *>(trow + 32 * cc);
        }
    };
    // layer-0 operand image of (tile, row tile rt): 8 chunks of 1 KiB (k step s = c >> 1, plane p = c & 1); this wave moves chunks wave and 4 + wave
    // to k-step slots wave >> 1 and 2 + (wave >> 1), plane wave & 1, by DMA
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>(lds);
    const int dma_voff = lane * 16;
    auto dma_image = [&](int tile, int rt) __attribute__((always_inline)) {
        const char *g = a.xp + ((size_t)tile * 4 + rt) * CH_XP_GROUP + wave * 1024;
        const unsigned l0 = lds_base + (unsigned)((wave >> 1) * SLOT + (wave & 1) * 1024 + rt * 2048);
        cw_dma_image(g, g + 4096, dma_voff, l0, l0 + 2 * SLOT);
    };
    // where a tile's weighted sums go: ONE buffer descriptor per tile on the tile's first sample (uniform) + a per-lane byte offset per pass, out of
    // range (the store is dropped) in the lanes that hold no sum and past the end of the tile's class; class 1 / 2 tiles hold 32 / 64 samples of 4 / 2 row slots
    struct TileOut { __amdgpu_buffer_rsrc_t rs, rs_aux; int kc, first, end, t; };
    auto tile_out = [&](int t) __attribute__((always_inline)) {
        TileOut o;
        o.t = t;
        const int tt = t < 0 ? 0 : t;
        o.kc = (tt >= cls.big_tiles ? 1 : 0) + (tt >= cls.big_tiles + cls.small_tiles ? 1 : 0);
        const int f0 = 16 * tt, f1 = cls.n_big + 32 * (tt - cls.big_tiles), f2 = cls.n_big + cls.n_small + 64 * (tt - cls.big_tiles - cls.small_tiles);
        o.first = o.kc == 0 ? f0 : (o.kc == 1 ? f1 : f2);
        o.end = o.kc == 0 ? cls.n_big : (o.kc == 1 ? cls.n_big + cls.n_small : n_valid);
        if (t < 0) { o.first = n_valid; o.end = 0; }
        int n = o.end - o.first; n = n < 0 ? 0 : (n > (16 << o.kc) ? (16 << o.kc) : n);
        o.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.X5) + (size_t)o.first * a.ld5 * 4, 0, n * a.ld5 * 4, 0x00020000);
        // the tile's 128 density inputs (ChainArgs::dsig; chain_sigma_kernel reads them)
        o.rs_aux = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.dsig) + (size_t)tt * 128 * 4, 0, t < 0 ? 0 : 128 * 4, 0x00020000);
        return o;
    };

    // training: one descriptor per kept layer over the whole buffer (the launcher checks rows x row bytes < 2^31: a byte offset with bit 31 set is out
    // of range, which is how the epilogue pieces of "no tile" drop their stores), running maxima of the layers' outputs and of the stored sums
    __amdgpu_buffer_rsrc_t hrs[4];
    float x5m = 0.f, x5_take = 0.f;                                        // (live inside a layer-3 epilogue's second half only)
    float4 tb[4];                                                          // a column tile's sixteen values on their way out (tr_group)
    unsigned mbits = 0u;                                                   // signs of the running epilogue's 32 values (bit 31 - i: value i > 0)
    __amdgpu_buffer_rsrc_t brs;
    if constexpr (TR) {
#pragma unroll
        for (int l = 0; l < 4; ++l) hrs[l] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.H[l]), 0, 0x7fffffff, 0x00020000);
    }
    if constexpr (TR) brs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(a.hbits), 0, a.hbits ? 0x7fffffff : 0, 0x00020000);
    (void)hrs; (void)x5m; (void)x5_take; (void)tb; (void)mbits; (void)brs;
#define CW_HMAX_PUT(l_, v_) asm volatile("ds_max_f32 %0, %1 offset:%2" :: "v"(q_exw), "v"(v_), "n"(CW_HMAX - (int)ch_lds_exch(4) + (l_) * 512) : "memory")

    // ---- prologue: first tile's row scalars, its four layer-0 images (DMA), its first table rows, layer-0 weights + block1.2's k steps 8..11
    load_ids(t_first, pid);
    __syncthreads();                                                       // constants / zeroed extras visible; nobody DMAs into LDS before everybody is here
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) dma_image(t_first, rt);
    load_table(pid[0], tv[0]);
    CW_LOAD_W(0, 0); CW_LOAD_W(0, 1); CW_LOAD_W(0, 2); CW_LOAD_W(0, 3);
    CW_LOAD_W(1, 8); CW_LOAD_W(1, 9); CW_LOAD_W(1, 10); CW_LOAD_W(1, 11);         // (what pass (3,3) of a previous tile would have fetched: cw_issue)
    cw_wait_vm<0>();
    __syncthreads();

    long long t_start = 0, w_start = 0, t_prev = 0, tm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int n_my = 0;
    long long tmf[4] = {0, 0, 0, 0}, tf_prev = 0;
    if (DB >= 2) { t_start = t_prev = clock64(); w_start = wall_clock64(); }

    // uniform scalars of the packed image (SGPRs)
    const float inv0 = __fmul_rn(pow2f(-14), meta[CH_META_DESCALE]), dw1 = meta[CH_META_DESCALE + 1], dw2 = meta[CH_META_DESCALE + 2], dw3 = meta[CH_META_DESCALE + 3],
                alpha_b = meta[4 * 256 + 256];
    (void)alpha_b;
    float amax = 0.f, ap = 0.f, sc_run = 1.f;                               // epilogue state carried between the pieces of one pass
    // constants of the running items: biases (and, for layer 3, alpha weights) as 16-B chunks of two items from the LDS copy, two chunks in flight.
    // A chunk is asked for two items (four micro-stages) before its first use -- the first two chunks of a pass's epilogue by the last
    // micro-stage of the epilogue before it --: in the short layer-0 passes a read asked for one micro-stage ahead was waited for (~100 cycles each).
    float4 bq[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, aq[2] = {bq[0], bq[0]};
    unsigned ph[8], pm[8];
    float s1x = 0.f, s1y = 0.f, s1a = 0.f, s1b = 0.f, s2x = 0.f, s2y = 0.f, s2a = 0.f, s2b = 0.f;     // values handed from an item's first step to its second
    // X5 rows of the row tile whose sums are being stored: the tile's descriptor (tile_out) + ONE per-lane byte offset, out of range (the store
    // is dropped) in the lanes that hold no sum; x5_flag = 1 / 0: the third step of the K-sum runs / is a no-op (4-slot samples)
    int x5_voff = 0x40000000;
    float x5_flag = 1.f, x5_flag2 = 1.f;                                    // (x5_flag2: the second step, a no-op for 2-slot samples)
    float kf[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float4 ex4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto read_cst = [&](int L4, int k) __attribute__((always_inline)) {   // chunk k (items 2 k, 2 k + 1) of constants row L4 (0..3: biases, 4: alpha weights)
        return (DB == 5 || DB == 7) ? make_float4(a.slope, 1.f, a.slope, 1.f) : CW_AT_F4(q_cst, L4 * 1024 + (k >> 2) * 128 + (k & 3) * 16);
    };
    // ---- epilogue of pass PP = (PL, PR), accumulator set se, cut into MICRO-STAGES of 3..8 VALU instructions.  A wave issues in order, and a
    //      dependent VALU instruction issues 8 cycles after its producer, an independent one 4: so every micro-stage holds two independent
    //      chains (the x and the y value of an item), an item's chain is cut in two stages that land behind different MFMAs, and LDS reads
    //      are issued well before their first use.  `slot` counts the 6 S MFMAs of the RUNNING pass, H = first slot of its second
    //      half; micro-stages [slot * MS / H, (slot + 1) * MS / H) run behind MFMA `slot`.  All arguments are constants after unrolling.
    // Training form: the sixteen values of (row tile PR, column tile c) leave as four 16-B stores.  Stored straight from the accumulator layout a quad's
    // four lanes would write to four different rows -- one address-unit cycle per (quad, cache line), 64 per instruction, and the kernel ran 0.60 ms
    // against 0.455 with (meaningless) consecutive addresses.  So the quad transposes first, the way cw_table_swap's loads do in reverse: piece x of
    // lane t ^ x by DPP (sub-steps 0, 1), slot r <- piece r ^ t inside each lane (2), then store r writes row (j & ~3) + r for the whole quad, lane t
    // its piece r ^ t: 64 consecutive bytes per quad (3, 4).  All arguments are constants after unrolling.
    auto tr_group = [&](int PL, int PR, int c, int sub, const TileOut &to) __attribute__((always_inline)) {
        const int se = PR & 1;
        auto dppx = [&](float v, int x) __attribute__((always_inline)) -> float {      // the value of lane t ^ x of the quad (bit pattern moved: the builtin on ints)
            const int i = __builtin_bit_cast(int, v);
            const int r = x == 1 ? __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, false) : x == 2 ? __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, false)
                                                                                             : __builtin_amdgcn_update_dpp(0, i, 0x1B, 0xf, 0xf, false);
            return __builtin_bit_cast(float, r);
        };
        if (sub == 0) {
            tb[0] = make_float4(acc[se][c][0], acc[se][c][1], acc[se][c][2], acc[se][c][3]);
            tb[1] = make_float4(dppx(acc[se][c][4], 1), dppx(acc[se][c][5], 1), dppx(acc[se][c][6], 1), dppx(acc[se][c][7], 1));
        } else if (sub == 1) {
            tb[2] = make_float4(dppx(acc[se][c][8], 2), dppx(acc[se][c][9], 2), dppx(acc[se][c][10], 2), dppx(acc[se][c][11], 2));
            tb[3] = make_float4(dppx(acc[se][c][12], 3), dppx(acc[se][c][13], 3), dppx(acc[se][c][14], 3), dppx(acc[se][c][15], 3));
        } else if (sub == 2) cw_table_swap(tb[0], tb[1], tb[2], tb[3]);
        else {
            // byte offset of (row (j & ~3) of row tile PR, this half's 64 bytes of column tile c) in H[PL]; the lane's row and half are laundered
            // (derived offsets hoisted out of the tile loop cost a register each); "no tile": bit 31 set = out of the descriptor's range
            const int ld4 = a.ldh[PL] * 4;
            const int sb = (to.t < 0 ? (int)0x80000000 : (to.t * 128 + 32 * PR) * ld4 + wave * 256) + 128 * c;
            const int jj = CW_KEEP(j);
            const int hvq = (jj & ~3) * ld4 + (CW_KEEP(h) * 64 + sb), t16 = (jj & 3) * 16;
#pragma unroll
            for (int r = 2 * (sub - 3); r < 2 * (sub - 3) + 2; ++r)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(tb[r].x), __float_as_uint(tb[r].y), __float_as_uint(tb[r].z), __float_as_uint(tb[r].w)}, hrs[PL],
                                                       hvq + r * ld4 + ((16 * r) ^ t16), 0, 0);
        }
    };
    auto epilogue_piece = [&](int PL, int PR, int S, int slot, const TileOut &to) __attribute__((always_inline)) {
        const int T = 6 * S, H = T / 2, se = PR & 1;
        const int exb = (PR & 1) * 512;
        const float inv_l = PL == 0 ? inv0 : inv[PR];
        const int PLn = PR == 3 ? (PL + 1) & 3 : PL;                       // layer of the epilogue that follows this one
        if (slot < H) {
            // -- first half.  Item i = (c, q), two values each, in two steps (st0: bias / table add, descale, slope product; st1: LeakyReLU, row
            //    maximum / alpha partial).  Long passes: micro-stages 2 i, 2 i + 1 = the two steps of item i (3 - 4 VALU behind an MFMA); the short
            //    layer-0 passes (S = 4) hold ~12 VALU per MFMA anyway: there two items go through a step together -- four independent chains instead
            //    of two, every instruction at least four issue slots behind the one it depends on.  Last micro-stage: row maximum / alpha partial ->
            //    exchange buffer
            auto fh_st0 = [&](int it, float &o_x, float &o_y, float &o_a, float &o_b) __attribute__((always_inline)) {
                const int c = it >> 3, q = it & 7, k = it >> 1;
                // scalar fp32 VALU on purpose: packed fp32 instructions (v_pk_fma_f32 ...) do not overlap with this wave's MFMAs -- one
                // of them behind an MFMA costs 18 cycles of matrix-pipe time, a v_fma_f32 none (tools/interleave_probe.hip)
                if (it == 0) { amax = 0.f; ap = 0.f; }
                if ((it & 1) == 0 && k >= 1 && k + 1 < 8) { bq[(k + 1) & 1] = read_cst(PL, k + 1); if (PL == 3) aq[(k + 1) & 1] = read_cst(4, k + 1); }
                float ax = (it & 1) ? bq[k & 1].z : bq[k & 1].x, ay = (it & 1) ? bq[k & 1].w : bq[k & 1].y;
                if (PL == 0) {
                    if (q == 0) cw_table_swap(CW_TV(PR)[c][0], CW_TV(PR)[c][1], CW_TV(PR)[c][2], CW_TV(PR)[c][3]);
                    const float4 t4 = CW_TV(PR)[c][q >> 1];
                    if ((q >> 1) == 0) { ax = cw_table_add<0>((q & 1) ? t4.z : t4.x, ax); ay = cw_table_add<0>((q & 1) ? t4.w : t4.y, ay); }
                    else if ((q >> 1) == 1) { ax = cw_table_add<1>((q & 1) ? t4.z : t4.x, ax); ay = cw_table_add<1>((q & 1) ? t4.w : t4.y, ay); }
                    else if ((q >> 1) == 2) { ax = cw_table_add<2>((q & 1) ? t4.z : t4.x, ax); ay = cw_table_add<2>((q & 1) ? t4.w : t4.y, ay); }
                    else { ax = cw_table_add<3>((q & 1) ? t4.z : t4.x, ax); ay = cw_table_add<3>((q & 1) ? t4.w : t4.y, ay); }
                }
                o_x = fmaf(acc[se][c][2 * q], inv_l, ax); o_y = fmaf(acc[se][c][2 * q + 1], inv_l, ay);
                o_a = __fmul_rn(o_x, a.slope); o_b = __fmul_rn(o_y, a.slope);
            };
            auto fh_st1 = [&](int it, float i_x, float i_y, float i_a, float i_b) __attribute__((always_inline)) {
                const int c = it >> 3, q = it & 7, k = it >> 1;
                const float vx = fmaxf(i_x, i_a), vy = fmaxf(i_y, i_b);
                acc[se][c][2 * q] = vx; acc[se][c][2 * q + 1] = vy;
                if (PL == 3) { ap = fmaf(vx, (it & 1) ? aq[k & 1].z : aq[k & 1].x, ap); ap = fmaf(vy, (it & 1) ? aq[k & 1].w : aq[k & 1].y, ap); }
                else amax = fmaxf(fmaxf(amax, fabsf(vx)), fabsf(vy));
                if constexpr (TR) {
                    if (PL == 3) amax = fmaxf(fmaxf(amax, fabsf(vx)), fabsf(vy));
                    else {                                                      // mbits = mbits * 2 + (v > 0), twice: thirty-two of these push the previous row tile's bits out
                        unsigned long long cy;
                        asm volatile("v_cmp_gt_f32 vcc, %2, 0\n\tv_cmp_gt_f32_e64 %1, %3, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1"
                                     : "+v"(mbits), "=&s"(cy) : "v"(vx), "v"(vy) : "vcc");
                    }
                }
                if (DB == 1) {
                    if (a.dbg && a.dbg_layer == PL && to.t >= 0) {
                        int te = to.t;                                          // laundered: no 64-bit induction variable
                        asm volatile("" : "+s"(te));
                        float *o = a.dbg + ((size_t)te * 128 + 32 * PR + j) * 256 + col0 + 32 * c + 2 * q;
                        o[0] = vx; o[1] = vy;
                    }
                }
            };
            const bool wide = S == 4;
            const int MS = wide ? 17 : 33, m0 = slot * MS / H, m1 = (slot + 1) * MS / H;
#pragma unroll
            for (int ms = m0; ms < m1; ++ms) {
                if (ms < MS - 1) {
                    if (wide) {
                        const int i0 = 2 * (ms >> 1), st = ms & 1;
                        if (st == 0) { fh_st0(i0, s1x, s1y, s1a, s1b); fh_st0(i0 + 1, s2x, s2y, s2a, s2b); }
                        else { fh_st1(i0, s1x, s1y, s1a, s1b); fh_st1(i0 + 1, s2x, s2y, s2a, s2b); }
                    } else {
                        if ((ms & 1) == 0) fh_st0(ms >> 1, s1x, s1y, s1a, s1b); else fh_st1(ms >> 1, s1x, s1y, s1a, s1b);
                    }
                } else {
                    float m;
                    // the two half-waves' partial results meet through v_permlane32_swap (one VALU instruction; a ds_bpermute's wait also drained the
                    // operand reads in flight); both halves then hold the same value and both store it (same address: no exec-masked branch)
                    float v_lo, v_hi;
                    if (PL == 3) {
                        cw_halves(ap, v_lo, v_hi); m = __fadd_rn(v_lo, v_hi); CW_AT(float, q_dsw, PR * 512) = m;
                        if constexpr (TR) { const float am = __fmul_rn(amax, to.t >= 0 ? 1.f : 0.f); CW_HMAX_PUT(3, am); }     // (no tile: garbage times 0; a NaN loses the comparison)
                    }
                    else {
                        cw_halves(amax, v_lo, v_hi);
                        m = fmaxf(v_lo, v_hi);
                        if (PL == 1 && PR == wave)
                            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(e0.x), fabsf(e0.y)), fmaxf(fabsf(e0.z), fabsf(e0.w))), fmaxf(fmaxf(fabsf(e1.x), fabsf(e1.y)), fabsf(e1.z))));
                    }
                    if (PL != 3 && DB != 6 && DB != 7) CW_AT(float, q_exw, exb) = m;
                    if constexpr (TR) { if (PL != 3) CW_HMAX_PUT(PL, m); }
                    if constexpr (TR) {
                        if (PL != 3)                                            // (counted in cw_train_stores; a NULL a.hbits has an empty descriptor: dropped)
                            __builtin_amdgcn_raw_buffer_store_b32(mbits, brs, CW_KEEP(lane) * 4, (int)(PL * a.hbits_stride * 4) + ((to.t * 4 + PR) * 4 + wave) * 256, 0);
                    }
                }
                if constexpr (TR) {                                             // column tile 0 is final after item 7 (micro-stage 15; wide: 7)
                    if (wide) {
                        if (ms == 9) tr_group(PL, PR, 0, 0, to);
                        if (ms == 11) tr_group(PL, PR, 0, 1, to);
                        if (ms == 13) tr_group(PL, PR, 0, 2, to);
                        if (ms == 15) { tr_group(PL, PR, 0, 3, to); tr_group(PL, PR, 0, 4, to); }
                    } else if (ms >= 17 && ms <= 25 && (ms & 1)) tr_group(PL, PR, 0, (ms - 17) >> 1, to);
                }
            }
            return;
        }
        const int k2 = slot - H, N2 = T - H;
        // the next epilogue's first two chunks of constants (its layer is known at compile time); the last micro-stage of either second half
        auto prefetch_next = [&]() __attribute__((always_inline)) {
            bq[0] = read_cst(PLn, 0); bq[1] = read_cst(PLn, 1);
            if (PLn == 3) { aq[0] = read_cst(4, 0); aq[1] = read_cst(4, 1); }
        };
        if (PL != 3) {
            // -- second half, hidden layers: micro-stage 0: exchange read; 1: row scale; 2 + i (i = 0..16): fp16 high parts of item i and low parts
            //    (residuals) of item i - 1, interleaved so that no instruction reads the register written just before it (+ the operand-plane
            //    stores after q = 3, 7); 19: the 7 extra inputs of block3.0; 20: the next epilogue's constants
            const int MS = 21, m0 = k2 * MS / N2, m1 = (k2 + 1) * MS / N2;
#pragma unroll
            for (int ms = m0; ms < m1; ++ms) {
                if constexpr (TR) { if (ms >= 1 && ms <= 5) tr_group(PL, PR, 1, ms - 1, to); }       // column tile 1's values leave
                if (ms == 0) ex4 = (DB == 5 || DB == 7) ? make_float4(amax, 1.f, 2.f, 3.f) : CW_AT_F4(q_exr, exb);
                else if (ms == 1) {
                    const int k = row_scale_exp(fmaxf(fmaxf(ex4.x, ex4.y), fmaxf(ex4.z, ex4.w)));
                    sc_run = pow2f(k);
                    inv[PR] = __fmul_rn(pow2f(-k), PL == 0 ? dw1 : PL == 1 ? dw2 : dw3);
                } else if (ms < 19) {
                    // h = RN16(x 2^k): the product with a power of two is exact, so the fused multiply-convert rounds once, like a separate multiply +
                    // v_cvt_pk_f16_f32; the y value goes to the high half of the same register.  m = RN16(x 2^k - h): the difference is exact in fp32
                    // (h holds the leading 11 bits of x 2^k).  One asm block per stage: the order inside is the point.
                    const int it = ms - 2, ip = it - 1;
                    if (it == 0) {
                        asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0"
                                     : "=&v"(ph[0]) : "v"(acc[se][0][0]), "v"(acc[se][0][1]), "v"(sc_run));
                    } else if (it < 16) {
                        const int c = it >> 3, q = it & 7, cp = ip >> 3, qp = ip & 7;
                        asm volatile("v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
                                     "v_fma_mixlo_f16 %1, %4, %6, -%7 op_sel_hi:[0,0,1]\n\t"
                                     "v_fma_mixhi_f16 %0, %3, %6, 0\n\t"
                                     "v_fma_mixhi_f16 %1, %5, %6, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                     : "=&v"(ph[q]), "=&v"(pm[qp])
                                     : "v"(acc[se][c][2 * q]), "v"(acc[se][c][2 * q + 1]), "v"(acc[se][cp][2 * qp]), "v"(acc[se][cp][2 * qp + 1]), "v"(sc_run), "v"(ph[qp]));
                    } else {
                        asm volatile("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                     : "=&v"(pm[7]) : "v"(acc[se][1][14]), "v"(acc[se][1][15]), "v"(sc_run), "v"(ph[7]));
                    }
                    // operand-plane stores, ONE per stage (two 16-B stores behind one MFMA do not fit its shadow: 13 cycles of data transfer each): the
                    // high parts of items q - 3 .. q are complete after stage q, their low parts one stage later
                    if (it < 16 && ((it & 3) == 3) && !(DB == 6 || DB == 7)) {
                        const int c = it >> 3, q = it & 7;
                        CW_AT(u32x4, q_pub, c * 2 * SLOT + PR * 2048 + (q == 7 ? 512 : 0)) = u32x4{ph[q - 3], ph[q - 2], ph[q - 1], ph[q]};
                    }
                    if (ip >= 0) {
                        const int cp = ip >> 3, qp = ip & 7;
                        if ((qp == 3 || qp == 7) && (DB == 6 || DB == 7)) { asm volatile("" :: "v"(ph[qp]), "v"(pm[qp]), "v"(ph[qp - 1]), "v"(pm[qp - 1]), "v"(ph[qp - 2]), "v"(pm[qp - 2]), "v"(ph[qp - 3]), "v"(pm[qp - 3])); }
                        else if (qp == 3 || qp == 7)
                            CW_AT(u32x4, q_pub, cp * 2 * SLOT + PR * 2048 + (qp == 7 ? 512 : 0) + 1024) = u32x4{pm[qp - 3], pm[qp - 2], pm[qp - 1], pm[qp]};
                    }
                } else if (ms == 19) {
                    if (PL == 1 && PR == wave && h == 0) {
                        unsigned xh[4], xm[4];
                        split2h(__fmul_rn(e0.x, sc_run), __fmul_rn(e0.y, sc_run), xh[0], xm[0]);
                        split2h(__fmul_rn(e0.z, sc_run), __fmul_rn(e0.w, sc_run), xh[1], xm[1]);
                        split2h(__fmul_rn(e1.x, sc_run), __fmul_rn(e1.y, sc_run), xh[2], xm[2]);
                        split2h(__fmul_rn(e1.z, sc_run), 0.f, xh[3], xm[3]);
                        CW_AT(u32x4, q_ext, PR * 2048) = u32x4{xh[0], xh[1], xh[2], xh[3]};
                        CW_AT(u32x4, q_ext, PR * 2048 + 1024) = u32x4{xm[0], xm[1], xm[2], xm[3]};
                        if constexpr (TR) {                                     // X3 = [H2 | colour3 | dir - viewdir | dir . viewdir | 0]  (not in the wait counts: not every wave issues them)
                            const int ev = ((to.t * 128 + 32 * PR + CW_KEEP(j)) * a.ldh[1] + 256) * 4;
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(e0.x), __float_as_uint(e0.y), __float_as_uint(e0.z), __float_as_uint(e0.w)}, hrs[1], ev, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(e1.x), __float_as_uint(e1.y), __float_as_uint(e1.z), 0u}, hrs[1], ev + 16, 0, 0);
                        }
                    }
                } else prefetch_next();
            }
            return;
        }
        // -- second half, layer 3: K-weighted sums (8 adjacent lanes: three DPP steps), X5 / sigma stores.
        //    micro-stage 0: exchange read, store offsets; 1 + 4 d + t: value pairs 2 d, 2 d + 1 (four values), step t = 0: times the row's weight, 1..3: the
        //    three lane steps (a float4 store after the last); 33: sigma; 34: the next epilogue's constants
        const float wq_e = PR == 3 ? wq_fin : wq[PR];
        // a tile of the second / third slot class (hnr_chain_plan): 8 samples of 4 row slots / 16 samples of 2 per row tile -- the sum stops after two
        // DPP steps / one
        const int kc_e = to.kc;
        const int MS = 35, m0 = k2 * MS / N2, m1 = (k2 + 1) * MS / N2;
#pragma unroll
        for (int ms = m0; ms < m1; ++ms) {
            if constexpr (TR) { if (ms >= 1 && ms <= 5) tr_group(PL, PR, 1, ms - 1, to); }
            if (ms == 0) {
                if (PR == 3) ex4 = CW_AT_F4(q_dsr, 0);
                // this row tile's samples: (4 << kc) of them from sample PR (4 << kc) of the tile; lanes past the tile's / class's end are out of the descriptor's range
                const int ls = (j >> (3 - kc_e)) + (4 << kc_e) * PR;
                const bool st = (j & ((8 >> kc_e) - 1)) == 0;
                x5_voff = st ? (ls * a.ld5 + col0) * 4 : 0x40000000;
                x5_flag = kc_e > 0 ? 0.f : 1.f;
                x5_flag2 = kc_e > 1 ? 0.f : 1.f;
                if constexpr (TR) x5m = 0.f;
                if constexpr (TR) x5_take = (st && to.first + CW_KEEP(ls) < to.end) ? 1.f : 0.f;
            } else if (ms < 33) {
                // the sum over a sample's 8 (4, 2) row slots = 8 (4, 2) adjacent lanes: pair swap, quad-pair swap, half-row mirror; the second and
                // third step as f += dpp(f) * flag (exactly f + dpp(f) or f).  Four values per step and one asm block per step: a DPP read sits four
                // instructions (or an MFMA) after the write it reads -- the assembler does not pad inline asm, and the compiler cannot expand the
                // steps into moves / selects / branches
                const int d = (ms - 1) >> 2, t = (ms - 1) & 3, c = d >> 2, e0i = 4 * (d & 3), kb = d & 1;      // two sets of sums: a set is stored while the next one is formed
                if (t == 0)
                    asm volatile("v_mul_f32 %0, %4, %8\n\tv_mul_f32 %1, %5, %8\n\tv_mul_f32 %2, %6, %8\n\tv_mul_f32 %3, %7, %8"
                                 : "=&v"(kf[kb][0]), "=&v"(kf[kb][1]), "=&v"(kf[kb][2]), "=&v"(kf[kb][3])
                                 : "v"(acc[se][c][e0i]), "v"(acc[se][c][e0i + 1]), "v"(acc[se][c][e0i + 2]), "v"(acc[se][c][e0i + 3]), "v"(wq_e));
                else if (t == 1)
                    asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                                 : "+v"(kf[kb][0]), "+v"(kf[kb][1]), "+v"(kf[kb][2]), "+v"(kf[kb][3]));
                else if (t == 2)
                    asm volatile("v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                                 : "+v"(kf[kb][0]), "+v"(kf[kb][1]), "+v"(kf[kb][2]), "+v"(kf[kb][3]) : "v"(x5_flag2));
                else {
                    asm volatile("v_fmac_f32_dpp %0, %0, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                                 "v_fmac_f32_dpp %2, %2, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %3, %4 row_half_mirror row_mask:0xf bank_mask:0xf"
                                 : "+v"(kf[kb][0]), "+v"(kf[kb][1]), "+v"(kf[kb][2]), "+v"(kf[kb][3]) : "v"(x5_flag));
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(kf[kb][0]), __float_as_uint(kf[kb][1]), __float_as_uint(kf[kb][2]), __float_as_uint(kf[kb][3])}, to.rs,
                                                           x5_voff + (32 * c + e0i) * 4, 0, 0);
                    if constexpr (TR) {
                        const float m4 = fmaxf(fmaxf(fabsf(kf[kb][0]), fabsf(kf[kb][1])), fmaxf(fabsf(kf[kb][2]), fabsf(kf[kb][3])));
                        x5m = fmaxf(x5m, __fmul_rn(m4, x5_take));                // lanes that store nothing hold partial sums: times 0 (a NaN from garbage loses the maximum)
                    }
                }
            } else if (ms == 33) {
                if constexpr (TR) CW_HMAX_PUT(4, x5m);
                // the tile's density inputs (the alpha dot of every row = the four waves' partial sums): wave w stores row tile w's 32 values into the
                // tile's slice of ChainArgs::dsig; softplus, the rows' weights and the K-sum are chain_sigma_kernel's (~160 instructions per wave and tile that
                // no MFMA of this kernel could hide: they ran in the 24-MFMA pass (0,0))
                if (PR == 3) {
                    const float d = __fadd_rn(__fadd_rn(ex4.x, ex4.y), __fadd_rn(ex4.z, ex4.w));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), to.rs_aux, h == 0 ? (wave * 32 + j) * 4 : 0x40000000, 0, 0);
                }
            } else prefetch_next();
        }
    };

    auto b_read = [&](int rt, int s, int p) __attribute__((always_inline)) {
        const int off = s * SLOT + (rt * 2 + p) * 1024;
        return off < 65536 ? CW_AT(const u32x4, q_b0, off) : off < 131072 ? CW_AT(const u32x4, q_b1, off - 65536) : CW_AT(const u32x4, q_b2, off - 131072);
    };

    TileOut to_fin = tile_out(-1);                                         // the tile whose last row tile's sums are still due (none yet)
    CW_REFRESH(0);
    bq[0] = read_cst(3, 0); bq[1] = read_cst(3, 1); aq[0] = read_cst(4, 0); aq[1] = read_cst(4, 1);      // the first epilogue piece is (3,3)'s (of no tile: its stores are dropped)
    bf[0][0] = b_read(0, 0, 0); bf[0][1] = b_read(0, 0, 1);                // iteration 0 of the first tile's pass (0,0); later tiles: asked for at the end of pass (3,3)
    for (int tile = t_first; tile < t_end; tile += t_step) {
        ++n_my;
        const int tile_next = tile + t_step;
        const int tile_nx = tile_next < t_end ? tile_next : tile;         // what is fetched ahead (the last tile fetches itself again: no branch in the passes)
        const TileOut to_cur = tile_out(tile);
        CW_REFRESH(0);
        bf[1][0] = b_read(0, 1, 0); bf[1][1] = b_read(0, 1, 1);            // iteration 1 of pass (0,0)
        cw_static_for<16>([&](auto Pc) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value;
            constexpr int L = P >> 2, rt = P & 3, S = cw_steps(L), G0 = cw_goff(P);
            constexpr int PP = (P + 15) & 15, PL = PP >> 2, PR = PP & 3, sm = rt & 1;
            constexpr int T = 6 * S, H = T / 2;
            const TileOut &to_e = P == 0 ? to_fin : to_cur;                // tile of the row tile whose epilogue runs here
            if (DB >= 2) { const long long t_ = clock64(); tm[(P + 15) & 15] += t_ - t_prev; t_prev = t_; if (P == 2) tmf[3] += t_ - tf_prev; if (P == 1) tf_prev = t_; }
            if constexpr (P != 0) CW_REFRESH(P);
            // ---- pass start: loads that ride ahead (each placed where no wait of the following passes lands right behind it)
            if (P == 6) {                                                  // block3.0's 17th k step: used by the last MFMAs of passes (2, 0..3)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        wx[c][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + (c * 2 + p) * 1024, CH_W2 + 16 * CH_WSTEP, 0));
            }
            if (P == 13) load_ids(tile_nx, pid_n);
            if (P == 1) load_rest(tile);                                   // (pass (0, 0) still reads the previous tile's weight of row tile 3: wq_fin)
            __builtin_amdgcn_sched_barrier(0);
            cw_static_for<6 * S>([&](auto kc) __attribute__((always_inline)) {
                // one piece per MFMA: slot = 6 i + 2 g + c (iteration i = k step cw_korder(L, i), term g, column tile c)
                constexpr int slot = decltype(kc)::value, it = slot / 6, s = cw_korder(L, it), g = (slot % 6) >> 1, c = slot & 1;
                if (slot == H) {
                    // DMA'd layer-0 images: row tile 0's must have landed before the barrier of (3,3) (read from that pass's last iterations on), row
                    // tile 3's before the barrier of (0,2); row tiles 1, 2: covered by the wait at the head of (0,0).  Counted in the asm loads this wave
                    // has issued since (cw_dma_count)
                    // (more than 63 operations issued since: the counter cannot hold that many, the DMA has landed)
                    if constexpr (P == 15) { constexpr int d_ = cw_dma_count(0, 15, TR); if constexpr (d_ <= 63) cw_wait_vm<d_>(); }
                    if constexpr (P == 2) { constexpr int d_ = cw_dma_count(3, 2, TR); if constexpr (d_ <= 63) cw_wait_vm<d_>(); }
                    if (DB >= 2 && P == 1) { const long long t_ = clock64(); tmf[0] += t_ - tf_prev; tf_prev = t_; }
                    cw_lds_barrier();
                    if (DB >= 2 && P == 1) { const long long t_ = clock64(); tmf[1] += t_ - tf_prev; tf_prev = t_; }
                    // every wave is past the first k steps of pass (3, rt): the next tile's image of row tile rt may overwrite their operand planes
                    if constexpr (L == 3) dma_image(tile_nx, rt);
                    // layer 0's table rows are asked for one and a half passes before their epilogue: row tile rt + 1 at the barrier of pass
                    // (0, rt) -- the set it goes into was consumed in this pass's first half --, the next tile's row tile 0 at the barrier of (3, 2)
                    if constexpr (!TR && L == 0 && rt < 3) load_table(pid[rt + 1], tv[TR ? 0 : (rt + 1) & 1]);
                    if constexpr (TR && L == 0 && rt >= 1) load_table(pid[rt], tv[0]);
                    if constexpr (P == 14) load_table(pid_n[0], tv[0]);
                    if (DB >= 2 && P == 1) { const long long t_ = clock64(); tmf[2] += t_ - tf_prev; tf_prev = t_; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                constexpr int wp = g == 0 ? 1 : 0, xp_ = g == 1 ? 1 : 0;   // wm*xh, wh*xm, wh*xh: smallest terms first
                constexpr int ring = (G0 + it) % 3;
                // first pass of a layer: wait for the k step's fragments, counted in the asm loads issued after them (cw_need_count).  Pass (0,0):
                // the eight X5 stores of (3,2)'s epilogue also follow layer 0's fetches (a wait that did not allow for them would drain them)
                if constexpr (rt == 0 && g == 0 && c == 0 && s < 16) {
                    constexpr int n_ = cw_need_count(32 * L + s, P, slot, TR) + (P == 0 ? 8 : 0);
                    if constexpr (n_ <= (TR ? 63 : 56)) cw_wait_vm<n_>();
                }
                constexpr int A0 = 16 * (s < 16 ? s : 0) + 4 * wp + 8 * c;
                if constexpr (DB == 4) { if (slot < 2) { acc[sm][c] = f32x16{} + bf[ring][xp_][0]; } }
                else if constexpr (L == 2 && s == 16) cw_mfma_vw(acc[sm][c], wx[c][wp], bf[ring][xp_]);
                else if constexpr (slot < 2) cw_mfma_first<A0>(acc[sm][c], bf[ring][xp_]);
                else cw_mfma<A0>(acc[sm][c], bf[ring][xp_]);
                __builtin_amdgcn_sched_barrier(0);                          // nothing of the epilogue is hoisted above the MFMA (an accumulator is read two MFMAs after its last writer at the earliest)
                // activation fragments two iterations ahead (possibly the next pass's; the next tile's first ones from the last iteration of (3,3))
                if constexpr (g < 2 && c == 0 && G0 + it + 2 < CW_GTOT) {
                    constexpr int i2 = it + 2 >= S ? it + 2 - S : it + 2, P2 = it + 2 >= S ? P + 1 : P;
                    if constexpr (P2 < 16 && i2 < cw_steps((P2 & 15) >> 2)) bf[(G0 + it + 2) % 3][g] = b_read(P2 & 3, cw_korder((P2 & 15) >> 2, i2), g);
                }
                if constexpr (g < 2 && c == 0 && P == 15 && it == 15) bf[0][g] = b_read(0, 0, g);
                // the weight stream (cw_issue)
                { constexpr int blk = cw_issue(P, slot); if constexpr (blk >= 0) CW_LOAD_W(blk >> 5, blk & 31); }
                if (DB != 3) epilogue_piece(PL, PR, S, slot, to_e);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // tile switch
        wq_fin = wq[3]; to_fin = to_cur;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) pid[rt] = pid_n[rt];
    }
    // ---- drain: the last tile's layer-3 epilogue of row tile 3
    CW_REFRESH(0);
    cw_static_for<6 * CH_S0>([&](auto kc) __attribute__((always_inline)) {
        constexpr int slot = decltype(kc)::value;
        if (slot == (6 * CH_S0) / 2) cw_lds_barrier();
        epilogue_piece(3, 3, CH_S0, slot, to_fin);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the last tile's fetch-ahead (DMA into this workgroup's LDS) must not outlive the workgroup
    if constexpr (TR) {
        // one atomic per workgroup and layer (as chain_kernel<4, 3>)
        __syncthreads();
        float *exch = reinterpret_cast<float *>(lds + ch_lds_exch(4));
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            float m = tid < 128 ? CW_LDS(float, CW_HMAX + l * 512 + 4 * tid) : 0.f;
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) exch[4 * l + wave] = m;
        }
        __syncthreads();
        if (tid < 5) {
            const float m = fmaxf(fmaxf(exch[4 * tid], exch[4 * tid + 1]), fmaxf(exch[4 * tid + 2], exch[4 * tid + 3]));
            unsigned *dst = tid < 4 ? (a.hmax ? a.hmax + tid : nullptr) : a.x5max;
            if (dst && m > 0.f) atomicMax(dst, __float_as_uint(m));
        }
    }
    if (DB >= 2 && blockIdx.x == 0 && lane == 0 && a.dbg) {               // block 0: cycles per pass [wave][16]
        long long *o = reinterpret_cast<long long *>(a.dbg) + 4 * 1024 + wave * 16;
        for (int i = 0; i < 16; ++i) o[i] = tm[i];
        if (wave == 0) { long long *o2 = reinterpret_cast<long long *>(a.dbg) + 4 * 1024 + 64; for (int i = 0; i < 4; ++i) o2[i] = tmf[i]; }
    }
    if (DB >= 2 && tid == 0 && a.dbg) {                                   // every block: {cycles, wall ticks, tiles}
        long long *o = reinterpret_cast<long long *>(a.dbg) + 4 * (size_t)blockIdx.x;
        o[0] = clock64() - t_start; o[1] = wall_clock64() - w_start; o[2] = n_my; o[3] = 0;
    }
}

// The samples' densities from the row tiles' density inputs chain_ws_kernel left (ChainArgs::dsig, one float per row): sigma = sum over the sample's row
// slots of softplus(d + alpha_b - 1) x aggregation weight, added in the order of the lane steps the chain kernels use (pairs, quad pairs, halves).
// One 128-thread block per tile, thread = row; raw2out_density: models/aggregators/point_aggregators.py:471-476.
__global__ __launch_bounds__(128) void chain_sigma_kernel(ChainArgs a)
{
    const ChainClasses cls = chain_classes(a.counts, a.cap_samples);
    const int tile = blockIdx.x;
    if (tile >= cls.n_tiles) return;
    const int lane = threadIdx.x & 63, rt = threadIdx.x >> 5, j = lane & 31;
    const char *aux = a.aux + ((size_t)tile * 4 + rt) * CH_AUX_GROUP;
    const float d = a.dsig[(size_t)tile * 128 + threadIdx.x], wq = reinterpret_cast<const float *>(aux + 128)[j];
    const float alpha_b = reinterpret_cast<const float *>(a.wimg + CH_META)[4 * 256 + 256];
    const int kc = chain_tile_class(cls, tile);
    float sg = __fmul_rn(chain_softplus_m1(__fadd_rn(d, alpha_b)), wq);
    sg = __fadd_rn(sg, __builtin_amdgcn_update_dpp(0.f, sg, 0xB1, 0xf, 0xf, false));
    { const float g = __builtin_amdgcn_update_dpp(0.f, sg, 0x4E, 0xf, 0xf, false); sg = __fadd_rn(sg, kc > 1 ? 0.f : g); }
    { const float g = __builtin_amdgcn_update_dpp(0.f, sg, 0x141, 0xf, 0xf, false); sg = __fadd_rn(sg, kc > 0 ? 0.f : g); }
    const int s_sig = chain_tile_first(cls, tile, kc) + (4 << kc) * rt + (j >> (3 - kc));
    if ((j & ((8 >> kc) - 1)) == 0 && s_sig < chain_class_end(cls, kc)) a.sigma[s_sig] = sg;
}

int launch_chain_ws(const ChainArgs &a, int grid, hipStream_t st, int mode)
{
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
        HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ws_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, cw_lds_bytes()));
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
    if (mode == 8) chain_ws_kernel<8><<<grid, 256, cw_lds_bytes(), st>>>(a);
    else if (mode == 2) chain_ws_kernel<2><<<grid, 256, cw_lds_bytes(), st>>>(a);
    else if (mode == 1) chain_ws_kernel<1><<<grid, 256, cw_lds_bytes(), st>>>(a);
    else chain_ws_kernel<0><<<grid, 256, cw_lds_bytes(), st>>>(a);
    HNR_LAUNCH_CHECK();
    const int tiles = (a.cap_samples + 15) / 16 + 2;                        // capacity (each slot class may end in a partial tile): the kernel reads the real tile count from the device counters
    chain_sigma_kernel<<<tiles, 128, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

}  // namespace hnr
