// Shared pieces of the "f16x2" dense arithmetic (csrc/chain.hip, csrc/mlp.hip): fp32 operands split into two fp16 terms under exact
// power-of-two scales, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation.  See the header of chain.hip.
#pragma once
#include <type_traits>

#include "hnr_common.h"

namespace hnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_ACT_EXP = 15;                     // a row's maximum is scaled into [2^14, 2^15)
constexpr int CH_W_EXP = 14;                       // a layer's largest weight is scaled into [2^13, 2^14)

// (x0, x1) -> packed fp16 pairs h, m with x = h + m + O(2^-22 |x|); round-to-nearest-even
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &ph, unsigned &pm)
{
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(ph), "v"(x0));                  // x0 - h.lo (exact)
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(ph), "v"(x1));    // x1 - h.hi
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pm) : "v"(r0), "v"(r1));
}

__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }   // -126 <= e <= 127

// scale exponent k of a row whose largest |value| is m: m * 2^k in [2^(CH_ACT_EXP-1), 2^CH_ACT_EXP)
__device__ __forceinline__ int row_scale_exp(float m)
{
    int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);          // biased exponent; m >= 0
    ex = ex < 48 ? 48 : (ex > 250 ? 250 : ex);                   // zero / tiny rows: scale 2^93 at most; inf / nan rows: garbage in, garbage out
    return CH_ACT_EXP + 126 - ex;
}

// A PAIR of floats with scalar arithmetic -- deliberately not ext_vector_type(2): vector float arithmetic (and the SLP vectoriser, hence
// -fno-slp-vectorize in the Makefile) becomes v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, and on gfx950 those give wrong results in lanes 48..63
// of a wave while ANOTHER wave of the same SIMD runs MFMAs (another stream's GEMM, another process's frame): tools/featmap_contention.py reproduces
// it in 764 of 2000 launches with the packed form and in 0 of 2000 without (profiles/README.md, round 4).  tests/test_abi_and_host.py checks that the
// built library contains none.
struct f32x2 { float x, y; };
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y)}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)}; }
__device__ __forceinline__ f32x2 f32x2_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }

// one dense layer of the tile: acc[rt][c] (+)= W[64 wave + 32 c .. +31, :] * X[32 rt .. +31, :]^T over S k steps.
// PD = prefetch distance of the weight fragments in k steps (ring of PD + 1); PRELOAD_ALL: all S steps up front (layer 0).
// woff = this lane's byte offset inside a k step of the weight image (first column tile of the wave + lane * 16), WSTEP = bytes per k step
// of the image ([column tile][plane 2][64 lanes][16 B]), SLOT = LDS bytes per k step of the activation planes ([row tile][plane 2][1 KiB]).
// BAR >= 0: a bare s_barrier (rendezvous only, no memory wait) after k step BAR -- the dual-group chain kernel pairs it with a barrier of the
// group that runs its epilogue meanwhile.
// PACE > 0 (dual-group kernel): every MFMA is followed by s_nop PACE - 1 (one less when a load is issued in the same slot).  A wave whose next
// MFMA waits at the issue stage for the matrix pipe blocks the SIMD's VALU port for every other wave (tools/coissue_probe.hip: the
// co-resident wave is starved completely); parked in s_nop for the ~24 cycles the pipe is busy anyway, it leaves the port to the other
// group's epilogue (6.2 instead of 5.0 cycles per VALU instruction there, 34.0 instead of 32.0 cycles per MFMA here).
// after(): called once, right after the layer's LAST weight-fragment loads have been issued: loads placed there (gathered addends ...) are
// not waited for by any later weight wait of the layer (vmcnt counts in order), yet run under the remaining k steps' MFMAs.
struct H2NoHook { __device__ __forceinline__ void operator()() const {} };
template <int RT, int CT, int S, int PRELOAD_ALL, int WSTEP, int SLOT, int BAR = -1, int WSAME = 0, int PACE = 0, int PDO = 0, class Mid, class After = H2NoHook>
__device__ __forceinline__ void h2_mfma_layer(__amdgpu_buffer_rsrc_t wsrd, int wbase, unsigned woff, const char *lds, int lane, f32x16 (&acc)[RT][CT], Mid mid, After after = After())
{
    constexpr int PD = PDO > 0 ? PDO : (RT * CT >= 8 ? 2 : 3);          // k steps of weight fragments in flight
    // fragment (s, column tile c of the wave, plane p) at s * WSTEP + (c * 2 + p) * 1024 + woff of the layer image; the per-lane part
    // is ONE 32-bit offset beside the uniform buffer descriptor, so no load needs a 64-bit address register pair
    asm volatile("" : "+s"(wbase));                                       // per-tile opaque: the k-step offsets are s_add'ed here, not hoisted out of the tile loop (SGPR spills)
    const char *bp = lds + lane * 16;                                     // fragment (s, rt, plane p) at s * SLOT + (rt * 2 + p) * 1024
    constexpr int NW = PRELOAD_ALL ? S : PD + 1;
    u32x4 wf[NW][CT][2], bf[2][RT][2];
    auto load_w = [&](int slot, int s) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) wf[slot][c][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + (c * 2 + p) * 1024, wbase + (WSAME ? 0 : s) * WSTEP, 0));   // WSAME: probe only (every k step re-reads step 0: L1 hits)
    };
    auto load_b = [&](int slot, int s) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[slot][rt][p] = *reinterpret_cast<const u32x4 *>(bp + s * SLOT + (rt * 2 + p) * 1024);
    };
    if (PRELOAD_ALL) {
#pragma unroll
        for (int s = 0; s < S; ++s) load_w(s, s);
    } else {
#pragma unroll
        for (int s = 0; s < PD && s < S; ++s) load_w(s, s);
    }
    __builtin_amdgcn_sched_barrier(0);
    mid();                                                                // loads the caller wants queued BEHIND the first weight fragments
    __builtin_amdgcn_sched_barrier(0);
    load_b(0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (PACE == 0) {
            if (!PRELOAD_ALL && s + PD < S) load_w((s + PD) % (PD + 1), s + PD);
            if (s + 1 < S) load_b((s + 1) & 1, s + 1);
        }
        if (s == (PRELOAD_ALL || S <= PD ? 0 : S - 1 - PD)) { __builtin_amdgcn_sched_barrier(0); after(); __builtin_amdgcn_sched_barrier(0); }
        const int ws = PRELOAD_ALL ? s : s % (PD + 1), bs = s & 1;
#define CH_W(c, p) __builtin_bit_cast(f16x8, wf[ws][c][p])
#define CH_X(rt, p) __builtin_bit_cast(f16x8, bf[bs][rt][p])
        if (PACE > 0) {
            // fixed issue order: MFMA, pause, (one load of a later k step), fence
            const bool has_b = s + 1 < S, has_w = !PRELOAD_ALL && s + PD < S;
#pragma unroll
            for (int m = 0; m < 3 * RT * CT; ++m) {
                const int term = m / (RT * CT), rt = (m % (RT * CT)) / CT, c = m % CT;
                const int wp = term == 0 ? 1 : 0, xp = term == 1 ? 1 : 0;             // wm*xh, wh*xm, wh*xh
                acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CH_W(c, wp), CH_X(rt, xp), acc[rt][c], 0, 0, 0);
                bool load = false;
                if (has_b && m < 2 * RT) {
                    const int r2 = m >> 1, p2 = m & 1;
                    bf[(s + 1) & 1][r2][p2] = *reinterpret_cast<const u32x4 *>(bp + (s + 1) * SLOT + (r2 * 2 + p2) * 1024);
                    load = true;
                } else if (has_w && m >= 2 * RT && m < 2 * RT + 2 * CT) {
                    const int i2 = m - 2 * RT, c2 = i2 >> 1, p2 = i2 & 1;
                    wf[(s + PD) % (PD + 1)][c2][p2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, woff + (c2 * 2 + p2) * 1024, wbase + (WSAME ? 0 : s + PD) * WSTEP, 0));
                    load = true;
                }
                if (load) asm volatile("s_nop %0" :: "n"(PACE > 1 ? PACE - 2 : 0)); else asm volatile("s_nop %0" :: "n"(PACE > 0 ? PACE - 1 : 0));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        // smallest terms first; RT * CT independent accumulators between two MFMAs on the same one
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CH_W(c, 1), CH_X(rt, 0), acc[rt][c], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CH_W(c, 0), CH_X(rt, 1), acc[rt][c], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(CH_W(c, 0), CH_X(rt, 0), acc[rt][c], 0, 0, 0);
        // issue order inside the k step: the fragment reads of step s+1 and the 4 weight loads of step s+PD go out under the
        // FIRST MFMAs (left alone, hipcc sinks the reads to the end of the step and the next step's first MFMA waits for LDS)
        if (s + 1 < S) {
#pragma unroll
            for (int i = 0; i < 2 * RT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
        if (!PRELOAD_ALL && s + PD < S) {
#pragma unroll
            for (int i = 0; i < 2 * CT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        }
#undef CH_W
#undef CH_X
        __builtin_amdgcn_sched_barrier(0);                                // keep the prefetch distance: no load of a later k step is hoisted across
        if (s == BAR) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
}

}  // namespace hnr
