// Probe (not part of the product): how much of a SIMD's matrix-pipe throughput survives when a SECOND wave on the same SIMD issues
// VALU / LDS work?  512 threads: waves 0-3 (one per SIMD) run a register-only v_mfma_f32_32x32x16_f16 loop (12 independent MFMAs per
// iteration on 4 accumulators), waves 4-7 (the same SIMDs) run a loop of one instruction kind.  Prints cycles per MFMA (ideal 32)
// and cycles per filler instruction, each alone and together.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/build/coissue_probe && tools/build/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VMODE, int ACC_AGPR, int NOP = 0>
__global__ __launch_bounds__(512, 1) void probe(int it_m, int it_v, float *out, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave < 4) {
        if (it_m <= 0) return;
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        u32x4 a[3], b[2];
        for (int i = 0; i < 3; ++i) for (int e = 0; e < 4; ++e) a[i][e] = 0x3c003c00u + ((tid * 7 + i * 13 + e) & 0xff);
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) b[i][e] = 0x3c003c00u + ((tid * 3 + i * 5 + e) & 0xff);
        const long long c0 = clock64();
#pragma unroll 1
        for (int i = 0; i < it_m; ++i) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                {
                    if (ACC_AGPR && NOP > 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n s_nop %3" : "+a"(acc[t]) : "v"(a[g]), "v"(b[t & 1]), "n"(NOP - 1));
                    else if (ACC_AGPR && NOP < 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n s_nop 15\n s_nop %3" : "+a"(acc[t]) : "v"(a[g]), "v"(b[t & 1]), "n"(-NOP - 1));
                    else if (ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a[g]), "v"(b[t & 1]));
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[g]), __builtin_bit_cast(f16x8, b[t & 1]), acc[t], 0, 0, 0);
                }
        }
        const long long c1 = clock64();
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * 512 + tid] = s;
        if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = c1 - c0;
    } else {
        if (it_v <= 0 || VMODE == 0) return;
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = 1.0f + 0.001f * (float)((tid + i) & 31);
        float y = 1.00001f, z = 1e-7f;
        unsigned lo = (unsigned)(tid & 255) * 16u + 512u * 16u;
        const long long c0 = clock64();
#pragma unroll 1
        for (int i = 0; i < it_v; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if (VMODE == 1) {            // 16 independent v_fma_f32
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
                } else if (VMODE == 2) {     // 16 v_pk_fma_f32 on 8 register pairs
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            f32x2 v = {x[2 * k], x[2 * k + 1]};
                            const f32x2 yy = {y, y}, zz = {z, z};
                            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(yy), "v"(zz));
                            x[2 * k] = v.x; x[2 * k + 1] = v.y;
                        }
                } else if (VMODE == 3) {     // 16 v_max3_f32
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(x[k]) : "v"(y), "v"(z));
                } else if (VMODE == 4) {     // 4 x (cvt_pk, fma_mix, fma_mix, cvt_pk) = the fp16 split
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned ph, pm; float r0, r1;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(x[4 * k]), "v"(x[4 * k + 1]));
                        asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(ph), "v"(x[4 * k]));
                        asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(ph), "v"(x[4 * k + 1]));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pm) : "v"(r0), "v"(r1));
                        x[4 * k + 2] = __uint_as_float(pm); x[4 * k + 3] = r1;
                    }
                } else if (VMODE == 5) {     // 4 ds_write_b128
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x4 v = {__float_as_uint(x[4 * k]), __float_as_uint(x[4 * k + 1]), __float_as_uint(x[4 * k + 2]), __float_as_uint(x[4 * k + 3])};
                        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lo), "v"(v), "n"(0) : "memory");
                    }
                } else if (VMODE == 6) {     // 16 v_add_f32 with a DPP source
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[k]));
                } else if (VMODE == 7) {     // 16 v_mul_f32 + 16 v_max_f32 (LeakyReLU pair)
#pragma unroll
                    for (int k = 0; k < 16; ++k) { float t; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x[k]), "v"(y)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(t)); }
                }
#define ONE_OP(M_, ASM_) else if (VMODE == M_) { _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM_ : "+v"(x[k]) : "v"(y), "v"(z)); }
                ONE_OP(10, "v_mul_f32_e32 %0, %1, %0")
                ONE_OP(11, "v_mul_f32_e64 %0, %1, %0")
                ONE_OP(12, "v_max3_f32 %0, %0, %1, %2")
                ONE_OP(13, "v_cvt_pk_f16_f32 %0, %0, %1")
                ONE_OP(14, "v_fma_mix_f32 %0, -%1, 1.0, %0 op_sel_hi:[1,0,0]")
                ONE_OP(15, "v_fmac_f32_e32 %0, %1, %2")
                ONE_OP(16, "v_cvt_f32_f16_e32 %0, %0")
                ONE_OP(17, "v_and_b32_e32 %0, %1, %0")
                ONE_OP(18, "v_fma_f32 %0, %0, %1, 1.0")
                ONE_OP(19, "v_fma_f32 %0, %0, %0, %0")
                ONE_OP(20, "v_max_f32_e64 %0, |%0|, |%1|")
                else if (VMODE == 30) {      // one chain of 16 dependent v_mul
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[0]) : "v"(y));
                } else if (VMODE == 31) {    // 8 x (v_pk_fma -> dependent v_max3)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        f32x2 v = {x[2 * k], x[2 * k + 1]};
                        const f32x2 yy = {y, y}, zz = {z, z};
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(yy), "v"(zz));
                        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(z) : "v"(v.x), "v"(v.y));
                        x[2 * k] = v.x; x[2 * k + 1] = v.y;
                    }
                } else if (VMODE == 32) {    // independent v_mul separated by s_nop 1
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32_e32 %0, %1, %0\n s_nop 1" : "+v"(x[k]) : "v"(y));
                } else if (VMODE == 33) {    // 16 independent v_mov_b32 dpp
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[k]) : "v"(y));
                } else if (VMODE == 34) {    // pairs: v_mul t <- x ; v_max x <- x, t  but issued as 16 muls then 16 maxes (dependent at distance 16)
                    float t[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t[k]) : "v"(x[k]), "v"(y));
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[k]) : "v"(t[k]));
                } else if (VMODE == 35) {    // split sequence, 4 independent pairs interleaved stage by stage
                    unsigned ph[4], pm[4]; float r0[4], r1[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph[k]) : "v"(x[4 * k]), "v"(x[4 * k + 1]));
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0[k]) : "v"(ph[k]), "v"(x[4 * k]));
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1[k]) : "v"(ph[k]), "v"(x[4 * k + 1]));
#pragma unroll
                    for (int k = 0; k < 4; ++k) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pm[k]) : "v"(r0[k]), "v"(r1[k])); x[4 * k + 2] = __uint_as_float(pm[k]); x[4 * k + 3] = r1[k]; }
                }
                else if (VMODE == 21) {      // 8 v_pk_mul_f32 + 8 v_pk_add_f32 on register pairs
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            f32x2 v = {x[2 * k], x[2 * k + 1]};
                            const f32x2 yy = {y, y};
                            if (r2 == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(yy));
                            else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(yy));
                            x[2 * k] = v.x; x[2 * k + 1] = v.y;
                        }
                }
                else if (VMODE == 8) {     // 4 ds_read_b128
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u32x4 v;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lo), "n"(0) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        x[4 * k] = __uint_as_float(v[0]);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long c1 = clock64();
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += x[i];
        out[blockIdx.x * 512 + tid] = s;
        if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = c1 - c0;
    }
}

template <int VMODE, int AG, int NOP = 0>
static void run(const char *name, int per_iter)
{
    float *out; long long *cyc;
    const int blocks = 256;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    long long h[8];
    const int it_m = 4000;                                           // 48 k MFMAs = 1.5 M cycles
    const int it_v_long = 1 << 15, it_v_short = 400 * 16 / per_iter;   // the short filler must END while the MFMAs (1.5 M cycles) still run
    double res[4] = {0, 0, 0, 0};
    // (a) MFMA alone, (b) filler alone, (c) MFMA with a filler that outlasts it, (d) filler that ends while the MFMAs still run
    for (int pass = 0; pass < 4; ++pass) {
        const int m = (pass == 1) ? 0 : it_m, v = pass == 0 ? 0 : (pass == 2 ? it_v_long : it_v_short);
        hipMemset(cyc, 0, blocks * 8 * 8);
        for (int rep = 0; rep < 2; ++rep) probe<VMODE, AG, NOP><<<blocks, 512, 32768>>>(m, v, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        if (pass == 0) res[0] = (double)h[0] / (it_m * 12.0);
        if (pass == 1) res[1] = (double)h[4] / (it_v_short * 4.0 * per_iter);
        if (pass == 2) res[2] = (double)h[0] / (it_m * 12.0);
        if (pass == 3) res[3] = (double)h[4] / (it_v_short * 4.0 * per_iter);
    }
    printf("nop %3d acc in %s: %-28s cycles/MFMA alone %.1f, with filler %.1f | cycles/filler instr alone %.2f, under MFMAs %.2f\n", NOP, AG ? "AGPR" : "VGPR", name, res[0], res[2], res[1], res[3]);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 1, 5>("v_fma_f32", 16);
    run<1, 1, 6>("v_fma_f32", 16);
    run<1, 1, 7>("v_fma_f32", 16);
    run<1, 0, 0>("v_fma_f32", 16);
    run<4, 1, 6>("split sequential", 16);
    run<4, 1, 7>("split sequential", 16);
    run<5, 1, 6>("ds_write_b128", 4);
    run<5, 1, 7>("ds_write_b128", 4);
    run<8, 1, 7>("ds_read_b128 + wait", 4);
    run<2, 1, 7>("v_pk_fma_f32", 16);
    return 0;
}
