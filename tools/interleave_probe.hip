// Probe (not part of the product): one wave per SIMD issues v_mfma_f32_32x32x16_f16 with K independent VALU instructions after each one.
// How many VALU instructions fit under an MFMA of the SAME wave, with the accumulators in VGPRs or in AGPRs?
//   hipcc --offload-arch=gfx950 -O3 tools/interleave_probe.hip -o tools/build/interleave_probe && tools/build/interleave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ACC_AGPR, int W_AGPR, int K, int KIND>
__global__ __launch_bounds__(256, 1) void probe(int iters, float *out, long long *cyc)
{
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 a[3], b[2];
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 4; ++e) a[i][e] = 0x3c003c00u + ((tid * 7 + i * 13 + e) & 0xff);
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) b[i][e] = 0x3c003c00u + ((tid * 3 + i * 5 + e) & 0xff);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 0.001f * (float)((tid + i) & 31);
    float y = 1.00001f, z = 1e-7f;
    const long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (ACC_AGPR && W_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "a"(a[g]), "v"(b[t & 1]));
                else if (ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a[g]), "v"(b[t & 1]));
                else if (W_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "a"(a[g]), "v"(b[t & 1]));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a[g]), "v"(b[t & 1]));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(y), "v"(z));
                    else if (KIND == 1) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x[k & 7]) : "v"(y));
                    else if (KIND == 2) { f32x2 v = {x[(2 * k) & 7], x[(2 * k + 1) & 7]}; const f32x2 yy = {y, y}, zz = {z, z}; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(yy), "v"(zz)); x[(2 * k) & 7] = v.x; x[(2 * k + 1) & 7] = v.y; }
                    else if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[k & 7]) : "v"(y));
                    else if (KIND == 4) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[k & 7]) : "v"(y));
                    else if (KIND == 5) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(x[k & 7]) : "v"(y), "v"(z));
                    else if (KIND == 6) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x[k & 7]) : "v"(y));
                    else if (KIND == 7) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[k & 7]));
                    else if (KIND == 8) { u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])}; asm volatile("ds_write_b128 %0, %1" :: "v"(tid * 16), "v"(v) : "memory"); }
                    else if (KIND == 9) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(tid * 16) : "memory"); x[k & 7] = __uint_as_float(v[0]); }
                    else if (KIND == 10) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[k & 7]) : "v"(y), "v"(z));
                    else if (KIND == 11) asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(x[k & 7]) : "v"(y));
                }
            }
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = c1 - c0;
}

template <int AG, int WG, int K, int KIND> static double run()
{
    float *out; long long *cyc; long long h[4];
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) probe<AG, WG, K, KIND><<<256, 256>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)h[0] / (iters * 12.0);
}
template <int AG, int WG, int KIND> static void row(const char *name)
{
    printf("%-34s cycles per MFMA with K VALU after each, K = 0..8: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", name, run<AG, WG, 0, KIND>(), run<AG, WG, 1, KIND>(),
           run<AG, WG, 2, KIND>(), run<AG, WG, 3, KIND>(), run<AG, WG, 4, KIND>(), run<AG, WG, 5, KIND>(), run<AG, WG, 6, KIND>(), run<AG, WG, 7, KIND>(), run<AG, WG, 8, KIND>());
}
int main()
{
    row<0, 1, 4>("acc VGPR, A AGPR, v_fma_mix_f32");
    row<0, 1, 5>("acc VGPR, A AGPR, v_max3_f32 |.|");
    row<0, 1, 6>("acc VGPR, A AGPR, v_max_f32_e32");
    row<0, 1, 7>("acc VGPR, A AGPR, v_add_f32 dpp");
    row<0, 1, 8>("acc VGPR, A AGPR, ds_write_b128");
    row<0, 1, 9>("acc VGPR, A AGPR, ds_read_b128");
    row<0, 1, 10>("acc VGPR, A AGPR, v_fmac_f32_e32");
    row<0, 1, 11>("acc VGPR, A AGPR, v_mul_f32_e64");
    return 0;
}
