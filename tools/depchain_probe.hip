// Probe: v_mfma_f32_32x32x16_f16 on NC accumulators in turn (NC = 1: every MFMA waits for its predecessor's result; 2: two chains) with K independent
// VALU instructions after each MFMA -- what a single accumulator chain per wave costs (the per-row-tile passes of a narrow weight-stationary MLP).
//   hipcc --offload-arch=gfx950 -O3 tools/depchain_probe.hip -o tools/build/depchain_probe && tools/build/depchain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NC, int K>
__global__ __launch_bounds__(256, 1) void probe(int iters, float *out, long long *cyc)
{
    const int tid = threadIdx.x;
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
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
        for (int g = 0; g < 12; ++g) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g % NC]) : "a"(a[g % 3]), "v"(b[g & 1]));
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(y), "v"(z));
        }
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = c1 - c0;
}
template <int NC, int K> static double run()
{
    float *out; long long *cyc; long long h[4];
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) probe<NC, K><<<256, 256>>>(iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)h[0] / (iters * 12.0);
}
int main()
{
    printf("one accumulator chain, K = 0..7 fillers: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f cycles per MFMA\n", run<1, 0>(), run<1, 1>(), run<1, 2>(), run<1, 3>(), run<1, 4>(), run<1, 5>(), run<1, 6>(), run<1, 7>());
    printf("two accumulator chains, K = 0..7 fillers: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f cycles per MFMA\n", run<2, 0>(), run<2, 1>(), run<2, 2>(), run<2, 3>(), run<2, 4>(), run<2, 5>(), run<2, 6>(), run<2, 7>());
    return 0;
}
