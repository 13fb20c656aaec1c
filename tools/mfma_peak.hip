// Probe (not part of the product): sustained rate of v_mfma_f32_32x32x2_f32 with operands in registers and nothing else in
// the loop -- the ceiling the dense layers can reach on this box at the clock the chip sustains under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Same loop with per-lane pseudo-random operands (8 A and 8 B registers, rotated): the sustained rate depends on the operand
// data (switching power), so this is the ceiling for real activations, the constant-operand loop the one for zeros.
__device__ inline float rnd(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)(x & 0xffffff) * (1.f / 8388608.f) - 1.f;       // [-1, 1)
}

__global__ __launch_bounds__(512) void mfma_loop_random(int iters, float *out, long long *cyc)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a[8], b[8];
    const unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 16u;
    for (int i = 0; i < 8; ++i) { a[i] = rnd(seed + i); b[i] = rnd(seed + 8 + i); }
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 3) & 7], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 5) & 7], b[u], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 5) & 7], b[(u + 3) & 7], acc[3], 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float *out, long long *cyc)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) mfma_loop<4><<<cus, 64 * waves>>>(iters, out, cyc);
            else mfma_loop_random<<<cus, 64 * waves>>>(iters, out, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const double flops = (double)cus * waves * iters * 16.0 * 32 * 32 * 2 * 2;
            const double wall_s = (double)h[1] / (wall_khz * 1e3);
            printf("%s operands, CUs %d waves/CU %d: %.3f ms  %.1f TFLOP/s   shader cycles %lld in %.3f ms -> %.3f GHz (clock64 rate)\n", mode ? "random" : "constant", cus, waves, ms,
                   flops / ms / 1e9, h[0], wall_s * 1e3, h[0] / wall_s / 1e9);
        }
    }
    return 0;
}
