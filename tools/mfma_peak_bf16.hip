// Probe (not part of the product): sustained rate of v_mfma_f32_32x32x16_bf16 with operands in registers and nothing else in
// the loop, with constant (zero-toggling) and per-lane pseudo-random operands -- the matrix-pipe ceiling of the split-bf16
// dense layer (csrc/linear_s3.hip) at the clock the chip sustains under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o tools/build/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned rnd(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two bf16 values in [-2, 2) with random mantissas
__device__ inline unsigned rnd_bf16x2(unsigned s) { const unsigned r = rnd(s); return (r & 0x80ff80ffu) | 0x3f003f00u; }

// RANDOM == 2: operands as the split-bf16 dense layer feeds them -- the h / m / l planes of normally distributed fp32 values
// (8 values per lane and plane), paired like the layer's six product terms
__device__ inline float gauss(unsigned s)
{
    const float u1 = (float)((rnd(s) >> 8) + 1) * (1.f / 16777217.f), u2 = (float)(rnd(s ^ 0x9e3779b9u) >> 8) * (1.f / 16777216.f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}
__device__ inline unsigned bf16_rne(float v) { unsigned u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; }
__device__ inline void split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    h = bf16_rne(x); const float r1 = x - __uint_as_float(h << 16);
    m = bf16_rne(r1); const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rne(r2);
}

template <int RANDOM, int DEP>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float *out, long long *cyc)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 a[4], b[4];
    const unsigned seed = (blockIdx.x * 512u + threadIdx.x) * 64u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            a[i][e] = RANDOM ? rnd_bf16x2(seed + 8 * i + e) : 0x3f803f80u;
            b[i][e] = RANDOM ? rnd_bf16x2(seed + 8 * i + 4 + e) : 0x3f803f80u;
        }
    if (RANDOM == 2) {
        // a[0..2] = h, m, l planes of 8 activations (scale 1), b[0..2] = planes of 8 weights (scale 1/16); a[3] / b[3] = second h planes
        for (int e = 0; e < 4; ++e) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3(gauss(seed + 2 * e), h0, m0, l0); split3(gauss(seed + 2 * e + 1), h1, m1, l1);
            a[0][e] = h0 | (h1 << 16); a[1][e] = m0 | (m1 << 16); a[2][e] = l0 | (l1 << 16);
            split3(gauss(seed + 100 + 2 * e) * 0.0625f, h0, m0, l0); split3(gauss(seed + 101 + 2 * e) * 0.0625f, h1, m1, l1);
            b[0][e] = h0 | (h1 << 16); b[1][e] = m0 | (m1 << 16); b[2][e] = l0 | (l1 << 16);
            split3(gauss(seed + 200 + 2 * e), h0, m0, l0); split3(gauss(seed + 201 + 2 * e), h1, m1, l1);
            a[3][e] = h0 | (h1 << 16);
            split3(gauss(seed + 300 + 2 * e) * 0.0625f, h0, m0, l0); split3(gauss(seed + 301 + 2 * e) * 0.0625f, h1, m1, l1);
            b[3][e] = h0 | (h1 << 16);
        }
    }
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bf16x8 A0 = __builtin_bit_cast(bf16x8, a[u]), A1 = __builtin_bit_cast(bf16x8, a[(u + 1) & 3]);
            const bf16x8 B0 = __builtin_bit_cast(bf16x8, b[u]), B1 = __builtin_bit_cast(bf16x8, b[(u + 3) & 3]);
            if (RANDOM == 2) {   // the layer's six terms on two alternating accumulators: l*h, h*l, m*m, m*h, h*m, h*h
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, a[0]), Am = __builtin_bit_cast(bf16x8, a[1]), Al = __builtin_bit_cast(bf16x8, a[2]);
                const bf16x8 Wh = __builtin_bit_cast(bf16x8, b[0]), Wm = __builtin_bit_cast(bf16x8, b[1]), Wl = __builtin_bit_cast(bf16x8, b[2]);
                const bf16x8 Vh = __builtin_bit_cast(bf16x8, b[3]);
                if (u == 0) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Wh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Vh, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Wl, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Wl, acc[1], 0, 0, 0);
                } else if (u == 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Wm, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Wm, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Wh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Vh, acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Wm, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Wm, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Wh, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Vh, acc[1], 0, 0, 0);
                }
            } else if (DEP) {       // six dependent MFMAs per accumulator, two accumulators alternating (the dense layer's order)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[3], 0, 0, 0);
            }
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

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    float *out; long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode)
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            switch (mode) {
            case 0: mfma_loop<0, 0><<<cus, 64 * waves>>>(iters, out, cyc); break;
            case 1: mfma_loop<1, 0><<<cus, 64 * waves>>>(iters, out, cyc); break;
            case 2: mfma_loop<0, 1><<<cus, 64 * waves>>>(iters, out, cyc); break;
            case 3: mfma_loop<1, 1><<<cus, 64 * waves>>>(iters, out, cyc); break;
            default: mfma_loop<2, 1><<<cus, 64 * waves>>>(iters, out, cyc); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            const double flops = (double)cus * waves * iters * 16.0 * 32 * 32 * 16 * 2;
            const double wall_s = (double)h[1] / (wall_khz * 1e3);
            printf("%s operands, %s, CUs %d waves/CU %d: %.3f ms  %.1f TFLOP/s   %.3f GHz (clock64 / wall clock)\n", mode == 4 ? "split-plane (h/m/l of normal values)" : (mode & 1) ? "random" : "constant",
                   mode >= 2 ? "2 dependent chains" : "4 independent accumulators", cus, waves, ms, flops / ms / 1e9, h[0] / wall_s / 1e9);
        }
    }
    return 0;
}
