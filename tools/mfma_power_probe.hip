// Probe (not part of the product): which ingredient of the weight-stationary split-bf16 kernel costs the clock?  A register-only
// v_mfma_f32_32x32x16_bf16 loop sustains ~1.87 GHz with split-plane operand data; the kernel's own MFMA stream runs at 1.4-1.7.
// Variants add, one at a time: B operands cycling through NB register fragments (the kernel keeps 72 resident), A fragments
// re-read from LDS every 12 MFMAs, VALU filler work, streaming global loads / stores.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/build/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline unsigned rnd_bf16x2(unsigned s) { const unsigned r = rnd(s); return (r & 0x80ff80ffu) | 0x3f003f00u; }

template <int NB, int LDSA, int VALU, int MEM>
__global__ __launch_bounds__(256, 1) void probe(int iters, float *out, long long *cyc, const float4 *src, float4 *dst)
{
    extern __shared__ char lds[];
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 b[NB], a[3];
    const unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 1024u;
    for (int i = 0; i < NB; ++i) for (int e = 0; e < 4; ++e) b[i][e] = rnd_bf16x2(seed + 4 * i + e);
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 4; ++e) a[i][e] = rnd_bf16x2(seed + 999 + 4 * i + e);
    u32x4 *la = reinterpret_cast<u32x4 *>(lds) + threadIdx.x;
    for (int i = 0; i < 24; ++i) la[i * 256] = a[i % 3];
    __syncthreads();
    float vx = (float)threadIdx.x, vy = 1.0001f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 ld = make_float4(0.f, 0.f, 0.f, 0.f), ld2 = ld;
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < NB / 6; ++g) {                 // one "k step": 6 B fragments (2 tiles x 3 planes), 12 MFMAs
            if (LDSA) { a[0] = la[((g * 3 + 0) % 24) * 256]; a[1] = la[((g * 3 + 1) % 24) * 256]; a[2] = la[((g * 3 + 2) % 24) * 256]; }
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, a[0]), Am = __builtin_bit_cast(bf16x8, a[1]), Al = __builtin_bit_cast(bf16x8, a[2]);
#define B_(j) __builtin_bit_cast(bf16x8, b[6 * g + (j)])
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, B_(0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, B_(3), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(2), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(5), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B_(1), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B_(4), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B_(0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, B_(3), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(1), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(4), acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, B_(3), acc[1], 0, 0, 0);
#undef B_
            if (VALU) {
#pragma unroll
                for (int v = 0; v < VALU; ++v) { vx = fmaf(vx, vy, 0.5f); vy = vy * 0.99999f + 1e-6f; }
            }
            if (MEM == 1 && (g & 3) == 0) {
                dst[gi] = ld;                               // the value loaded one round earlier: no wait on the load just issued
                ld = src[gi];
                gi += stride; if (gi >= ((size_t)1 << 26)) gi -= ((size_t)1 << 26);
            }
            // store-only variants, one store instruction per wave per 12 MFMAs (the kernel: 16 per 192), register data, no load
            if (MEM == 2) { reinterpret_cast<float2 *>(dst)[gi] = make_float2(vx, vy); gi += stride; if (gi >= ((size_t)1 << 26)) gi -= ((size_t)1 << 26); }
            if (MEM == 3) { dst[gi] = make_float4(vx, vy, vx, vy); gi += stride; if (gi >= ((size_t)1 << 26)) gi -= ((size_t)1 << 26); }
            if (MEM == 4) { reinterpret_cast<float *>(dst)[gi] = vx; gi += stride; if (gi >= ((size_t)1 << 26)) gi -= ((size_t)1 << 26); }
            // load-only: one 16-B load per lane per 24 MFMAs (the kernel: 8 per 192), consumed two rounds later
            if (MEM == 5 && (g & 1) == 0) { vx += ld.x; ld = ld2; ld2 = src[gi]; gi += stride; if (gi >= ((size_t)1 << 26)) gi -= ((size_t)1 << 26); }
        }
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    float s = vx + vy + ld.x + ld2.y;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

template <int NB, int LDSA, int VALU, int MEM>
static void run(const char *name, int iters, float *out, long long *cyc, float4 *src, float4 *dst, int cus, int wall_khz)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        probe<NB, LDSA, VALU, MEM><<<cus, 256, 24 * 256 * 16>>>(iters, out, cyc, src, dst);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double flops = (double)cus * 4 * iters * (NB / 6) * 12.0 * 32 * 32 * 16 * 2;
        const double wall_s = (double)h[1] / (wall_khz * 1e3);
        const double util = (double)iters * (NB / 6) * 12.0 * 32.0 / (double)h[0];
        if (rep) printf("%-62s %8.3f ms  %7.1f TFLOP/s  %.3f GHz  MFMA pipe %.0f %%\n", name, ms, flops / ms / 1e9, h[0] / wall_s / 1e9, 100.0 * util);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; long long *cyc; float4 *src, *dst;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    (void)hipMalloc(&src, (size_t)1 << 30); (void)hipMalloc(&dst, (size_t)1 << 30);
    (void)hipMemset(src, 0, (size_t)1 << 30);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    run<6, 0, 0, 0>("6 B fragments in registers, A fixed", iters * 12, out, cyc, src, dst, cus, wall_khz);
    run<72, 0, 0, 0>("72 B fragments in registers (288 VGPR/AGPR), A fixed", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 0, 0>("72 B fragments + A fragments from LDS every 12 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 12, 0>("  + 24 VALU per 12 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 12, 1>("  + one 16-B load and store per lane per 48 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 0, 2>("72 B frags + LDS A + one 8-B store per lane per 12 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 0, 3>("72 B frags + LDS A + one 16-B store per lane per 12 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 0, 4>("72 B frags + LDS A + one 4-B store per lane per 12 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    run<72, 1, 0, 5>("72 B frags + LDS A + one 16-B load per lane per 24 MFMAs", iters, out, cyc, src, dst, cus, wall_khz);
    return 0;
}
