#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// feasibility: can hipcc keep 18 + 8 + 8 + 8 k steps of weight fragments (2 planes x 16 B per lane each = 336 registers) resident across a tile loop?
__global__ __launch_bounds__(256, 1) void ws_probe(const u32x4 *__restrict__ wimg, const char *__restrict__ xin, float *__restrict__ out, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 w0[18][2], w1[8][2], w2[8][2], w3[8][2];
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) w0[s][p] = wimg[((s * 4 + wave) * 2 + p) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            w1[s][p] = wimg[(((18 + s) * 4 + wave) * 2 + p) * 64 + lane];
            w2[s][p] = wimg[(((26 + s) * 4 + wave) * 2 + p) * 64 + lane];
            w3[s][p] = wimg[(((34 + s) * 4 + wave) * 2 + p) * 64 + lane];
        }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // stand-in prologue: copy a tile's planes into LDS
        for (int i = threadIdx.x; i < 18 * 4096 / 16; i += 256) reinterpret_cast<u32x4 *>(lds)[i] = reinterpret_cast<const u32x4 *>(xin + (size_t)tile * 18 * 4096)[i];
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        auto layer = [&](auto &w, int S, int base) {
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (s >= S) break;
                u32x4 x[2][2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) x[rt][p] = *reinterpret_cast<const u32x4 *>(lds + base + s * 4096 + (rt * 2 + p) * 1024 + lane * 16);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[s][1]), __builtin_bit_cast(f16x8, x[rt][0]), acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[s][0]), __builtin_bit_cast(f16x8, x[rt][1]), acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[s][0]), __builtin_bit_cast(f16x8, x[rt][0]), acc[rt], 0, 0, 0);
            }
        };
        layer(w0, 18, 0);
        __syncthreads();
        layer(w1, 8, 0);
        __syncthreads();
        layer(w2, 8, 0);
        __syncthreads();
        layer(w3, 8, 0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((size_t)tile * 2 + rt) * 16 * 256 + r * 256 + threadIdx.x] = acc[rt][r];
        __syncthreads();
    }
}
