// Cost of the chain kernel's per-point table gather through the CU's vector-memory front end (TA), two lane -> address mappings:
//   A (shipped in r3): lane (h, j) reads 16 B of row j per instruction (a quad's four lanes hit four different rows)
//   B: a quad's four lanes read 64 contiguous bytes of ONE row; rows dealt to instructions (needs a 4x4 quad transpose afterwards)
// hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/build/gather_probe && tools/build/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void gather_kernel(const float *tab, const int *rows, int n_tiles, float *out, long long *cyc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, j = lane & 31;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long long t0 = clock64();
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int *rp = rows + (size_t)t * 128 + rt * 32;
            float4 v[8];
            if (MODE == 0) {
                const float *trow = tab + (size_t)rp[j] * 256 + 64 * wave + 16 * h;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[cc * 4 + q] = *reinterpret_cast<const float4 *>(trow + 32 * cc + 4 * q);
            } else if (MODE == 2) {                                       // C: a PAIR of lanes reads 32 contiguous bytes of one row (one v_swap_b32 stage afterwards)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = i >> 1, g2 = i & 1, b = j & 1;
                        const float *trow = tab + (size_t)rp[(j & ~1) + r] * 256 + 64 * wave + 32 * cc + 16 * h + 4 * (2 * g2 + (b ^ r));
                        v[cc * 4 + i] = *reinterpret_cast<const float4 *>(trow);
                    }
            } else {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float *trow = tab + (size_t)rp[(j & ~3) + i] * 256 + 64 * wave + 32 * cc + 16 * h + 4 * (j & 3);
                        v[cc * 4 + i] = *reinterpret_cast<const float4 *>(trow);
                    }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv)
{
    const int N = 2000000, n_tiles = 64 * 256;
    float *tab; int *rows; float *out; long long *cyc;
    CK(hipMalloc(&tab, (size_t)N * 1024)); CK(hipMemset(tab, 0, (size_t)N * 1024));
    std::vector<int> hr((size_t)n_tiles * 128);
    std::mt19937 g(1);
    // neighbouring rows of the chain kernel are the K neighbours of a sample: random points of a small neighbourhood; here: uniformly random (worst case)
    const int distinct = argc > 1 ? atoi(argv[1]) : N;                    // few distinct rows: L2-resident, shows the front end's own limit
    for (auto &r : hr) r = (int)((g() % distinct) * (long long)(N / distinct));
    CK(hipMalloc(&rows, hr.size() * 4)); CK(hipMemcpy(rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (mode == 0) gather_kernel<0><<<256, 256>>>(tab, rows, n_tiles, out, cyc); else if (mode == 1) gather_kernel<1><<<256, 256>>>(tab, rows, n_tiles, out, cyc); else gather_kernel<2><<<256, 256>>>(tab, rows, n_tiles, out, cyc);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long hc[256]; CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
            double mean = 0; for (int i = 0; i < 256; ++i) mean += hc[i]; mean /= 256;
            printf("mode %c: %.3f ms, %.0f cycles per row tile per workgroup (4 waves x 8 loads), %.1f GB/s\n", "ABC"[mode], ms, mean / (n_tiles / 256 * 4),
                   (double)n_tiles * 128 * 1024 / ms * 1e-6);
        }
    return 0;
}
