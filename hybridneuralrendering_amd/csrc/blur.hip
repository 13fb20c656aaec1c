// Blur-handling module with pre-defined kernels (the "select the best-matching blur per patch" step of training on blurry
// captures): /root/reference/models/base_rendering_model.py:677-745 (`blur_update_output`, faster_version), called from
// mvs_points_volumetric_model.py:145-146 between the render and the losses.
//
// The rendered batch is a (patch_num*patch_size)^2 grid of rays made of patch_num^2 patches.  Per patch and colour channel the
// reference convolves the patch with each of the N kernels (F.conv2d = cross-correlation, zero padding ks/2, normalised by the
// same convolution of a ones patch so borders are not darkened), appends the un-blurred patch as candidate N, takes the
// candidate with the smallest L1 distance to the ground-truth patch, and puts it back.  Gradients flow through the selected
// convolution only (argmin is piecewise constant).  The reference does this with two F.conv2d calls over [147,1,8,8], a
// 5-D tile/abs/sum, fancy indexing and a Python re-assembly loop; here it is one block per patch.
#include "hnr_common.h"

namespace hnr {

constexpr int BLUR_MAX_PS = 16;      // patch size (8 in every shipped script)
constexpr int BLUR_MAX_N = 31;       // pre-defined kernels (12 or 36 in the scripts)

struct BlurArgs {
    const float *color, *gt;          // [S*S,3] ray layout: ray = row * S + col, S = patch_num * patch_size
    const float *kernels;             // [N, ks, ks]
    int N, ks, pn, ps;
    int n_patches, patch_major;       // patch_major: ray = (patch * ps + y) * ps + x (a rank's whole patches); else the pn x pn grid
    float *out;                       // [S*S,3]
    int32_t *select;                  // [n_patches] chosen candidate (N = un-blurred)
};

__device__ __forceinline__ size_t blur_ray(int p, int y, int x, int pn, int ps, int patch_major)
{
    if (patch_major) return ((size_t)p * ps + y) * ps + x;
    return (size_t)((p / pn) * ps + y) * (pn * ps) + ((p % pn) * ps + x);
}

__global__ __launch_bounds__(256) void blur_select_kernel(BlurArgs a)
{
    __shared__ float s_in[3][BLUR_MAX_PS][BLUR_MAX_PS], s_gt[3][BLUR_MAX_PS][BLUR_MAX_PS];
    __shared__ float s_diff[BLUR_MAX_N + 1];
    __shared__ int s_sel;
    const int p = blockIdx.x;
    const int ps = a.ps, half = a.ks / 2;
    const int npos = 3 * ps * ps;
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        s_in[c][y][x] = a.color[3 * ray + c];
        s_gt[c][y][x] = a.gt[3 * ray + c];
    }
    if (threadIdx.x <= a.N) s_diff[threadIdx.x] = 0.f;
    __syncthreads();
    auto blurred = [&](int n, int c, int y, int x) {
        if (n == a.N) return s_in[c][y][x];
        const float *k = a.kernels + (size_t)n * a.ks * a.ks;
        float acc = 0.f, msk = 0.f;
        for (int dy = 0; dy < a.ks; ++dy) {
            const int yy = y + dy - half;
            if (yy < 0 || yy >= ps) continue;
            for (int dx = 0; dx < a.ks; ++dx) {
                const int xx = x + dx - half;
                if (xx < 0 || xx >= ps) continue;
                const float w = k[dy * a.ks + dx];
                acc = fmaf(s_in[c][yy][xx], w, acc);
                msk += w;
            }
        }
        return acc / msk;
    };
    // L1 distance of every candidate: (candidate, position) pairs over the block, wave partial sums -> LDS atomics
    for (int n = 0; n <= a.N; ++n) {
        float d = 0.f;
        for (int t = threadIdx.x; t < npos; t += blockDim.x) {
            const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
            d += fabsf(blurred(n, c, y, x) - s_gt[c][y][x]);
        }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_diff[n], d);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int n = 1; n <= a.N; ++n) if (s_diff[n] < s_diff[best]) best = n;      // first minimum (torch.argmin)
        s_sel = best;
        a.select[p] = best;
    }
    __syncthreads();
    const int sel = s_sel;
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        a.out[3 * ray + c] = blurred(sel, c, y, x);
    }
}

struct BlurBwdArgs {
    const float *g_out;               // [S*S,3]
    const float *kernels;
    const int32_t *select;
    int N, ks, pn, ps;
    int n_patches, patch_major;
    float *g_in;                      // [S*S,3]
};

__global__ __launch_bounds__(256) void blur_select_bwd_kernel(BlurBwdArgs a)
{
    __shared__ float s_g[3][BLUR_MAX_PS][BLUR_MAX_PS];     // g_out / mask_out of the selected kernel
    const int p = blockIdx.x;
    const int ps = a.ps, half = a.ks / 2;
    const int npos = 3 * ps * ps;
    const int sel = a.select[p];
    const float *k = a.kernels + (size_t)(sel < a.N ? sel : 0) * a.ks * a.ks;
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        float g = a.g_out[3 * ray + c];
        if (sel < a.N) {
            float msk = 0.f;
            for (int dy = 0; dy < a.ks; ++dy) {
                const int yy = y + dy - half;
                if (yy < 0 || yy >= ps) continue;
                for (int dx = 0; dx < a.ks; ++dx) {
                    const int xx = x + dx - half;
                    if (xx >= 0 && xx < ps) msk += k[dy * a.ks + dx];
                }
            }
            g /= msk;
        }
        s_g[c][y][x] = g;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        float acc;
        if (sel >= a.N) {
            acc = s_g[c][y][x];
        } else {
            // out[y0,x0] = sum in[y0+dy-half, x0+dx-half] k[dy,dx]  =>  d in[y,x] = sum_{y0,x0} g[y0,x0] k[y-y0+half, x-x0+half]
            acc = 0.f;
            for (int y0 = 0; y0 < ps; ++y0) {
                const int dy = y - y0 + half;
                if (dy < 0 || dy >= a.ks) continue;
                for (int x0 = 0; x0 < ps; ++x0) {
                    const int dx = x - x0 + half;
                    if (dx < 0 || dx >= a.ks) continue;
                    acc = fmaf(s_g[c][y0][x0], k[dy * a.ks + dx], acc);
                }
            }
        }
        a.g_in[3 * ray + c] = acc;
    }
}

}  // namespace hnr

using namespace hnr;

static int blur_check(int N, int ks, int pn, int ps, const char *who)
{
    if (N < 0 || N > BLUR_MAX_N || ks <= 0 || (ks & 1) == 0 || pn == 0 || ps <= 0 || ps > BLUR_MAX_PS) {
        set_error("%s: unsupported sizes (N <= %d kernels, odd kernel size, patch size <= %d)", who, BLUR_MAX_N, BLUR_MAX_PS);
        return HNR_ERR_BADARG;
    }
    return HNR_OK;
}

extern "C" int hnr_blur_select(const float *d_color, const float *d_gt, const float *d_kernels, int n_kernels, int kernel_size,
                               int patch_num, int patch_size, float *d_out, int32_t *d_select, void *stream)
{
    if (blur_check(n_kernels, kernel_size, patch_num, patch_size, "hnr_blur_select") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_color || !d_gt || (n_kernels > 0 && !d_kernels) || !d_out || !d_select) { set_error("hnr_blur_select: NULL argument"); return HNR_ERR_BADARG; }
    BlurArgs a;
    a.color = d_color; a.gt = d_gt; a.kernels = d_kernels; a.N = n_kernels; a.ks = kernel_size; a.ps = patch_size;
    a.patch_major = patch_num < 0; a.pn = patch_num < 0 ? 1 : patch_num; a.n_patches = patch_num < 0 ? -patch_num : patch_num * patch_num;
    a.out = d_out; a.select = d_select;
    blur_select_kernel<<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_blur_select_bwd(const float *d_g_out, const float *d_kernels, const int32_t *d_select, int n_kernels, int kernel_size,
                                   int patch_num, int patch_size, float *d_g_in, void *stream)
{
    if (blur_check(n_kernels, kernel_size, patch_num, patch_size, "hnr_blur_select_bwd") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_g_out || (n_kernels > 0 && !d_kernels) || !d_select || !d_g_in) { set_error("hnr_blur_select_bwd: NULL argument"); return HNR_ERR_BADARG; }
    BlurBwdArgs a;
    a.g_out = d_g_out; a.kernels = d_kernels; a.select = d_select; a.N = n_kernels; a.ks = kernel_size; a.ps = patch_size;
    a.patch_major = patch_num < 0; a.pn = patch_num < 0 ? 1 : patch_num; a.n_patches = patch_num < 0 ? -patch_num : patch_num * patch_num;
    a.g_in = d_g_in;
    blur_select_bwd_kernel<<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
