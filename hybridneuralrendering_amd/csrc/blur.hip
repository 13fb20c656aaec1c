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

// One 1024-thread block per patch; the (candidate, position) pairs -- 13 x 192 in the shipped configuration -- are spread over all 16 waves (the first
// form walked the candidates one after the other with 192 of 256 threads busy: 113 us on the critical path of the training step for seven patches).
// Every |candidate - gt| term goes to LDS and each candidate's L1 distance is summed in a FIXED order (positions lane, lane + 64, ... then a butterfly),
// so the selection cannot change from run to run on a near-tie (LDS float atomics across waves could).
constexpr int BLUR_CHUNK = 4096;     // (candidate, position) terms held in LDS at a time
__global__ __launch_bounds__(1024) void blur_select_kernel(BlurArgs a)
{
    __shared__ float s_in[3][BLUR_MAX_PS][BLUR_MAX_PS], s_gt[3][BLUR_MAX_PS][BLUR_MAX_PS];
    __shared__ float s_term[BLUR_CHUNK];
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
        return hnr_div(acc, msk);
    };
    const int per_chunk = BLUR_CHUNK / npos;                   // candidates per LDS chunk (npos <= 768: at least 5)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (int n0 = 0; n0 <= a.N; n0 += per_chunk) {
        const int nc = min(per_chunk, a.N + 1 - n0);
        for (int it = threadIdx.x; it < nc * npos; it += blockDim.x) {
            const int n = n0 + it / npos, t = it % npos;
            const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
            s_term[it] = fabsf(blurred(n, c, y, x) - s_gt[c][y][x]);
        }
        __syncthreads();
        for (int m = wave; m < nc; m += n_waves) {             // one wave per candidate: fixed summation order
            float d = 0.f;
            for (int t = lane; t < npos; t += 64) d += s_term[m * npos + t];
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (lane == 0) s_diff[n0 + m] = d;
        }
        __syncthreads();
    }
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
            g = hnr_div(g, msk);
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


// ------------------------------------------------------------------------------------------------------------------------
// Learnable blur kernels: /root/reference/models/base_rendering_model.py:827-1020 (`learnable_blur_update_output`,
// faster_version).  The training shell turns each patch of the rendered batch and of the ground truth into a grey patch
// (:886-893), a small predictor network (owned by the aggregator, point_aggregators.py:1339-1344) maps the two grey patches
// to one ks x ks kernel per patch, the kernels are normalised / blended with the identity (:897-911, a few hundred floats:
// torch), and every patch is convolved with ITS kernel (grouped F.conv2d = cross-correlation, zero padding ks/2) with one of
// three border rules (:915-923).  Here: one block per patch for the grey patches, for the convolution and for its backward
// (gradients w.r.t. the rendered colours AND the per-patch kernels, so the predictor trains).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int BLUR_MAX_KS = 15;

struct BlurLearnArgs {
    const float *color, *gt;          // [rays,3]
    const float *kernels;             // [n_patches, ks, ks]
    const float *g_out, *g_gray;      // backward inputs
    int ks, pn, ps, n_patches, patch_major, mode;
    float *out;                       // [rays,3]
    float *gray;                      // [n_patches, 2, ps, ps]: channel 0 = ground truth, 1 = render (torch.cat order, :888 / :892)
    float *g_color, *g_kernels;       // backward outputs
};

__global__ __launch_bounds__(256) void blur_gray_kernel(BlurLearnArgs a)
{
    const int p = blockIdx.x, ps = a.ps;
    for (int t = threadIdx.x; t < ps * ps; t += blockDim.x) {
        const int y = t / ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        const float *g = a.gt + 3 * ray, *c = a.color + 3 * ray;
        a.gray[((size_t)p * 2 + 0) * ps * ps + t] = hnr_div((g[0] + g[1]) + g[2], 3.0f);
        a.gray[((size_t)p * 2 + 1) * ps * ps + t] = hnr_div((c[0] + c[1]) + c[2], 3.0f);
    }
}

__global__ __launch_bounds__(256) void blur_gray_bwd_kernel(BlurLearnArgs a)
{
    const int p = blockIdx.x, ps = a.ps;
    for (int t = threadIdx.x; t < ps * ps; t += blockDim.x) {
        const int y = t / ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        const float g = hnr_div(a.g_gray[((size_t)p * 2 + 1) * ps * ps + t], 3.0f);
        a.g_color[3 * ray + 0] = g; a.g_color[3 * ray + 1] = g; a.g_color[3 * ray + 2] = g;
    }
}

// mode 0: out = conv / (mask + 1e-10); mode 1 / 2: out = conv + (1 - mask) * in, mask = the same convolution of a ones patch
// (mode 2 takes no gradient through the mask)
template <int BWD>
__global__ __launch_bounds__(256) void blur_apply_kernel(BlurLearnArgs a)
{
    __shared__ float s_in[3][BLUR_MAX_PS][BLUR_MAX_PS], s_g[3][BLUR_MAX_PS][BLUR_MAX_PS], s_conv[3][BLUR_MAX_PS][BLUR_MAX_PS];
    __shared__ float s_msk[BLUR_MAX_PS][BLUR_MAX_PS], s_dm[BLUR_MAX_PS][BLUR_MAX_PS], s_k[BLUR_MAX_KS * BLUR_MAX_KS];
    const int p = blockIdx.x, ps = a.ps, ks = a.ks, half = ks / 2;
    const int npos = 3 * ps * ps;
    for (int t = threadIdx.x; t < ks * ks; t += blockDim.x) s_k[t] = a.kernels[(size_t)p * ks * ks + t];
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        s_in[c][y][x] = a.color[3 * ray + c];
        if (BWD) s_g[c][y][x] = a.g_out[3 * ray + c];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        float acc = 0.f, msk = 0.f;
        for (int dy = 0; dy < ks; ++dy) {
            const int yy = y + dy - half;
            if (yy < 0 || yy >= ps) continue;
            for (int dx = 0; dx < ks; ++dx) {
                const int xx = x + dx - half;
                if (xx < 0 || xx >= ps) continue;
                const float w = s_k[dy * ks + dx];
                acc = fmaf(s_in[c][yy][xx], w, acc);
                msk += w;
            }
        }
        if (!BWD) {
            const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
            a.out[3 * ray + c] = a.mode == 0 ? hnr_div(acc, msk + 1e-10f) : acc + (1.f - msk) * s_in[c][y][x];
        } else {
            s_conv[c][y][x] = acc;
            if (c == 0) s_msk[y][x] = msk;
        }
    }
    if (!BWD) return;
    __syncthreads();
    // gradient w.r.t. the convolution result (s_g, in place) and w.r.t. the mask value of every output pixel (s_dm)
    for (int t = threadIdx.x; t < ps * ps; t += blockDim.x) {
        const int y = t / ps, x = t % ps;
        const float m = s_msk[y][x];
        float dm = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float g = s_g[c][y][x];
            if (a.mode == 0) {
                const float d = m + 1e-10f;
                dm -= hnr_div(g * s_conv[c][y][x], d * d);
                s_g[c][y][x] = hnr_div(g, d);
            } else {
                dm -= g * s_in[c][y][x];
            }
        }
        s_dm[y][x] = a.mode == 2 ? 0.f : dm;
    }
    __syncthreads();
    // d in[y,x] = sum_{y0,x0} gc[y0,x0] k[y-y0+half, x-x0+half]  (+ (1 - mask[y,x]) g[y,x] for the pass-through term of modes 1 / 2;
    // s_g still holds g itself in those modes)
    for (int t = threadIdx.x; t < npos; t += blockDim.x) {
        const int c = t / (ps * ps), y = (t / ps) % ps, x = t % ps;
        float acc = 0.f;
        for (int y0 = 0; y0 < ps; ++y0) {
            const int dy = y - y0 + half;
            if (dy < 0 || dy >= ks) continue;
            for (int x0 = 0; x0 < ps; ++x0) {
                const int dx = x - x0 + half;
                if (dx < 0 || dx >= ks) continue;
                acc = fmaf(s_g[c][y0][x0], s_k[dy * ks + dx], acc);
            }
        }
        if (a.mode != 0) acc += (1.f - s_msk[y][x]) * s_g[c][y][x];
        const size_t ray = blur_ray(p, y, x, a.pn, ps, a.patch_major);
        a.g_color[3 * ray + c] = acc;
    }
    // d k[dy,dx] = sum over output pixels whose tap (dy,dx) falls inside the patch of (sum_c gc * in[tap] + d mask)
    for (int t = threadIdx.x; t < ks * ks; t += blockDim.x) {
        const int dy = t / ks, dx = t % ks;
        float acc = 0.f;
        for (int y = 0; y < ps; ++y) {
            const int yy = y + dy - half;
            if (yy < 0 || yy >= ps) continue;
            for (int x = 0; x < ps; ++x) {
                const int xx = x + dx - half;
                if (xx < 0 || xx >= ps) continue;
                float v = s_dm[y][x];
                for (int c = 0; c < 3; ++c) v = fmaf(s_g[c][y][x], s_in[c][yy][xx], v);
                acc += v;
            }
        }
        a.g_kernels[(size_t)p * ks * ks + t] = acc;
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
    blur_select_kernel<<<a.n_patches, 1024, 0, (hipStream_t)stream>>>(a);
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

static int blur_learn_fill(BlurLearnArgs &a, int kernel_size, int patch_num, int patch_size, int mode, const char *who)
{
    if (kernel_size <= 0 || (kernel_size & 1) == 0 || kernel_size > BLUR_MAX_KS || patch_num == 0 || patch_size <= 0 || patch_size > BLUR_MAX_PS ||
        mode < 0 || mode > 2) {
        set_error("%s: unsupported sizes (odd kernel size <= %d, patch size <= %d, boundary_mode 0..2)", who, BLUR_MAX_KS, BLUR_MAX_PS);
        return HNR_ERR_BADARG;
    }
    memset(&a, 0, sizeof(a));
    a.ks = kernel_size; a.ps = patch_size; a.mode = mode;
    a.patch_major = patch_num < 0; a.pn = patch_num < 0 ? 1 : patch_num; a.n_patches = patch_num < 0 ? -patch_num : patch_num * patch_num;
    return HNR_OK;
}

extern "C" int hnr_blur_gray_patches(const float *d_color, const float *d_gt, int patch_num, int patch_size, float *d_gray, void *stream)
{
    BlurLearnArgs a;
    if (blur_learn_fill(a, 1, patch_num, patch_size, 0, "hnr_blur_gray_patches") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_color || !d_gt || !d_gray) { set_error("hnr_blur_gray_patches: NULL argument"); return HNR_ERR_BADARG; }
    a.color = d_color; a.gt = d_gt; a.gray = d_gray;
    blur_gray_kernel<<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_blur_gray_patches_bwd(const float *d_g_gray, int patch_num, int patch_size, float *d_g_color, void *stream)
{
    BlurLearnArgs a;
    if (blur_learn_fill(a, 1, patch_num, patch_size, 0, "hnr_blur_gray_patches_bwd") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_g_gray || !d_g_color) { set_error("hnr_blur_gray_patches_bwd: NULL argument"); return HNR_ERR_BADARG; }
    a.g_gray = d_g_gray; a.g_color = d_g_color;
    blur_gray_bwd_kernel<<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_blur_apply(const float *d_color, const float *d_kernels, int kernel_size, int patch_num, int patch_size,
                              int boundary_mode, float *d_out, void *stream)
{
    BlurLearnArgs a;
    if (blur_learn_fill(a, kernel_size, patch_num, patch_size, boundary_mode, "hnr_blur_apply") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_color || !d_kernels || !d_out) { set_error("hnr_blur_apply: NULL argument"); return HNR_ERR_BADARG; }
    a.color = d_color; a.kernels = d_kernels; a.out = d_out;
    blur_apply_kernel<0><<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_blur_apply_bwd(const float *d_g_out, const float *d_color, const float *d_kernels, int kernel_size, int patch_num,
                                  int patch_size, int boundary_mode, float *d_g_color, float *d_g_kernels, void *stream)
{
    BlurLearnArgs a;
    if (blur_learn_fill(a, kernel_size, patch_num, patch_size, boundary_mode, "hnr_blur_apply_bwd") != HNR_OK) return HNR_ERR_BADARG;
    if (!d_g_out || !d_color || !d_kernels || !d_g_color || !d_g_kernels) { set_error("hnr_blur_apply_bwd: NULL argument"); return HNR_ERR_BADARG; }
    a.g_out = d_g_out; a.color = d_color; a.kernels = d_kernels; a.g_color = d_g_color; a.g_kernels = d_g_kernels;
    blur_apply_kernel<1><<<a.n_patches, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
