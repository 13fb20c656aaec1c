// The loss terms of the shipped training configurations on the device, value and gradients in two small kernels
// (SURVEY 8f "next" row 2, second half).
//
// Replaces the part of BaseRenderingModel.compute_losses (models/base_rendering_model.py:1060-1245) that the launch scripts
// enable (dev_scripts/w_scannet_etf/scene241.sh:146-151):
//   * colour item `ray_masked_coarse_raycolor` (:1113-1118): two torch.masked_select copies of the batch + nn.MSELoss over the
//     rays with ray_mask > 0 (0 when no ray is valid, :1144), `loss_total += loss * weight + 1e-6` (:1198), then
//     `loss_total *= frame_weight` when the dataset provides one (:1204-1205);
//   * zero-one regulariser on `conf_coefficient` (:1228-1240): mean(log(v) + log(1 - v)), v = clamp(x, eps, 1 - eps), added
//     with its weight after the frame-weight scaling.
// torch autograd then walks those graphs backwards; here the gradients come out of the second kernel directly.
// Reductions are deterministic: per-block partial sums in double, added in block order by one thread.
#include "hnr_common.h"

namespace hnr {

constexpr int LOSS_BLOCKS = 256;

struct LossArgs {
    const float *color, *gt;          // [R,3]
    const int8_t *ray_mask;           // [R]
    const float *conf;                // [n_conf]
    int R; long long n_conf;
    int conf_per_ray;                 // > 0: conf is [R, conf_per_ray] and only the rows of rays with ray_mask > 0 count (n_conf = R * conf_per_ray)
    float eps, w_color, w_zero_one, frame_weight;
    const float *d_frame_weight;      // optional: the item's scalar on the device (a captured step replays with another value); wins over frame_weight
    double *partial;                  // [LOSS_BLOCKS][3]: squared error, valid rays, zero-one sum
    float *out;                       // {total, colour, zero_one, n_valid}
    float *g_color, *g_conf;          // may be NULL (value only)
};

__global__ __launch_bounds__(256) void loss_partial_kernel(LossArgs a)
{
    __shared__ double s[3][4];
    double se = 0.0, nv = 0.0, zo = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long r = t0; r < a.R; r += stride) {
        if (a.ray_mask[r] > 0) {
            for (int c = 0; c < 3; ++c) { const float d = a.color[3 * r + c] - a.gt[3 * r + c]; se += (double)(d * d); }
            nv += 1.0;
        }
    }
    for (long long i = t0; i < a.n_conf; i += stride) {
        if (a.conf_per_ray > 0 && !(a.ray_mask[i / a.conf_per_ray] > 0)) continue;
        const float v = fminf(fmaxf(a.conf[i], a.eps), 1.f - a.eps);
        zo += (double)(logf(v) + logf(1.f - v));
    }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); nv += __shfl_xor(nv, o); zo += __shfl_xor(zo, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[0][wave] = se; s[1][wave] = nv; s[2][wave] = zo; }
    __syncthreads();
    if (threadIdx.x < 3) a.partial[blockIdx.x * 3 + threadIdx.x] = (s[threadIdx.x][0] + s[threadIdx.x][1]) + (s[threadIdx.x][2] + s[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void loss_finish_kernel(LossArgs a)
{
    __shared__ float s_scale[2];
    if (threadIdx.x == 0) {
        double se = 0.0, nv = 0.0, zo = 0.0;
        for (int b = 0; b < LOSS_BLOCKS; ++b) { se += a.partial[3 * b]; nv += a.partial[3 * b + 1]; zo += a.partial[3 * b + 2]; }
        const float fw = a.d_frame_weight ? a.d_frame_weight[0] : a.frame_weight;
        const float lc = nv > 0.0 ? (float)hnr_div64(se, 3.0 * nv) : 0.f;
        const double n_conf = a.conf_per_ray > 0 ? nv * (double)a.conf_per_ray : (double)a.n_conf;
        const float lz = n_conf > 0.0 ? (float)hnr_div64(zo, n_conf) : 0.f;
        s_scale[0] = nv > 0.0 ? (float)hnr_div64(2.0, 3.0 * nv) * a.w_color * fw : 0.f;
        s_scale[1] = n_conf > 0.0 ? hnr_div(a.w_zero_one, (float)n_conf) : 0.f;
        if (blockIdx.x == 0) {
            a.out[0] = (lc * a.w_color + 1e-6f) * fw + lz * a.w_zero_one;
            a.out[1] = lc; a.out[2] = lz; a.out[3] = (float)nv;
        }
    }
    __syncthreads();
    if (!a.g_color) return;
    const float sc = s_scale[0], sz = s_scale[1];
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long r = t0; r < a.R; r += stride) {
        const bool on = a.ray_mask[r] > 0;
        for (int c = 0; c < 3; ++c) a.g_color[3 * r + c] = on ? (a.color[3 * r + c] - a.gt[3 * r + c]) * sc : 0.f;
    }
    for (long long i = t0; i < a.n_conf; i += stride) {
        const float x = a.conf[i];
        const bool on = a.conf_per_ray <= 0 || a.ray_mask[i / a.conf_per_ray] > 0;
        // torch.clamp passes the gradient where eps <= x <= 1 - eps
        a.g_conf[i] = (on && x >= a.eps && x <= 1.f - a.eps) ? (hnr_div(1.f, x) - hnr_div(1.f, 1.f - x)) * sz : 0.f;
    }
}

}  // namespace hnr

using namespace hnr;

extern "C" int64_t hnr_shipped_loss_scratch_bytes(void) { return (int64_t)(LOSS_BLOCKS * 3 * sizeof(double)); }

static int shipped_loss_impl(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int64_t n_conf, int conf_per_ray,
                             float zero_epsilon, float w_color, float w_zero_one, float frame_weight, float *d_out4, float *d_g_color,
                             float *d_g_conf, void *d_scratch, void *stream, const float *d_frame_weight = nullptr)
{
    if (R < 0 || n_conf < 0 || conf_per_ray < 0 || !(zero_epsilon >= 0.f && zero_epsilon < 0.5f)) { set_error("hnr_shipped_loss: bad sizes or zero_epsilon"); return HNR_ERR_BADARG; }
    if ((R > 0 && (!d_color || !d_gt || !d_ray_mask)) || (n_conf > 0 && !d_conf) || !d_out4 || !d_scratch || (!d_g_color != !d_g_conf && n_conf > 0 && R > 0)) {
        set_error("hnr_shipped_loss: NULL argument (the two gradient outputs go together)"); return HNR_ERR_BADARG;
    }
    LossArgs a;
    a.color = d_color; a.gt = d_gt; a.ray_mask = d_ray_mask; a.conf = d_conf; a.R = R; a.n_conf = n_conf; a.conf_per_ray = conf_per_ray;
    a.eps = zero_epsilon; a.w_color = w_color; a.w_zero_one = w_zero_one; a.frame_weight = frame_weight; a.d_frame_weight = d_frame_weight;
    a.partial = (double *)d_scratch; a.out = d_out4; a.g_color = d_g_color; a.g_conf = d_g_conf;
    loss_partial_kernel<<<LOSS_BLOCKS, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    loss_finish_kernel<<<LOSS_BLOCKS, 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_shipped_loss(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int64_t n_conf,
                                float zero_epsilon, float w_color, float w_zero_one, float frame_weight, float *d_out4, float *d_g_color,
                                float *d_g_conf, void *d_scratch, void *stream)
{
    return shipped_loss_impl(d_color, d_gt, d_ray_mask, R, d_conf, n_conf, 0, zero_epsilon, w_color, w_zero_one, frame_weight, d_out4, d_g_color, d_g_conf, d_scratch, stream);
}

extern "C" int hnr_shipped_loss_rows(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int conf_per_ray,
                                     float zero_epsilon, float w_color, float w_zero_one, float frame_weight, float *d_out4, float *d_g_color,
                                     float *d_g_conf, void *d_scratch, void *stream)
{
    return shipped_loss_impl(d_color, d_gt, d_ray_mask, R, d_conf, (int64_t)R * conf_per_ray, conf_per_ray, zero_epsilon, w_color, w_zero_one, frame_weight, d_out4,
                             d_g_color, d_g_conf, d_scratch, stream);
}

// ... with the item's frame weight read from the device (d_frame_weight[0]): a training step captured in a hipGraph is replayed for items with different weights.
extern "C" int hnr_shipped_loss_rows_fw(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int conf_per_ray,
                                        float zero_epsilon, float w_color, float w_zero_one, const float *d_frame_weight, float *d_out4, float *d_g_color,
                                        float *d_g_conf, void *d_scratch, void *stream)
{
    if (!d_frame_weight) { set_error("hnr_shipped_loss_rows_fw: NULL d_frame_weight"); return HNR_ERR_BADARG; }
    return shipped_loss_impl(d_color, d_gt, d_ray_mask, R, d_conf, (int64_t)R * conf_per_ray, conf_per_ray, zero_epsilon, w_color, w_zero_one, 1.0f, d_out4,
                             d_g_color, d_g_conf, d_scratch, stream, d_frame_weight);
}
