// Hole probing: which rendered pixels become new neural points.
// /root/reference/run/train_ft.py:527-549 builds, per probed frame, full-image maps from the per-ray probe outputs (opt.prob == 1,
// models/neural_points_volumetric_model.py:392-416) and combines them with torch ops + `bloat_inds` (:571-581, a 3x3 dilation of the
// missed pixels through index arithmetic).  Here: one pass marks the missed pixels, one pass decides every ray; the selected pixels come
// back as a [h, w] map of ray ids so that the caller's compaction (`nonzero`) yields the reference's row-major order.
#include "hnr_common.h"

namespace hnr {

struct ProbeArgs {
    const float *pix;                 // [R,2] (x, y) pixel of every cast ray
    const float *ray_mask;            // [R]  > 0: the ray found neighbours
    const float *gt, *color;          // [R,3] ground truth, rendered colour
    const float *far_dist, *opacity;  // [R]  ray_max_far_dist, ray_max_shading_opacity
    float bg[3];
    int R, h, w;
    float far_thresh, opacity_thresh;
    int32_t *miss;                    // [h*w] scratch: 1 where a cast ray missed although the ground truth is not background
    int32_t *sel;                     // [h*w] out: ray id + 1 of the selected pixels, 0 elsewhere
};

__global__ void probe_mark_kernel(ProbeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const int x = (int)a.pix[2 * r], y = (int)a.pix[2 * r + 1];
    if (x < 0 || x >= a.w || y < 0 || y >= a.h) return;
    const float dx = a.gt[3 * r] - a.bg[0], dy = a.gt[3 * r + 1] - a.bg[1], dz = a.gt[3 * r + 2] - a.bg[2];
    const bool miss = a.ray_mask[r] < 1.f && sqrtf(dx * dx + dy * dy + dz * dz) > 0.002f;          // :534
    if (miss) a.miss[y * a.w + x] = 1;
}

__global__ void probe_select_kernel(ProbeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const int x = (int)a.pix[2 * r], y = (int)a.pix[2 * r + 1];
    if (x < 0 || x >= a.w || y < 0 || y >= a.h) return;
    if (!(a.ray_mask[r] > 0.f) || !(a.opacity[r] > a.opacity_thresh)) return;                        // :544-546
    bool near = false;                                                                               // a missed pixel in the 3x3 neighbourhood (:536-538)
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w && a.miss[yy * a.w + xx]) near = true;
        }
    if (!near && a.far_thresh > 0.f) {                                                               // :540-543
        const float dx = a.gt[3 * r] - a.color[3 * r], dy = a.gt[3 * r + 1] - a.color[3 * r + 1], dz = a.gt[3 * r + 2] - a.color[3 * r + 2];
        near = a.far_dist[r] > a.far_thresh && sqrtf(dx * dx + dy * dy + dz * dz) < 0.1f;
    }
    if (near) a.sel[y * a.w + x] = r + 1;
}

}  // namespace hnr

using namespace hnr;

extern "C" int hnr_probe_select(const float *d_pixel_idx, const float *d_ray_mask, const float *d_gt, const float *d_raycolor, const float *bg3_host,
                                const float *d_far_dist, const float *d_opacity, int R, int h, int w, float far_thresh, float opacity_thresh,
                                int32_t *d_miss_scratch, int32_t *d_sel, void *stream)
{
    if (R < 0 || h <= 0 || w <= 0 || (int64_t)h * w > (1 << 30)) { set_error("hnr_probe_select: bad sizes"); return HNR_ERR_BADARG; }
    if (!bg3_host || !d_miss_scratch || !d_sel) { set_error("hnr_probe_select: NULL argument"); return HNR_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    HNR_HIP_CHECK(hipMemsetAsync(d_miss_scratch, 0, (size_t)h * w * sizeof(int32_t), st));
    HNR_HIP_CHECK(hipMemsetAsync(d_sel, 0, (size_t)h * w * sizeof(int32_t), st));
    if (R == 0) return HNR_OK;
    if (!d_pixel_idx || !d_ray_mask || !d_gt || !d_raycolor || !d_far_dist || !d_opacity) { set_error("hnr_probe_select: NULL argument"); return HNR_ERR_BADARG; }
    ProbeArgs a;
    a.pix = d_pixel_idx; a.ray_mask = d_ray_mask; a.gt = d_gt; a.color = d_raycolor; a.far_dist = d_far_dist; a.opacity = d_opacity;
    a.bg[0] = bg3_host[0]; a.bg[1] = bg3_host[1]; a.bg[2] = bg3_host[2];
    a.R = R; a.h = h; a.w = w; a.far_thresh = far_thresh; a.opacity_thresh = opacity_thresh; a.miss = d_miss_scratch; a.sel = d_sel;
    probe_mark_kernel<<<cdiv(R, 256), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    probe_select_kernel<<<cdiv(R, 256), 256, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
