// Sparse exchange of the point-buffer gradients of a patch-sharded training step ("sparse: all-gather (unique point id, 39-float grad) for
// touched points only ... followed by local scatter-add"; BASELINE config C5).  The reference has no counterpart: its DataParallel wrapper runs on
// gpu_ids[0] only (models/neural_points_volumetric_model.py:161-167) and autograd writes dense [N, C] gradients (neural_points.py:712-720).
//
// A rank's batch touches a few thousand of N = 2-4 M points; the forward call leaves their ids (ascending) and their number on the device
// (hnr_render_train_touched).  hnr_point_grad_pack copies the touched rows of the four dense gradients side by side into fixed-capacity records
//   rec [capacity + 2][40] floats:  row 0              header {records, this rank's valid rays, overflow flag}
//                                   rows 1..capacity   {point id (int32 bits) | emb 32 | conf 1 | dir 3 | colour 3}; unused: id -1, zeros
//                                   row capacity + 1   point 0 when the batch did not touch it (the empty neighbour slots' d conf_coefficient lands
//                                                      there through the reference's index clamp, neural_points.py:711)
// the ranks' records meet in ONE all-gather (parallel.PointGradExchange), and hnr_point_grad_apply rewrites the dense gradients as
// sum_r (n_r / n) g_r  (n_r = rank r's valid rays: the loss is a mean over the batch's valid rays): for every point some rank touched, the record of the
// LOWEST rank that holds it owns the sum and adds the ranks' rows in rank order -- one writer per element, no atomics, the same bits on every rank.
// No host read, no count exchange, no dense temporaries.
#include "hnr_common.h"

namespace hnr {

constexpr int XW = 40;                    // floats per record: id + 32 + 1 + 3 + 3

__device__ __forceinline__ float xg_value(const float *emb, const float *conf, const float *dir, const float *color, size_t p, int c /*0..38*/)
{
    if (c < 32) return emb[p * 32 + c];
    if (c == 32) return conf[p];
    if (c < 36) return dir[p * 3 + (c - 33)];
    return color[p * 3 + (c - 36)];
}

__global__ __launch_bounds__(256) void point_grad_pack_kernel(const int32_t *__restrict__ ids, const long long *__restrict__ count, int cap, const float *__restrict__ n_valid,
                                                              const float *__restrict__ emb, const float *__restrict__ conf, const float *__restrict__ dir,
                                                              const float *__restrict__ color, float *__restrict__ rec)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)(cap + 2) * XW) return;
    const int s = (int)(t / XW), c = (int)(t - (long long)s * XW);
    const long long cnt = count[0];
    const int n = cnt > cap ? cap : (int)cnt;
    float v = 0.f;
    if (s == 0) {
        v = c == 0 ? (float)n : c == 1 ? n_valid[0] : c == 2 ? (cnt > cap ? 1.f : 0.f) : 0.f;
    } else if (s <= cap) {
        const int j = s - 1;
        const bool live = j < n;
        if (c == 0) v = __int_as_float(live ? ids[j] : -1);
        else if (live) v = xg_value(emb, conf, dir, color, (size_t)ids[j], c - 1);
    } else {
        const bool zero_in = n > 0 && ids[0] == 0;      // ascending ids: point 0 is touched iff it comes first
        if (c == 0) v = __int_as_float(zero_in ? -1 : 0);
        else if (!zero_in) v = xg_value(emb, conf, dir, color, 0, c - 1);
    }
    rec[t] = v;
}

// slot (1..cap + 1) of point p in rank r's records, or -1
__device__ __forceinline__ int xg_find(const float *__restrict__ rec_r, int cap, int p)
{
    if (p == 0 && __float_as_int(rec_r[(size_t)(cap + 1) * XW]) == 0) return cap + 1;
    int lo = 1, hi = (int)rec_r[0];                      // live slots 1..n, ascending ids
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int q = __float_as_int(rec_r[(size_t)mid * XW]);
        if (q == p) return mid;
        if (q < p) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

__global__ __launch_bounds__(256) void point_grad_apply_kernel(const float *__restrict__ all, int world, int cap, float *__restrict__ emb, float *__restrict__ conf,
                                                               float *__restrict__ dir, float *__restrict__ color, int n_points, float *__restrict__ out2)
{
    const int lane = threadIdx.x & 63;
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t stride = (size_t)(cap + 2) * XW;
    float tot = 0.f, over = 0.f;
    for (int r = 0; r < world; ++r) { tot += all[r * stride + 1]; over = fmaxf(over, all[r * stride + 2]); }
    tot = fmaxf(tot, 1.0f);
    if (w == 0 && lane == 0 && out2) { out2[0] = tot; out2[1] = over; }
    if (w >= (long long)world * (cap + 1)) return;
    const int r = (int)(w / (cap + 1)), s = 1 + (int)(w - (long long)r * (cap + 1));
    const float *mine = all + r * stride;
    const int p = __float_as_int(mine[(size_t)s * XW]);
    if (p < 0 || p >= n_points) return;
    for (int q = 0; q < r; ++q)
        if (xg_find(all + q * stride, cap, p) >= 0) return;                 // a lower rank owns this point's sum
    if (lane >= XW - 1) return;
    float sum = 0.f;
    for (int q = r; q < world; ++q) {
        const int pos = q == r ? s : xg_find(all + q * stride, cap, p);
        if (pos < 0) continue;
        const float scale = hnr_div(all[q * stride + 1], tot);
        sum = __fadd_rn(sum, __fmul_rn(all[q * stride + (size_t)pos * XW + 1 + lane], scale));
    }
    const size_t pp = (size_t)p;
    if (lane < 32) emb[pp * 32 + lane] = sum;
    else if (lane == 32) conf[pp] = sum;
    else if (lane < 36) dir[pp * 3 + (lane - 33)] = sum;
    else color[pp * 3 + (lane - 36)] = sum;
}

}  // namespace hnr

using namespace hnr;

extern "C" int hnr_point_grad_pack(const int32_t *d_ids, const int64_t *d_count, int capacity, const float *d_n_valid, const float *d_g_emb, const float *d_g_conf,
                                   const float *d_g_dir, const float *d_g_color, float *d_rec, void *stream)
{
    if (!d_ids || !d_count || !d_n_valid || !d_g_emb || !d_g_conf || !d_g_dir || !d_g_color || !d_rec || capacity <= 0 || capacity > (1 << 24)) {
        set_error("hnr_point_grad_pack: NULL argument or capacity out of range (1 .. 2^24)"); return HNR_ERR_BADARG;
    }
    const long long n = (long long)(capacity + 2) * XW;
    point_grad_pack_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_ids, reinterpret_cast<const long long *>(d_count), capacity, d_n_valid, d_g_emb, d_g_conf,
                                                                         d_g_dir, d_g_color, d_rec);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_point_grad_apply(const float *d_all_rec, int n_ranks, int capacity, float *d_g_emb, float *d_g_conf, float *d_g_dir, float *d_g_color, int n_points,
                                    float *d_out2, void *stream)
{
    if (!d_all_rec || !d_g_emb || !d_g_conf || !d_g_dir || !d_g_color || n_ranks <= 0 || n_ranks > 1024 || capacity <= 0 || capacity > (1 << 24) || n_points <= 0) {
        set_error("hnr_point_grad_apply: NULL argument or sizes out of range"); return HNR_ERR_BADARG;
    }
    const long long waves = (long long)n_ranks * (capacity + 1);
    point_grad_apply_kernel<<<cdiv(waves * 64, 256), 256, 0, (hipStream_t)stream>>>(d_all_rec, n_ranks, capacity, d_g_emb, d_g_conf, d_g_dir, d_g_color, n_points, d_out2);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
