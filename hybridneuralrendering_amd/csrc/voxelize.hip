// Voxel down-sampling of an initial point cloud (SURVEY 8f "next" row 4): one representative point per occupied voxel.
//
// Replaces models/mvs/mvs_utils.py:537-563 (`construct_vox_points_closest`, called at run/train_ft.py:164 and :725 on the MVS /
// depth points before they become neural points), which needs torch_scatter: cell = floor((xyz - space_min) / vox_size) (fp32
// subtract, fp32 divide), torch.unique(dim=0) over the int32 cells (lexicographic order), scatter_mean = per-voxel centroid,
// scatter_min of |xyz - centroid| = the point closest to its voxel's centroid.
//
// Here: 63-bit cell keys -> stable radix sort of (key, point id) (rocprim) -> head flags + scan = voxel ids in the reference's
// lexicographic order -> one thread per voxel walks its (short) run twice: centroid as a sequential fp32 sum in point-id order
// (torch_scatter's CPU order; its CUDA path uses atomics and is not reproducible), then the first point with the smallest
// residual.  Deterministic; no atomics.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "hnr_common.h"

namespace hnr {

constexpr int VOX_BITS = 21;        // cells per axis < 2^21

__global__ void vox_keys_kernel(const float *__restrict__ xyz, int n, float mx, float my, float mz, float sz, unsigned long long *__restrict__ keys,
                                int *__restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qx = floorf(hnr_div(__fsub_rn(xyz[3 * i + 0], mx), sz));
    const float qy = floorf(hnr_div(__fsub_rn(xyz[3 * i + 1], my), sz));
    const float qz = floorf(hnr_div(__fsub_rn(xyz[3 * i + 2], mz), sz));
    const float lim = (float)(1 << VOX_BITS);
    if (!(qx >= 0.f && qx < lim && qy >= 0.f && qy < lim && qz >= 0.f && qz < lim)) { atomicOr(bad, 1); keys[i] = ~0ull >> 1; return; }
    keys[i] = ((unsigned long long)(unsigned)qx << (2 * VOX_BITS)) | ((unsigned long long)(unsigned)qy << VOX_BITS) | (unsigned long long)(unsigned)qz;
}

__global__ void vox_heads_kernel(const unsigned long long *__restrict__ keys_sorted, int n, int *__restrict__ head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys_sorted[i] != keys_sorted[i - 1]) ? 1 : 0;
}

// vid1[i] = inclusive scan of the head flags = 1 + voxel of sorted entry i
__global__ void vox_reduce_kernel(const float *__restrict__ xyz, const unsigned long long *__restrict__ keys_sorted, const int *__restrict__ perm,
                                  const int *__restrict__ head, const int *__restrict__ vid1, int n, float *__restrict__ centroid,
                                  int *__restrict__ grid_idx, int *__restrict__ min_idx, int *__restrict__ inverse, long long *__restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (inverse) inverse[perm[i]] = vid1[i] - 1;
    if (i == n - 1) count[0] = (long long)vid1[i];
    if (!head[i]) return;
    const int v = vid1[i] - 1;
    const unsigned long long key = keys_sorted[i];
    int e = i;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (; e < n && keys_sorted[e] == key; ++e) {
        const int p = perm[e];
        sx += xyz[3 * p + 0]; sy += xyz[3 * p + 1]; sz += xyz[3 * p + 2];
    }
    const float cnt = (float)(e - i);
    const float cx = hnr_div(sx, cnt), cy = hnr_div(sy, cnt), cz = hnr_div(sz, cnt);
    float best = 0.f;
    int arg = -1;
    for (int k = i; k < e; ++k) {
        const int p = perm[k];
        const float dx = __fsub_rn(xyz[3 * p + 0], cx), dy = __fsub_rn(xyz[3 * p + 1], cy), dz = __fsub_rn(xyz[3 * p + 2], cz);
        const float r = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        if (arg < 0 || r < best) { best = r; arg = p; }          // first minimum in point-id order
    }
    centroid[3 * v + 0] = cx; centroid[3 * v + 1] = cy; centroid[3 * v + 2] = cz;
    grid_idx[3 * v + 0] = (int)(key >> (2 * VOX_BITS));
    grid_idx[3 * v + 1] = (int)((key >> VOX_BITS) & ((1u << VOX_BITS) - 1));
    grid_idx[3 * v + 2] = (int)(key & ((1u << VOX_BITS) - 1));
    min_idx[v] = arg;
}

}  // namespace hnr

using namespace hnr;

static size_t vox_align(size_t v) { return (v + 255) & ~(size_t)255; }

static int vox_layout(int64_t n, size_t *sort_bytes, size_t *scan_bytes, size_t *total)
{
    size_t sb = 0, cb = 0;
    rocprim::counting_iterator<int> iota(0);
    if (rocprim::radix_sort_pairs(nullptr, sb, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, iota, (int *)nullptr, (size_t)n, 0,
                                  3 * VOX_BITS, (hipStream_t) nullptr) != hipSuccess)
        return -1;
    if (rocprim::inclusive_scan(nullptr, cb, (const int *)nullptr, (int *)nullptr, (size_t)n, rocprim::plus<int>(), (hipStream_t) nullptr) != hipSuccess)
        return -1;
    if (sort_bytes) *sort_bytes = sb;
    if (scan_bytes) *scan_bytes = cb;
    // keys, keys_sorted (u64), perm, head, vid (i32), bad flag, rocprim temp (max of both)
    *total = 2 * vox_align(8 * (size_t)n) + 3 * vox_align(4 * (size_t)n) + 256 + vox_align(sb > cb ? sb : cb);
    return 0;
}

extern "C" int64_t hnr_voxel_downsample_scratch_bytes(int64_t n)
{
    if (n <= 0) return 256;
    size_t total = 0;
    if (vox_layout(n, nullptr, nullptr, &total) != 0) return -1;
    return (int64_t)total;
}

extern "C" int hnr_voxel_downsample(const float *d_xyz, int n, const float *space_min, float vox_size, float *d_centroid, int32_t *d_grid_idx,
                                    int32_t *d_min_idx, int32_t *d_inverse, int64_t *d_count, void *d_scratch, int64_t scratch_bytes, void *stream)
{
    if (n < 0 || !space_min || !(vox_size > 0.f)) { set_error("hnr_voxel_downsample: bad argument (n >= 0, vox_size > 0)"); return HNR_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { if (d_count) HNR_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int64_t), st)); return HNR_OK; }
    size_t sb = 0, cb = 0, total = 0;
    if (!d_xyz || !d_centroid || !d_grid_idx || !d_min_idx || !d_count || !d_scratch || vox_layout(n, &sb, &cb, &total) != 0 || (size_t)scratch_bytes < total) {
        set_error("hnr_voxel_downsample: NULL argument or scratch smaller than hnr_voxel_downsample_scratch_bytes(n)"); return HNR_ERR_BADARG;
    }
    char *p = (char *)d_scratch;
    unsigned long long *keys = (unsigned long long *)p; p += vox_align(8 * (size_t)n);
    unsigned long long *keys_sorted = (unsigned long long *)p; p += vox_align(8 * (size_t)n);
    int *perm = (int *)p; p += vox_align(4 * (size_t)n);
    int *head = (int *)p; p += vox_align(4 * (size_t)n);
    int *vid = (int *)p; p += vox_align(4 * (size_t)n);
    int *bad = (int *)p; p += 256;
    void *tmp = p;
    HNR_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
    vox_keys_kernel<<<cdiv(n, 256), 256, 0, st>>>(d_xyz, n, space_min[0], space_min[1], space_min[2], vox_size, keys, bad);
    HNR_LAUNCH_CHECK();
    rocprim::counting_iterator<int> iota(0);
    size_t sz = sb;
    HNR_HIP_CHECK(rocprim::radix_sort_pairs(tmp, sz, keys, keys_sorted, iota, perm, (size_t)n, 0, 3 * VOX_BITS, st));
    vox_heads_kernel<<<cdiv(n, 256), 256, 0, st>>>(keys_sorted, n, head);
    HNR_LAUNCH_CHECK();
    sz = cb;
    HNR_HIP_CHECK(rocprim::inclusive_scan(tmp, sz, head, vid, (size_t)n, rocprim::plus<int>(), st));
    vox_reduce_kernel<<<cdiv(n, 256), 256, 0, st>>>(d_xyz, keys_sorted, perm, head, vid, n, d_centroid, d_grid_idx, d_min_idx, d_inverse, (long long *)d_count);
    HNR_LAUNCH_CHECK();
    int h_bad = 0;
    HNR_HIP_CHECK(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HNR_HIP_CHECK(hipStreamSynchronize(st));
    if (h_bad) { set_error("hnr_voxel_downsample: a point lies outside [space_min, space_min + 2^21 * vox_size) or is not finite"); return HNR_ERR_BADARG; }
    return HNR_OK;
}
