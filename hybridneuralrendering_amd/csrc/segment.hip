// Sum-by-key of gradient rows without per-element atomics: rows are ordered by key with a stable radix sort (rocprim), then
// every wave walks a short run of the sorted order with a running sum and writes one result per key.
//
// Used by the backward pass for the two "many rows -> few destinations" reductions that torch autograd performs with
// index_add / scatter kernels in the reference:
//   * d(per-point table) = sum of block1's first-layer gradient rows over the rows that reference the point
//     (gather of neural_points.py:709-720 transposed; 272 k rows of 256 floats -> ~10^5 points in a training batch);
//   * d(feature map pixel) = sum of the image-feature gradient rows over the samples that reproject to the pixel
//     (point_aggregators.py:1077-1089, :1193 transposed).
// A direct atomicAdd scatter costs 1.06 ms / 1.15 ms per step for these two (rocprofv3, profiles/r01_train_kernel_stats.csv).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "hnr_common.h"
#include "train_internal.h"

namespace hnr {

constexpr int SEG_CHUNK = 16;       // sorted entries per wave

// dst[key * dst_stride + c] += sum over the rows with that key of (A[row, c] + B[row, c]),  c < n_cols (multiple of 4, <= 256).
// Keys < 0 are skipped.  A key's rows may straddle two waves' runs, so the (few) results are added with atomics.
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                               const int32_t *__restrict__ keys_sorted, const int32_t *__restrict__ perm,
                                                               int64_t M, int n_cols, float *__restrict__ dst, int64_t dst_stride)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t e0 = wave * SEG_CHUNK;
    if (e0 >= M) return;
    const int64_t e1 = e0 + SEG_CHUNK < M ? e0 + SEG_CHUNK : M;
    const bool active = 4 * lane < n_cols;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur = keys_sorted[e0];
    auto flush = [&](int key) {
        if (key >= 0 && active) {
            float *d = dst + (size_t)key * dst_stride + 4 * lane;
            atomicAdd(d, acc.x); atomicAdd(d + 1, acc.y); atomicAdd(d + 2, acc.z); atomicAdd(d + 3, acc.w);
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    for (int64_t e = e0; e < e1; ++e) {
        const int key = keys_sorted[e];
        if (key != cur) { flush(cur); cur = key; }
        if (key >= 0 && active) {
            const int row = perm[e];
            float4 v = reinterpret_cast<const float4 *>(A + (size_t)row * lda)[lane];
            if (B) {
                const float4 w = reinterpret_cast<const float4 *>(B + (size_t)row * ldb)[lane];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    flush(cur);
}

// Deterministic variant for DENSE keys 0 .. n_keys-1 (the compact indices of the points a batch touches): one wave per key finds its
// run in the sorted order by binary search and adds the rows in that (stable = row) order -- no atomics, bit-identical run to run.
// dst row = dst_index ? dst_index[key] : key;  accumulate != 0: dst += sum (dst rows are unique per key, so a plain read-modify-write).
__global__ __launch_bounds__(256) void segment_sum_rows_det_kernel(const float *__restrict__ A, int lda, const int32_t *__restrict__ keys_sorted,
                                                                   const int32_t *__restrict__ perm, int64_t M, int n_cols, int n_keys,
                                                                   const int32_t *__restrict__ dst_index, float *__restrict__ dst,
                                                                   int64_t dst_stride, int accumulate, const long long *__restrict__ d_nkeys = nullptr,
                                                                   const int32_t *__restrict__ seg_start = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int key = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (d_nkeys && *d_nkeys < n_keys) n_keys = (int)*d_nkeys;
    if (key >= n_keys) return;
    int64_t lo = 0, hi = M;                               // first entry with keys_sorted >= key
    if (seg_start) lo = seg_start[key];                   // (precomputed once per sort: 20 dependent loads per wave otherwise)
    else while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys_sorted[mid] < key) lo = mid + 1; else hi = mid; }
    if (4 * lane >= n_cols) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t e = lo; e < M && keys_sorted[e] == key; ++e) {
        const float4 v = reinterpret_cast<const float4 *>(A + (size_t)perm[e] * lda)[lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 *d = reinterpret_cast<float4 *>(dst + (size_t)(dst_index ? dst_index[key] : key) * dst_stride) + lane;
    if (accumulate) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    *d = acc;
}

// Point-major row list (csrc/backward.hip unique_points_dc: start / count per compact point, the rows of a segment in arbitrary order): one
// WORKGROUP per key.  The segment's row indices are staged in LDS and sorted by rank counting (row indices are distinct: an element's rank is
// the number of smaller ones -- cnt broadcast LDS reads per element, no shuffles); wave w then adds the w-th quarter of the sorted rows in
// ascending order with eight row loads in flight, and the four partial sums are added in wave order: a fixed order, bit-identical run to run,
// without the 21 launches of a device-wide stable sort.  (One wave per key, four loads in flight, segments of more than 64 rows by repeated
// minimum search, took 229 us on the C3 batch: 8 145 touched points with 27 rows at the median and up to 171 -- the chip waited for the few
// waves with the long segments.)  Segments of more than SEG_LDS_ROWS rows: wave 0 alone, by repeated minimum search.
constexpr int SEG_LDS_ROWS = 2048;
constexpr int SEG_WAVE_ROWS = 16;       // segments up to this many rows: one wave, no barrier
__global__ __launch_bounds__(256) void segment_sum_rows_csr_kernel(const float *__restrict__ A, int lda, const int32_t *__restrict__ row_list,
                                                                   const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_count, int n_cols,
                                                                   int n_keys, const long long *__restrict__ d_nkeys, float *__restrict__ dst, int64_t dst_stride,
                                                                   const float *__restrict__ A2, int lda2, int n_cols2, float *__restrict__ dst2, int64_t dst_stride2,
                                                                   unsigned *__restrict__ absmax)
{
    // (A2: an optional second, narrow matrix summed over the same segments in the same pass by the first n_cols2 / 4 lanes; absmax: optional,
    // max |dst| of the rows written -- absmax_publish, hnr_common.h)
    float mx = 0.f;
    __shared__ int s_ids[SEG_LDS_ROWS], s_sorted[SEG_LDS_ROWS];
    __shared__ float4 s_part[4][64], s_part2[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (d_nkeys && *d_nkeys < n_keys) n_keys = (int)*d_nkeys;
    const bool on = 4 * lane < n_cols, on2 = A2 && 4 * lane < n_cols2;
    const float *base = A + 4 * lane, *base2 = A2 + 4 * lane;
    auto add = [&](float4 &a, const float4 &v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    // Keys are taken four at a time, one per wave.  SHORT segments (<= SEG_WAVE_ROWS rows: what a batch of scattered patches produces -- config 5 touches
    // 66 k points with ~5 rows each, where the workgroup-per-key form below spent its time in three barriers and an LDS sort per key: 458 us) are summed
    // by their wave alone: ids in lanes, rank by comparison, rows added one after the other in ascending row order (a fixed order; for such a segment the
    // sum is a plain left-to-right one instead of four quarter sums).  The longer ones of the four follow, each by the whole workgroup.
    __shared__ int s_long[4];
    for (int g = blockIdx.x * 4; g < n_keys; g += gridDim.x * 4) {
        {
            const int key = g + wave;
            const int cnt = key < n_keys ? seg_count[key] : 0;
            if (cnt > 0 && cnt <= SEG_WAVE_ROWS) {
                const int lo = seg_start[key];
                const int mine = lane < cnt ? row_list[lo + lane] : 0x7fffffff;
                int rank = 0;
                for (int j = 0; j < cnt; ++j) rank += __shfl(mine, j) < mine ? 1 : 0;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e0 = 0; e0 < cnt; e0 += 4) {                      // four rows' loads in flight
                    int r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned long long b = __ballot(lane < cnt && rank == e0 + i);
                        r[i] = b ? __shfl(mine, __ffsll((long long)b) - 1) : -1;
                    }
                    float4 v[4], v2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = (on && r[i] >= 0) ? *reinterpret_cast<const float4 *>(base + (size_t)r[i] * lda) : make_float4(0.f, 0.f, 0.f, 0.f);
                        v2[i] = (on2 && r[i] >= 0) ? *reinterpret_cast<const float4 *>(base2 + (size_t)r[i] * lda2) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (r[i] >= 0) { add(acc, v[i]); add(acc2, v2[i]); }
                }
                if (on) {
                    reinterpret_cast<float4 *>(dst + (size_t)key * dst_stride)[lane] = acc;
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
                }
                if (on2) reinterpret_cast<float4 *>(dst2 + (size_t)key * dst_stride2)[lane] = acc2;
            }
            if (lane == 0) s_long[wave] = (cnt > SEG_WAVE_ROWS || (cnt == 0 && key < n_keys)) ? key : -1;
        }
        __syncthreads();
      for (int kk = 0; kk < 4; ++kk) {
        const int key = s_long[kk];
        if (key < 0) continue;                                             // (uniform over the workgroup: read from LDS)
        const int lo = seg_start[key], cnt = seg_count[key];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cnt <= SEG_LDS_ROWS) {
            for (int i = tid; i < cnt; i += 256) s_ids[i] = row_list[lo + i];
            __syncthreads();
            for (int i = tid; i < cnt; i += 256) {
                const int mine = s_ids[i];
                int rank = 0;
                for (int j = 0; j < cnt; ++j) rank += s_ids[j] < mine ? 1 : 0;
                s_sorted[rank] = mine;
            }
            __syncthreads();
            const int q = (cnt + 3) >> 2, e0 = min(wave * q, cnt), e1 = min(e0 + q, cnt);
            int e = e0;
            for (; e + 8 <= e1; e += 8) {
                int r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = s_sorted[e + i];
                if (on) {
                    float4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4 *>(base + (size_t)r[i] * lda);
#pragma unroll
                    for (int i = 0; i < 8; ++i) add(acc, v[i]);
                }
                if (on2) {
                    float4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4 *>(base2 + (size_t)r[i] * lda2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) add(acc2, v[i]);
                }
            }
            for (; e < e1; ++e) {
                const int row = s_sorted[e];
                if (on) add(acc, *reinterpret_cast<const float4 *>(base + (size_t)row * lda));
                if (on2) add(acc2, *reinterpret_cast<const float4 *>(base2 + (size_t)row * lda2));
            }
        } else if (wave == 0) {
            int last = -1;
            for (int e = 0; e < cnt; ++e) {
                int m = 0x7fffffff;
                for (int i = lane; i < cnt; i += 64) { const int r = row_list[lo + i]; m = (r > last && r < m) ? r : m; }
                for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
                last = m;
                if (on) add(acc, *reinterpret_cast<const float4 *>(base + (size_t)m * lda));
                if (on2) add(acc2, *reinterpret_cast<const float4 *>(base2 + (size_t)m * lda2));
            }
        }
        s_part[wave][lane] = acc;
        s_part2[wave][lane] = acc2;
        __syncthreads();
        if (wave == 0) {
            float4 t = s_part[0][lane];
            add(t, s_part[1][lane]); add(t, s_part[2][lane]); add(t, s_part[3][lane]);
            if (on) {
                reinterpret_cast<float4 *>(dst + (size_t)key * dst_stride)[lane] = t;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
            }
            float4 t2 = s_part2[0][lane];
            add(t2, s_part2[1][lane]); add(t2, s_part2[2][lane]); add(t2, s_part2[3][lane]);
            if (on2) reinterpret_cast<float4 *>(dst2 + (size_t)key * dst_stride2)[lane] = t2;
        }
        __syncthreads();                                 // the LDS lists and partials are reused by the block's next key
      }
        __syncthreads();                                 // (s_long is rewritten by the next group)
    }
    if (absmax) {                                         // (every wave may hold a maximum now)
        __shared__ float s_mx[4];
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) s_mx[wave] = mx;
        __syncthreads();
        if (tid == 0) absmax_publish(absmax, fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3])));
    }
}

}  // namespace hnr

using namespace hnr;

extern "C" int hnr_segment_sum_rows_det(const float *d_A, int lda, const int32_t *d_keys_sorted, const int32_t *d_perm, int64_t M, int n_cols,
                                        int n_keys, const int32_t *d_dst_index, float *d_dst, int64_t dst_stride, int accumulate, void *stream)
{
    if (M < 0 || n_keys < 0 || n_cols <= 0 || n_cols > 256 || (n_cols & 3) || lda < n_cols || (lda & 3) || dst_stride < n_cols || (dst_stride & 3)) {
        set_error("hnr_segment_sum_rows_det: bad sizes (n_cols a multiple of 4, <= 256; strides multiples of 4)"); return HNR_ERR_BADARG;
    }
    if (n_keys == 0) return HNR_OK;
    if (!d_A || !d_keys_sorted || !d_perm || !d_dst || ((uintptr_t)d_A & 15) || ((uintptr_t)d_dst & 15)) { set_error("hnr_segment_sum_rows_det: NULL / unaligned argument"); return HNR_ERR_BADARG; }
    segment_sum_rows_det_kernel<<<cdiv((int64_t)n_keys * 64, 256), 256, 0, (hipStream_t)stream>>>(d_A, lda, d_keys_sorted, d_perm, M, n_cols, n_keys, d_dst_index,
                                                                                                 d_dst, dst_stride, accumulate);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_segment_sum_rows_csr(const float *d_A, int lda, const int32_t *d_row_list, const int32_t *d_seg_start, const int32_t *d_seg_count,
                                        int n_cols, int n_keys, float *d_dst, int64_t dst_stride, const float *d_A2, int lda2, int n_cols2,
                                        float *d_dst2, int64_t dst_stride2, uint32_t *d_absmax, void *stream)
{
    if (n_keys < 0 || n_cols <= 0 || n_cols > 256 || (n_cols & 3) || lda < n_cols || (lda & 3) || dst_stride < n_cols || (dst_stride & 3) ||
        (d_A2 && (n_cols2 <= 0 || n_cols2 > 256 || (n_cols2 & 3) || lda2 < n_cols2 || (lda2 & 3) || dst_stride2 < n_cols2 || (dst_stride2 & 3) || !d_dst2))) {
        set_error("hnr_segment_sum_rows_csr: bad sizes (column counts multiples of 4, <= 256; strides multiples of 4)"); return HNR_ERR_BADARG;
    }
    if (n_keys == 0) return HNR_OK;
    if (!d_A || !d_row_list || !d_seg_start || !d_seg_count || !d_dst || ((uintptr_t)d_A & 15) || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_A2 & 15) || ((uintptr_t)d_dst2 & 15)) {
        set_error("hnr_segment_sum_rows_csr: NULL / unaligned argument"); return HNR_ERR_BADARG;
    }
    return hnr::segment_sum_rows_csr_dc(d_A, lda, d_row_list, d_seg_start, d_seg_count, n_cols, n_keys, nullptr, d_dst, dst_stride, d_A2, lda2, n_cols2, d_dst2,
                                        dst_stride2, d_absmax, (hipStream_t)stream);
}

extern "C" int64_t hnr_sort_rows_scratch_bytes(int64_t M)
{
    if (M <= 0) return 16;
    size_t sz = 0;
    rocprim::counting_iterator<int32_t> iota(0);
    if (rocprim::radix_sort_pairs(nullptr, sz, (const int32_t *)nullptr, (int32_t *)nullptr, iota, (int32_t *)nullptr, (size_t)M, 0, 32,
                                  (hipStream_t) nullptr) != hipSuccess)
        return -1;
    return (int64_t)sz + 16;
}

extern "C" int hnr_sort_rows_by_key(const int32_t *d_keys, int64_t M, int32_t *d_keys_sorted, int32_t *d_perm, void *d_scratch,
                                    int64_t scratch_bytes, void *stream)
{
    if (M < 0) { set_error("hnr_sort_rows_by_key: bad size"); return HNR_ERR_BADARG; }
    if (M == 0) return HNR_OK;
    if (!d_keys || !d_keys_sorted || !d_perm || !d_scratch) { set_error("hnr_sort_rows_by_key: NULL argument"); return HNR_ERR_BADARG; }
    size_t sz = (size_t)scratch_bytes;
    rocprim::counting_iterator<int32_t> iota(0);
    // signed keys: negative (= skipped) keys sort first
    HNR_HIP_CHECK(rocprim::radix_sort_pairs(d_scratch, sz, d_keys, d_keys_sorted, iota, d_perm, (size_t)M, 0, 32, (hipStream_t)stream));
    return HNR_OK;
}

extern "C" int hnr_segment_sum_rows(const float *d_A, int lda, const float *d_B, int ldb, const int32_t *d_keys_sorted,
                                    const int32_t *d_perm, int64_t M, int n_cols, float *d_dst, int64_t dst_stride, void *stream)
{
    if (M < 0 || n_cols <= 0 || n_cols > 256 || (n_cols & 3) || lda < n_cols || (lda & 3) || (d_B && (ldb < n_cols || (ldb & 3))) ||
        dst_stride < n_cols) {
        set_error("hnr_segment_sum_rows: bad sizes (n_cols a multiple of 4, <= 256; row strides multiples of 4)"); return HNR_ERR_BADARG;
    }
    if (M == 0) return HNR_OK;
    if (!d_A || !d_keys_sorted || !d_perm || !d_dst) { set_error("hnr_segment_sum_rows: NULL argument"); return HNR_ERR_BADARG; }
    const int64_t waves = (M + SEG_CHUNK - 1) / SEG_CHUNK;
    segment_sum_rows_kernel<<<cdiv(waves * 64, 256), 256, 0, (hipStream_t)stream>>>(d_A, lda, d_B, ldb, d_keys_sorted, d_perm, M, n_cols, d_dst,
                                                                                    dst_stride);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// device-count form (csrc/render_train.hip): the grid is sized for keys_cap keys, the number of keys is read on the device
namespace hnr {
// sort of NON-NEGATIVE keys < 2^bits (compact point indices + a sentinel): fewer radix passes than the 32-bit signed sort
int sort_rows_by_key_bits(const int32_t *d_keys, int64_t M, int bits, int32_t *d_keys_sorted, int32_t *d_perm, void *d_scratch, int64_t scratch_bytes, hipStream_t st)
{
    if (M <= 0) return HNR_OK;
    size_t sz = (size_t)scratch_bytes;
    rocprim::counting_iterator<int32_t> iota(0);
    HNR_HIP_CHECK(rocprim::radix_sort_pairs(d_scratch, sz, reinterpret_cast<const uint32_t *>(d_keys), reinterpret_cast<uint32_t *>(d_keys_sorted), iota, d_perm, (size_t)M, 0,
                                            (unsigned)bits, st));
    return HNR_OK;
}
// start[k] = first sorted entry of key k (every dense key 0 .. n_keys-1 occurs at least once)
__global__ void segment_starts_kernel(const int32_t *__restrict__ keys_sorted, int64_t M, int32_t *__restrict__ start)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M) return;
    const int k = keys_sorted[e];
    if (k >= 0 && (e == 0 || keys_sorted[e - 1] != k)) start[k] = (int32_t)e;
}
int segment_starts(const int32_t *d_keys_sorted, int64_t M, int32_t *d_start, hipStream_t st)
{
    if (M <= 0) return HNR_OK;
    segment_starts_kernel<<<cdiv(M, 256), 256, 0, st>>>(d_keys_sorted, M, d_start);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
int segment_sum_rows_det_dc(const float *d_A, int lda, const int32_t *d_keys_sorted, const int32_t *d_perm, int64_t M, int n_cols, int keys_cap,
                            const long long *d_nkeys, const int32_t *d_start, float *d_dst, int64_t dst_stride, hipStream_t st)
{
    if (keys_cap <= 0) return HNR_OK;
    segment_sum_rows_det_kernel<<<cdiv((int64_t)keys_cap * 64, 256), 256, 0, st>>>(d_A, lda, d_keys_sorted, d_perm, M, n_cols, keys_cap, nullptr, d_dst, dst_stride, 0, d_nkeys,
                                                                                d_start);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
}  // namespace hnr

namespace hnr {
int segment_sum_rows_csr_dc(const float *d_A, int lda, const int32_t *d_row_list, const int32_t *d_seg_start, const int32_t *d_seg_count, int n_cols, int keys_cap,
                            const long long *d_nkeys, float *d_dst, int64_t dst_stride, const float *d_A2, int lda2, int n_cols2, float *d_dst2, int64_t dst_stride2,
                            uint32_t *d_absmax, hipStream_t st)
{
    if (keys_cap <= 0) return HNR_OK;
    if (d_A2 && (n_cols2 <= 0 || n_cols2 > 256 || (n_cols2 & 3))) { set_error("segment_sum_rows_csr: bad second matrix"); return HNR_ERR_BADARG; }
    int64_t blocks = (keys_cap + 3) / 4;             // four keys per workgroup round, a fixed grid striding over the keys the device count leaves
    if (blocks > 4096) blocks = 4096;
    segment_sum_rows_csr_kernel<<<(int)blocks, 256, 0, st>>>(d_A, lda, d_row_list, d_seg_start, d_seg_count, n_cols, keys_cap, d_nkeys, d_dst, dst_stride,
                                                                                   d_A2, lda2, d_A2 ? n_cols2 : 0, d_dst2, dst_stride2, d_absmax);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
}  // namespace hnr
