// Voxel grid over the neural point cloud, built once per cloud version.
//
// Replaces build_occ_vox = claim_occ + map_coor2occ + fill_occ2pnts of the reference
// (models/neural_points/query_point_indices_worldcoords.py:540-602, kernels :237-381), which
// allocates three dense X*Y*Z int32 grids plus a [max_o,P] table and refills them for every
// 2304-ray chunk.  Here the table is a sorted CSR behind a 1-bit-per-cell brick index:
//   occ_rec / dil : 4x4x4-cell bricks, one 64-bit word each (16 B + 8 B per 64 cells),
//   cell_rng      : {start,count} per occupied cell,
//   pts           : float4 {x,y,z,id} sorted by cell, point-id order inside a cell,
// so a voxel's candidates are one contiguous burst and a 3^3 neighbourhood touches <= 8 words.
//
// Semantics kept from the reference under the serial linearisation documented in
// oracle/query_oracle.c: lists hold the FIRST P points of a voxel in point-id order; voxels beyond
// max_o in first-appearance order are dropped; the voxel that would own slot 0 (the voxel of the
// first in-bounds point) keeps its occupancy but never lists points (`voxel_idx > 0`, :366).
#include <stdarg.h>
#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "hnr_common.h"

namespace hnr {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ---------------------------------------------------------------------------------- kernels
__global__ void bounds_kernel(const float *__restrict__ xyz, int n, float *out6)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = xyz[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        // float atomic min/max through the ordered-int trick
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int imn = __float_as_int(mn[a]), imx = __float_as_int(mx[a]);
            if (imn >= 0) atomicMin((int *)out6 + a, imn); else atomicMax((unsigned *)out6 + a, (unsigned)imn);
            if (imx >= 0) atomicMax((int *)out6 + 3 + a, imx); else atomicMin((unsigned *)out6 + 3 + a, (unsigned)imx);
        }
    }
}

__global__ void bounds_init_kernel(float *out6)
{
    if (threadIdx.x < 3) out6[threadIdx.x] = INFINITY;
    else if (threadIdx.x < 6) out6[threadIdx.x] = -INFINITY;
}

// pass 1 (claim_occ :237-297): mark the cell of every in-bounds point; remember the first in-bounds point.
__global__ void mark_cells_kernel(const float *__restrict__ xyz, int n, GridView g, unsigned long long *bits,
                                  int *first_inb, unsigned long long *n_inb)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool inb = false;
    if (i < n) {
        int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx);
        int y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy);
        int z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
        inb = in_bounds(g, x, y, z);
        if (inb) {
            atomicOr(&bits[brick_word(g, x, y, z)], 1ull << brick_bit(x, y, z));
            atomicMin(first_inb, i);
        }
    }
    unsigned long long b = __ballot(inb);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_inb, (unsigned long long)__popcll(b));
}

__global__ void popc_kernel(const unsigned long long *__restrict__ bits, uint32_t n_words, uint32_t *cnt)
{
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) cnt[w] = (uint32_t)__popcll(bits[w]);
}

__device__ __forceinline__ bool slot_of_point(const float *xyz, int i, const GridView &g,
                                              const unsigned long long *bits, const uint32_t *prefix, uint32_t &slot)
{
    int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx);
    int y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy);
    int z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
    if (!in_bounds(g, x, y, z)) return false;
    uint32_t w = brick_word(g, x, y, z);
    int b = brick_bit(x, y, z);
    unsigned long long bb = bits[w];
    if (!((bb >> b) & 1ull)) return false;
    slot = prefix[w] + (uint32_t)__popcll(bb & ((1ull << b) - 1ull));
    return true;
}

// max_o overflow only: first point id of every voxel (= first-appearance order of the serial claim_occ)
__global__ void first_id_kernel(const float *__restrict__ xyz, int n, GridView g, const unsigned long long *bits,
                                const uint32_t *prefix, uint32_t *first_id)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t slot;
    if (i < n && slot_of_point(xyz, i, g, bits, prefix, slot)) atomicMin(&first_id[slot], (uint32_t)i);
}

// keys for the sort: slot of the point's voxel, or 0xFFFFFFFF when the point is not listed anywhere
__global__ void assign_keys_kernel(const float *__restrict__ xyz, int n, GridView g, const unsigned long long *bits,
                                   const uint32_t *prefix, uint32_t *keys, uint32_t *vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t slot;
    keys[i] = slot_of_point(xyz, i, g, bits, prefix, slot) ? slot : 0xFFFFFFFFu;
    vals[i] = (uint32_t)i;
}

// overflow: clear the bit of every voxel whose first point id is beyond the max_o-th smallest.
// Reads the pre-kill snapshot (bits_in) and writes bits_out, so slots stay consistent.
__global__ void kill_overflow_kernel(const float *__restrict__ xyz, int n, GridView g,
                                     const unsigned long long *bits_in, unsigned long long *bits_out,
                                     const uint32_t *prefix, const uint32_t *first_id,
                                     const uint32_t *sorted_first, int max_o)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t slot;
    if (!slot_of_point(xyz, i, g, bits_in, prefix, slot)) return;
    if (first_id[slot] != (uint32_t)i) return;          // one thread per voxel: its first point
    if ((uint32_t)i <= sorted_first[max_o - 1]) return; // kept
    int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx);
    int y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy);
    int z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
    atomicAnd(&bits_out[brick_word(g, x, y, z)], ~(1ull << brick_bit(x, y, z)));
}

__global__ void cell_bounds_kernel(const uint32_t *__restrict__ keys, int n_listed, int *start, int *end)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_listed) return;
    uint32_t k = keys[j];
    if (j == 0 || keys[j - 1] != k) start[k] = j;
    if (j == n_listed - 1 || keys[j + 1] != k) end[k] = j + 1;
}

__global__ void cell_rng_kernel(const int *__restrict__ start, const int *__restrict__ end, int n_occ, int P,
                                int slot0, int2 *cell_rng, unsigned long long *n_over_p)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool over = false;
    if (s < n_occ) {
        int c = end[s] - start[s];
        over = c > P && s != slot0;
        if (c > P) c = P;
        if (s == slot0) c = 0;                     // `if (voxel_idx > 0)`  (:366): slot 0 is never filled
        cell_rng[s] = make_int2(start[s], c);
    }
    unsigned long long b = __ballot(over);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_over_p, (unsigned long long)__popcll(b));
}

__global__ void slot_of_first_kernel(const float *__restrict__ xyz, const int *first_inb, int n, GridView g,
                                     const unsigned long long *bits, const uint32_t *prefix, int *slot0)
{
    int i = *first_inb;
    uint32_t slot;
    *slot0 = (i >= 0 && i < n && slot_of_point(xyz, i, g, bits, prefix, slot)) ? (int)slot : -1;
}

__global__ void gather_pts_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ vals, int n_listed, float4 *pts)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_listed) return;
    uint32_t i = vals[j];
    pts[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float((int)i));
}

// map_coor2occ dilation (:324-332): every occupied voxel sets the query_size neighbourhood, clipped to dims.
__global__ void dilate_kernel(GridView g, const unsigned long long *__restrict__ bits, uint32_t n_words,
                              int bx, int qx, int qy, int qz, unsigned long long *dil)
{
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned long long bb = bits[w];
    if (!bb) return;
    (void)bx;
    int wz = w % g.bz, wy = (w / g.bz) % g.by, wx = w / (g.bz * g.by);
    while (bb) {
        int b = __ffsll((long long)bb) - 1;
        bb &= bb - 1;
        int x = wx * 4 + (b >> 4), y = wy * 4 + ((b >> 2) & 3), z = wz * 4 + (b & 3);
        int x0 = max(0, x - qx / 2), x1 = min(g.dx, x + (qx + 1) / 2);
        int y0 = max(0, y - qy / 2), y1 = min(g.dy, y + (qy + 1) / 2);
        int z0 = max(0, z - qz / 2), z1 = min(g.dz, z + (qz + 1) / 2);
        for (int xx = x0; xx < x1; ++xx)
            for (int yy = y0; yy < y1; ++yy)
                for (int zz = z0; zz < z1; ++zz)
                    atomicOr(&dil[brick_word(g, xx, yy, zz)], 1ull << brick_bit(xx, yy, zz));
    }
}

__global__ void pack_rec_kernel(const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ prefix,
                                uint32_t n_words, uint4 *rec)
{
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned long long b = bits[w];
    rec[w] = make_uint4((uint32_t)b, (uint32_t)(b >> 32), prefix[w], 0u);
}

__global__ void count_bits_kernel(const unsigned long long *__restrict__ bits, uint32_t n_words, unsigned long long *total)
{
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = (w < n_words) ? (unsigned long long)__popcll(bits[w]) : 0ull;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}

__global__ void export_dense_kernel(GridView g, uint8_t *coor_occ, int32_t *cell_count, int32_t *cell_first)
{
    int64_t vol = (int64_t)g.dx * g.dy * g.dz;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vol) return;
    int z = (int)(i % g.dz), y = (int)((i / g.dz) % g.dy), x = (int)(i / ((int64_t)g.dz * g.dy));
    uint32_t w = brick_word(g, x, y, z);
    int b = brick_bit(x, y, z);
    coor_occ[i] = (uint8_t)((g.dil[w] >> b) & 1ull);
    uint4 rec = g.occ_rec[w];
    unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
    if ((bb >> b) & 1ull) {
        uint32_t slot = rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull));
        int2 rg = g.cell_rng[slot];
        cell_count[i] = rg.y;
        cell_first[i] = rg.y > 0 ? __float_as_int(g.pts[rg.x].w) : -1;
    } else {
        cell_count[i] = -1;
        cell_first[i] = -1;
    }
}

__global__ void count_listed_kernel(const uint32_t *__restrict__ keys, int n, unsigned long long *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b = __ballot(i < n && keys[i] != 0xFFFFFFFFu);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}

// ---------------------------------------------------------------------------------- 3x3x3 neighbourhood lists (GridView::nb_*)
// One wave per brick of the dilated mask, lane = cell.  The 27 cells in the reference's order: x-major, then y, then z, the own cell pulled to the front
// (layer 0 before layer 1, query_point_indices_worldcoords.py:478-491); cells outside the grid or without points contribute nothing.
__device__ __forceinline__ bool nb_cell(const GridView &g, int vx, int vy, int vz, int2 &rg)
{
    if (!in_bounds(g, vx, vy, vz)) return false;
    const uint4 rec = g.occ_rec[brick_word(g, vx, vy, vz)];
    const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
    const int b = brick_bit(vx, vy, vz);
    if (!((bb >> b) & 1ull)) return false;
    rg = g.cell_rng[rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull))];
    return true;
}

template <bool FILL>
__global__ __launch_bounds__(256) void nb_lists_kernel(GridView g, const unsigned long long *__restrict__ dil, const uint32_t *__restrict__ dprefix, uint32_t n_words,
                                                       const uint32_t *__restrict__ nb_start, uint32_t *__restrict__ nb_cnt, uint2 *__restrict__ nb_rng,
                                                       float4 *__restrict__ nb_pts)
{
    const uint32_t w = (uint32_t)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n_words) return;
    const unsigned long long bb = dil[w];
    const int b = threadIdx.x & 63;
    if (!((bb >> b) & 1ull)) return;
    const int wz = (int)(w % (uint32_t)g.bz), wy = (int)((w / (uint32_t)g.bz) % (uint32_t)g.by), wx = (int)(w / ((uint32_t)g.bz * (uint32_t)g.by));
    const int x = wx * 4 + (b >> 4), y = wy * 4 + ((b >> 2) & 3), z = wz * 4 + (b & 3);
    const uint32_t slot = dprefix[w] + (uint32_t)__popcll(bb & ((1ull << b) - 1ull));
    uint32_t c0 = 0, total = 0, cells1 = 0, cell0 = 0;
    uint32_t out = FILL ? nb_start[slot] : 0u;
    int2 rg;
    if (nb_cell(g, x, y, z, rg)) {
        cell0 = 1; c0 = (uint32_t)rg.y; total = c0;
        if (FILL) for (int j = 0; j < rg.y; ++j) nb_pts[out++] = g.pts[rg.x + j];
    }
    // layout of a run (entries of 16 B): [own cell: c0 entries, padded to a multiple of 4][shell 1: total - c0 entries, padded to a multiple of 4] -- every
    // run and both of its parts start on a 64-byte line, so the four lanes of a quad that read four consecutive candidates touch ONE line (knn_quad_kernel)
    const uint32_t c0p = (c0 + 3u) & ~3u;
    if (FILL) out = nb_start[slot] + c0p;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                if (!nb_cell(g, x + dx, y + dy, z + dz, rg)) continue;
                ++cells1; total += (uint32_t)rg.y;
                if (FILL) for (int j = 0; j < rg.y; ++j) nb_pts[out++] = g.pts[rg.x + j];
            }
    if (FILL) nb_rng[slot] = make_uint2(nb_start[slot], c0 | (total << 6) | (cells1 << 17) | (cell0 << 22));
    else nb_cnt[slot] = c0p + ((total - c0 + 3u) & ~3u);
}

// brick_near (GridView): one thread per brick, OR over the 5x5x5 bricks around it of "the dilated mask has a cell here"
__global__ void brick_near_kernel(GridView g, const unsigned long long *__restrict__ dil, int bx, uint32_t n_words, uint8_t *__restrict__ near)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const int wz = (int)(w % (uint32_t)g.bz), wy = (int)((w / (uint32_t)g.bz) % (uint32_t)g.by), wx = (int)(w / ((uint32_t)g.bz * (uint32_t)g.by));
    bool any = false;
    for (int x = max(0, wx - 2); x <= min(bx - 1, wx + 2) && !any; ++x)
        for (int y = max(0, wy - 2); y <= min(g.by - 1, wy + 2) && !any; ++y)
            for (int z = max(0, wz - 2); z <= min(g.bz - 1, wz + 2); ++z)
                if (dil[((size_t)x * g.by + y) * g.bz + z] != 0ull) { any = true; break; }
    near[w] = any ? 1 : 0;
}

__global__ void pack_dil_rec_kernel(const unsigned long long *__restrict__ dil, const uint32_t *__restrict__ dprefix, uint32_t n_words, uint4 *rec)
{
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned long long b = dil[w];
    rec[w] = make_uint4((uint32_t)b, (uint32_t)(b >> 32), dprefix[w], 0u);
}

// ---------------------------------------------------------------------------------- host
template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
    T *release() { T *q = p; p = nullptr; return q; }
};

#define GB_CHECK(e)                                                                       \
    do {                                                                                  \
        hipError_t _e = (e);                                                              \
        if (_e != hipSuccess) {                                                           \
            set_error("hnr_grid_build: %s -> %s", #e, hipGetErrorString(_e));             \
            return (_e == hipErrorOutOfMemory) ? HNR_ERR_NOMEM : HNR_ERR_HIP;             \
        }                                                                                 \
    } while (0)

static int build_impl(hnr_grid *g, const float *d_xyz, int n, hipStream_t st)
{
    const hnr_grid_params *p = &g->p;
    const uint32_t n_words = g->n_words;
    const GridView v = g->view();          // table pointers are still null: only geometry is used below
    const int TB = 256;

    DevBuf<unsigned long long> bits, bits2, dil, scal;   // scal: [0]=n_inb [1]=n_over_p [2]=n_dilated [3]=n_listed
    DevBuf<uint32_t> cnt, prefix, keys, vals, keys2, vals2, first_id, first_sorted;
    DevBuf<int> ints;                                    // [0]=first in-bounds point id, [1]=its slot
    DevBuf<int> start, end;
    DevBuf<char> tmp;

    GB_CHECK(bits.alloc(n_words));
    GB_CHECK(cnt.alloc(n_words));
    GB_CHECK(prefix.alloc(n_words));
    GB_CHECK(scal.alloc(4));
    GB_CHECK(ints.alloc(2));
    GB_CHECK(keys.alloc(n)); GB_CHECK(vals.alloc(n)); GB_CHECK(keys2.alloc(n)); GB_CHECK(vals2.alloc(n));
    GB_CHECK(hipMemsetAsync(bits.p, 0, (size_t)n_words * 8, st));
    GB_CHECK(hipMemsetAsync(scal.p, 0, 4 * 8, st));
    const int big = 0x7fffffff;
    GB_CHECK(hipMemcpyAsync(ints.p, &big, sizeof(int), hipMemcpyHostToDevice, st));

    // one scratch buffer for every rocprim call below
    size_t s1 = 0, s2 = 0, s3 = 0;
    GB_CHECK(rocprim::exclusive_scan(nullptr, s1, cnt.p, prefix.p, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st));
    GB_CHECK(rocprim::radix_sort_pairs(nullptr, s2, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 32, st));
    GB_CHECK(rocprim::radix_sort_keys(nullptr, s3, keys.p, keys2.p, (size_t)n, 0, 32, st));
    size_t tmp_bytes = s1 > s2 ? s1 : s2;
    if (s3 > tmp_bytes) tmp_bytes = s3;
    GB_CHECK(tmp.alloc(tmp_bytes));

    // pass 1: which cells hold points
    mark_cells_kernel<<<cdiv(n, TB), TB, 0, st>>>(d_xyz, n, v, bits.p, ints.p, scal.p);
    GB_CHECK(hipGetLastError());

    auto scan_words = [&](uint32_t &n_occ_out) -> hipError_t {
        popc_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(bits.p, n_words, cnt.p);
        size_t sz = tmp_bytes;
        hipError_t e = rocprim::exclusive_scan((void *)tmp.p, sz, cnt.p, prefix.p, 0u, (size_t)n_words,
                                               rocprim::plus<uint32_t>(), st);
        if (e != hipSuccess) return e;
        uint32_t last_p = 0, last_c = 0;
        e = hipMemcpyAsync(&last_p, prefix.p + (n_words - 1), 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(&last_c, cnt.p + (n_words - 1), 4, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        n_occ_out = last_p + last_c;
        return e;
    };

    uint32_t n_occ = 0;
    GB_CHECK(scan_words(n_occ));
    int64_t n_dropped = 0;
    if (n_occ > (uint32_t)p->max_o) {
        // keep the max_o voxels that appear first in point-id order (serial claim_occ order)
        GB_CHECK(first_id.alloc(n_occ));
        GB_CHECK(first_sorted.alloc(n_occ));
        GB_CHECK(bits2.alloc(n_words));
        GB_CHECK(hipMemsetAsync(first_id.p, 0xff, (size_t)n_occ * 4, st));
        first_id_kernel<<<cdiv(n, TB), TB, 0, st>>>(d_xyz, n, v, bits.p, prefix.p, first_id.p);
        size_t sz = tmp_bytes;   // n_occ <= n, so the scratch sized for n keys is enough
        GB_CHECK(rocprim::radix_sort_keys((void *)tmp.p, sz, first_id.p, first_sorted.p, (size_t)n_occ, 0, 32, st));
        GB_CHECK(hipMemcpyAsync(bits2.p, bits.p, (size_t)n_words * 8, hipMemcpyDeviceToDevice, st));
        kill_overflow_kernel<<<cdiv(n, TB), TB, 0, st>>>(d_xyz, n, v, bits.p, bits2.p, prefix.p, first_id.p,
                                                        first_sorted.p, p->max_o);
        GB_CHECK(hipGetLastError());
        n_dropped = (int64_t)n_occ - p->max_o;
        GB_CHECK(hipMemcpyAsync(bits.p, bits2.p, (size_t)n_words * 8, hipMemcpyDeviceToDevice, st));
        GB_CHECK(scan_words(n_occ));
    }

    // pass 2: sort points by voxel slot (radix sort is stable: point-id order inside a voxel)
    assign_keys_kernel<<<cdiv(n, TB), TB, 0, st>>>(d_xyz, n, v, bits.p, prefix.p, keys.p, vals.p);
    GB_CHECK(hipGetLastError());
    {
        size_t sz = tmp_bytes;
        GB_CHECK(rocprim::radix_sort_pairs((void *)tmp.p, sz, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 32, st));
    }
    count_listed_kernel<<<cdiv(n, TB), TB, 0, st>>>(keys2.p, n, scal.p + 3);
    slot_of_first_kernel<<<1, 1, 0, st>>>(d_xyz, ints.p, n, v, bits.p, prefix.p, ints.p + 1);
    GB_CHECK(hipGetLastError());
    unsigned long long h_scal[4];
    int h_ints[2];
    GB_CHECK(hipMemcpyAsync(h_scal, scal.p, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    GB_CHECK(hipMemcpyAsync(h_ints, ints.p, sizeof(h_ints), hipMemcpyDeviceToHost, st));
    GB_CHECK(hipStreamSynchronize(st));
    const int n_listed = (int)h_scal[3];
    const int slot0 = h_ints[1];

    // pass 3: CSR ranges, packed points, dilated mask, packed brick records
    DevBuf<int2> cell_rng;
    DevBuf<float4> pts;
    DevBuf<uint4> rec;
    GB_CHECK(start.alloc(n_occ)); GB_CHECK(end.alloc(n_occ));
    GB_CHECK(cell_rng.alloc(n_occ));
    GB_CHECK(pts.alloc(n_listed));
    GB_CHECK(rec.alloc(n_words));
    GB_CHECK(dil.alloc(n_words));
    GB_CHECK(hipMemsetAsync(dil.p, 0, (size_t)n_words * 8, st));
    if (n_listed > 0) {
        cell_bounds_kernel<<<cdiv(n_listed, TB), TB, 0, st>>>(keys2.p, n_listed, start.p, end.p);
        gather_pts_kernel<<<cdiv(n_listed, TB), TB, 0, st>>>(d_xyz, vals2.p, n_listed, pts.p);
    }
    if (n_occ > 0)
        cell_rng_kernel<<<cdiv(n_occ, TB), TB, 0, st>>>(start.p, end.p, (int)n_occ, p->P, slot0, cell_rng.p, scal.p + 1);
    dilate_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(v, bits.p, n_words, g->bd[0], p->query_size[0], p->query_size[1],
                                                   p->query_size[2], dil.p);
    count_bits_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(dil.p, n_words, scal.p + 2);
    pack_rec_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(bits.p, prefix.p, n_words, rec.p);
    GB_CHECK(hipGetLastError());
    GB_CHECK(hipMemcpyAsync(h_scal, scal.p, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    GB_CHECK(hipStreamSynchronize(st));

    g->occ_rec = rec.release();
    g->dil = dil.release();
    g->cell_rng = cell_rng.release();
    g->pts = pts.release();
    // ---- brick-level "anything near" mask for the march's coarse level (GridView::brick_near).  Only with HNR_MARCH_TWO_LEVEL=1: the two-level march
    //      returns the same samples (tests/test_query_gpu.py runs it) and is NOT faster on the bench frame (0.242 vs 0.233 ms: a coarse probe costs what
    //      a fine one does -- the three exact divisions -- and too many groups of a cluttered room lie within two bricks of the mask)
    if (const char *e2 = getenv("HNR_MARCH_TWO_LEVEL"); e2 && atoi(e2) != 0) {
        DevBuf<uint8_t> nearb;
        GB_CHECK(nearb.alloc(n_words));
        const GridView v1 = g->view();
        brick_near_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(v1, g->dil, g->bd[0], n_words, nearb.p);
        GB_CHECK(hipGetLastError());
        g->brick_near = nearb.release();
    }
    // ---- 3x3x3 neighbourhood lists for the k-NN (hnr_common.h, GridView::nb_*): count, scan, fill.  Needs P <= 63 (the packed record) and the lists to
    //      fit 32-bit indices; HNR_NB_LISTS=0 skips them (the k-NN then walks the 27 cells itself: knn3_kernel).
    int64_t nb_bytes = 0;
    {
        const char *e = getenv("HNR_NB_LISTS");
        const int64_t n_dil = (int64_t)h_scal[2];
        if (!(e && atoi(e) == 0) && p->P <= 63 && n_dil > 0 && n_listed > 0 && 27ll * n_listed + 8ll * n_dil < (1ll << 31)) {
            DevBuf<uint32_t> dcnt, dprefix, nb_cnt, nb_start;
            DevBuf<uint4> drec;
            DevBuf<uint2> nbr;
            DevBuf<float4> nbp;
            DevBuf<char> tmp2;
            GB_CHECK(dcnt.alloc(n_words)); GB_CHECK(dprefix.alloc(n_words));
            GB_CHECK(nb_cnt.alloc(n_dil + 1)); GB_CHECK(nb_start.alloc(n_dil + 1));
            GB_CHECK(drec.alloc(n_words)); GB_CHECK(nbr.alloc(n_dil));
            size_t t1 = 0, t2 = 0;
            GB_CHECK(rocprim::exclusive_scan(nullptr, t1, dcnt.p, dprefix.p, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st));
            GB_CHECK(rocprim::exclusive_scan(nullptr, t2, nb_cnt.p, nb_start.p, 0u, (size_t)n_dil + 1, rocprim::plus<uint32_t>(), st));
            GB_CHECK(tmp2.alloc(t1 > t2 ? t1 : t2));
            popc_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(g->dil, n_words, dcnt.p);
            { size_t sz = t1; GB_CHECK(rocprim::exclusive_scan((void *)tmp2.p, sz, dcnt.p, dprefix.p, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st)); }
            GB_CHECK(hipMemsetAsync(nb_cnt.p, 0, (size_t)(n_dil + 1) * 4, st));
            const GridView v2 = g->view();                                  // (occ_rec, cell_rng, pts are in place; the nb_* pointers still null)
            const int nb_blocks = cdiv((int64_t)n_words * 64, 256);
            nb_lists_kernel<false><<<nb_blocks, 256, 0, st>>>(v2, g->dil, dprefix.p, n_words, nullptr, nb_cnt.p, nullptr, nullptr);
            { size_t sz = t2; GB_CHECK(rocprim::exclusive_scan((void *)tmp2.p, sz, nb_cnt.p, nb_start.p, 0u, (size_t)n_dil + 1, rocprim::plus<uint32_t>(), st)); }
            uint32_t nb_total = 0;
            GB_CHECK(hipMemcpyAsync(&nb_total, nb_start.p + n_dil, 4, hipMemcpyDeviceToHost, st));
            GB_CHECK(hipStreamSynchronize(st));
            // + 4 zeroed records: the quad / XP k-NN forms read four records from a run's shell-1 start, which is nb_total for an empty last run
            GB_CHECK(nbp.alloc((size_t)nb_total + 4));
            GB_CHECK(hipMemsetAsync(nbp.p + nb_total, 0, 4 * sizeof(float4), st));
            nb_lists_kernel<true><<<nb_blocks, 256, 0, st>>>(v2, g->dil, dprefix.p, n_words, nb_start.p, nullptr, nbr.p, nbp.p);
            pack_dil_rec_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(g->dil, dprefix.p, n_words, drec.p);
            GB_CHECK(hipGetLastError());
            GB_CHECK(hipStreamSynchronize(st));                             // (the scratch buffers above die with this scope)
            g->dil_rec = drec.release(); g->nb_rng = nbr.release(); g->nb_pts = nbp.release();
            nb_bytes = (int64_t)n_words * 16 + n_dil * 8 + ((int64_t)nb_total + 4) * 16;
        }
    }
    g->st.n_points = n;
    g->st.n_inbounds = (int64_t)h_scal[0];
    g->st.n_occ = n_occ;
    g->st.n_dropped_voxels = n_dropped;
    g->st.n_cells_over_P = (int64_t)h_scal[1];
    g->st.n_dilated = (int64_t)h_scal[2];
    g->st.n_words = n_words;
    g->st.bytes = (int64_t)n_words * (16 + 8 + 1) + (int64_t)n_occ * 8 + (int64_t)n_listed * 16 + nb_bytes;
    return HNR_OK;
}

}  // namespace hnr

using namespace hnr;

extern "C" const char *hnr_version(void) { return "hnr-hip 0.1.0 gfx950"; }
extern "C" const char *hnr_last_error(void) { return hnr::last_error(); }

extern "C" int hnr_points_bounds(const float *d_xyz, int n, float *d_out6, void *stream)
{
    if (!d_xyz || !d_out6 || n <= 0) { set_error("hnr_points_bounds: bad argument"); return HNR_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    bounds_init_kernel<<<1, 64, 0, st>>>(d_out6);
    int blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
    bounds_kernel<<<blocks, 256, 0, st>>>(d_xyz, n, d_out6);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_grid_free(hnr_grid *g)
{
    if (!g) return HNR_OK;
    if (g->occ_rec) (void)hipFree(g->occ_rec);
    if (g->dil) (void)hipFree(g->dil);
    if (g->cell_rng) (void)hipFree(g->cell_rng);
    if (g->pts) (void)hipFree(g->pts);
    if (g->dil_rec) (void)hipFree(g->dil_rec);
    if (g->nb_rng) (void)hipFree(g->nb_rng);
    if (g->nb_pts) (void)hipFree(g->nb_pts);
    if (g->brick_near) (void)hipFree(g->brick_near);
    delete g;
    return HNR_OK;
}

extern "C" int hnr_grid_get_stats(const hnr_grid *g, hnr_grid_stats *out)
{
    if (!g || !out) return HNR_ERR_BADARG;
    *out = g->st;
    return HNR_OK;
}

extern "C" int hnr_grid_get_params(const hnr_grid *g, hnr_grid_params *out)
{
    if (!g || !out) return HNR_ERR_BADARG;
    *out = g->p;
    return HNR_OK;
}

extern "C" int hnr_grid_build(const float *d_xyz, int n, const hnr_grid_params *p, void *stream, hnr_grid **out)
{
    if (out) *out = nullptr;
    if (!d_xyz || !p || !out || n <= 0 || p->P <= 0 || p->max_o <= 0) {
        set_error("hnr_grid_build: bad argument"); return HNR_ERR_BADARG;
    }
    for (int a = 0; a < 3; ++a)
        if (p->dims[a] <= 0 || !(p->cell[a] > 0.0f) || p->query_size[a] < 0) {
            set_error("hnr_grid_build: dims/cell must be positive, query_size non-negative"); return HNR_ERR_BADARG;
        }
    for (int a = 0; a < 3; ++a)
        if (p->dims[a] >= (1 << 24)) {          // (the march tests cell indices as floats: exact below 2^24)
            set_error("hnr_grid_build: grid of %d x %d x %d cells is too large", p->dims[0], p->dims[1], p->dims[2]); return HNR_ERR_TOOBIG;
        }
    int bd[3];
    for (int a = 0; a < 3; ++a) bd[a] = (p->dims[a] + 3) / 4;
    int64_t nw64 = (int64_t)bd[0] * bd[1] * bd[2];
    if (nw64 >= (1ll << 31)) {
        set_error("hnr_grid_build: grid of %d x %d x %d cells is too large", p->dims[0], p->dims[1], p->dims[2]);
        return HNR_ERR_TOOBIG;
    }
    hnr_grid *g = new hnr_grid();
    memset((void *)g, 0, sizeof(*g));
    g->p = *p;
    memcpy(g->bd, bd, sizeof(bd));
    g->n_words = (uint32_t)nw64;
    int rc = build_impl(g, d_xyz, n, (hipStream_t)stream);
    if (rc != HNR_OK) { hnr_grid_free(g); return rc; }
    *out = g;
    return HNR_OK;
}

extern "C" int hnr_grid_export_dense(const hnr_grid *g, uint8_t *d_coor_occ, int32_t *d_cell_count,
                                     int32_t *d_cell_first, void *stream)
{
    if (!g || !d_coor_occ || !d_cell_count || !d_cell_first) { set_error("hnr_grid_export_dense: bad argument"); return HNR_ERR_BADARG; }
    int64_t vol = (int64_t)g->p.dims[0] * g->p.dims[1] * g->p.dims[2];
    if (vol >= (1ll << 31) * 256ll) { set_error("hnr_grid_export_dense: volume too large"); return HNR_ERR_TOOBIG; }
    export_dense_kernel<<<cdiv(vol, 256), 256, 0, (hipStream_t)stream>>>(g->view(), d_coor_occ, d_cell_count, d_cell_first);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
