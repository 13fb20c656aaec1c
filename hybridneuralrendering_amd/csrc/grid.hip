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

#include <utility>

#include "hnr_common.h"

namespace hnr {

void grid_upd_free(hnr_grid_upd *u);

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
    // (one atomicMin / atomicAdd per WORKGROUP on the two scalars: per in-bounds thread they were 2 M same-address atomics, 0.4 ms of the build)
    __shared__ int s_first[4];
    __shared__ int s_cnt[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool inb = false;
    if (i < n) {
        int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx);
        int y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy);
        int z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
        inb = in_bounds(g, x, y, z);
        if (inb) atomicOr(&bits[brick_word(g, x, y, z)], 1ull << brick_bit(x, y, z));
    }
    const unsigned long long b = __ballot(inb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[wave] = __popcll(b);
        s_first[wave] = b ? (int)(blockIdx.x * blockDim.x + wave * 64 + (__ffsll((long long)b) - 1)) : 0x7fffffff;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        int c = 0, f = 0x7fffffff;
        for (int w = 0; w < nw; ++w) { c += s_cnt[w]; f = min(f, s_first[w]); }
        if (c) { atomicAdd(n_inb, (unsigned long long)c); atomicMin(first_inb, f); }
    }
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
                                int slot0, int2 *cell_rng, uint32_t *cell_total, unsigned long long *n_over_p)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool over = false;
    if (s < n_occ) {
        int c = end[s] - start[s];
        cell_total[s] = (uint32_t)c;               // unclamped: hnr_grid_grow needs to know whether a list is full
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
    const bool have = i >= 0 && i < n && slot_of_point(xyz, i, g, bits, prefix, slot);
    slot0[0] = have ? (int)slot : -1;
    slot0[1] = slot0[2] = -1;                      // the cell itself (brick word, bit): hnr_grid_grow re-derives the slot after cells were inserted
    if (have) {
        const int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx), y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy), z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
        slot0[1] = (int)brick_word(g, x, y, z); slot0[2] = brick_bit(x, y, z);
    }
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

// listed keys come first in the sorted array (0xFFFFFFFF = not listed sorts last): their number is the position of the first 0xFFFFFFFF -- one thread's
// binary search (a counting kernel with one atomic per wave on one address took 0.38 ms)
__global__ void count_listed_kernel(const uint32_t *__restrict__ keys_sorted, int n, unsigned long long *out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys_sorted[mid] != 0xFFFFFFFFu) lo = mid + 1; else hi = mid;
    }
    *out = (unsigned long long)lo;
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
    DevBuf<int> ints;                                    // [0]=first in-bounds point id, [1]=its slot, [2], [3]=its brick word, bit
    DevBuf<int> start, end;
    DevBuf<char> tmp;

    GB_CHECK(bits.alloc(n_words));
    GB_CHECK(cnt.alloc(n_words));
    GB_CHECK(prefix.alloc(n_words));
    GB_CHECK(scal.alloc(4));
    GB_CHECK(ints.alloc(4));
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
    count_listed_kernel<<<1, 64, 0, st>>>(keys2.p, n, scal.p + 3);
    slot_of_first_kernel<<<1, 1, 0, st>>>(d_xyz, ints.p, n, v, bits.p, prefix.p, ints.p + 1);
    GB_CHECK(hipGetLastError());
    unsigned long long h_scal[4];
    int h_ints[4];
    GB_CHECK(hipMemcpyAsync(h_scal, scal.p, sizeof(h_scal), hipMemcpyDeviceToHost, st));
    GB_CHECK(hipMemcpyAsync(h_ints, ints.p, sizeof(h_ints), hipMemcpyDeviceToHost, st));
    GB_CHECK(hipStreamSynchronize(st));
    const int n_listed = (int)h_scal[3];
    const int slot0 = h_ints[1];

    // pass 3: CSR ranges, packed points, dilated mask, packed brick records
    DevBuf<int2> cell_rng;
    DevBuf<float4> pts;
    DevBuf<uint4> rec;
    DevBuf<uint32_t> cell_total;
    // slack behind the compact tables (HNR_GRID_SLACK percent, default 25): room for hnr_grid_grow to append lists / runs without reallocating
    int slack_pct = 25;
    if (const char *es = getenv("HNR_GRID_SLACK")) { slack_pct = atoi(es); if (slack_pct < 0) slack_pct = 0; if (slack_pct > 400) slack_pct = 400; }
    auto with_slack = [&](uint64_t n_items, uint64_t floor_items) -> uint64_t { return n_items + (n_items * (uint64_t)slack_pct) / 100 + (slack_pct ? floor_items : 0); };
    const uint64_t occ_cap = with_slack(n_occ, 1024), pts_cap = with_slack((uint64_t)n_listed, 4096);
    GB_CHECK(start.alloc(n_occ)); GB_CHECK(end.alloc(n_occ));
    GB_CHECK(cell_rng.alloc(occ_cap));
    GB_CHECK(cell_total.alloc(occ_cap));
    GB_CHECK(pts.alloc(pts_cap));
    GB_CHECK(rec.alloc(n_words));
    GB_CHECK(dil.alloc(n_words));
    GB_CHECK(hipMemsetAsync(dil.p, 0, (size_t)n_words * 8, st));
    if (n_listed > 0) {
        cell_bounds_kernel<<<cdiv(n_listed, TB), TB, 0, st>>>(keys2.p, n_listed, start.p, end.p);
        gather_pts_kernel<<<cdiv(n_listed, TB), TB, 0, st>>>(d_xyz, vals2.p, n_listed, pts.p);
    }
    if (n_occ > 0)
        cell_rng_kernel<<<cdiv(n_occ, TB), TB, 0, st>>>(start.p, end.p, (int)n_occ, p->P, slot0, cell_rng.p, cell_total.p, scal.p + 1);
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
    g->cell_total = cell_total.release();
    g->n_occ = n_occ; g->occ_cap = (uint32_t)occ_cap; g->pts_used = (uint32_t)n_listed; g->pts_cap = (uint32_t)pts_cap;
    g->first_inb = h_ints[0]; g->slot0_word = h_ints[2]; g->slot0_bit = h_ints[3];
    g->n_dil = 0; g->nb_used = g->nb_cap = g->dil_cap = 0;
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
            const uint64_t dil_cap = with_slack((uint64_t)n_dil, 1024);
            GB_CHECK(drec.alloc(n_words)); GB_CHECK(nbr.alloc(dil_cap));
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
            uint64_t nb_cap = with_slack((uint64_t)nb_total, 65536) + 4;
            if (nb_cap >= (1ull << 31)) nb_cap = (uint64_t)nb_total + 4;      // (indices stay below 2^31)
            GB_CHECK(nbp.alloc((size_t)nb_cap));
            GB_CHECK(hipMemsetAsync(nbp.p + nb_total, 0, 4 * sizeof(float4), st));
            nb_lists_kernel<true><<<nb_blocks, 256, 0, st>>>(v2, g->dil, dprefix.p, n_words, nb_start.p, nullptr, nbr.p, nbp.p);
            pack_dil_rec_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(g->dil, dprefix.p, n_words, drec.p);
            GB_CHECK(hipGetLastError());
            GB_CHECK(hipStreamSynchronize(st));                             // (the scratch buffers above die with this scope)
            g->dil_rec = drec.release(); g->nb_rng = nbr.release(); g->nb_pts = nbp.release();
            g->n_dil = (uint32_t)n_dil; g->dil_cap = (uint32_t)dil_cap; g->nb_used = nb_total; g->nb_cap = (uint32_t)nb_cap;
            nb_bytes = (int64_t)n_words * 16 + (int64_t)dil_cap * 8 + (int64_t)nb_cap * 16;
        }
    }
    g->st.n_points = n;
    g->st.n_inbounds = (int64_t)h_scal[0];
    g->st.n_occ = n_occ;
    g->st.n_dropped_voxels = n_dropped;
    g->st.n_cells_over_P = (int64_t)h_scal[1];
    g->st.n_dilated = (int64_t)h_scal[2];
    g->st.n_words = n_words;
    g->st.bytes = (int64_t)n_words * (16 + 8 + 1) + (int64_t)occ_cap * (8 + 4) + (int64_t)pts_cap * 16 + nb_bytes;
    return HNR_OK;
}


// ---------------------------------------------------------------------------------- incremental update after grow_points
// The reference appends new points (models/neural_points/neural_points.py:376-402) and, because it rebuilds its tables for every chunk anyway, has no
// notion of updating them; run/train_ft.py:926-952 saves and exits instead.  Here the tables are per cloud version, so a grown cloud either rebuilds
// them (hnr_grid_build: a sort of all N points, the 0.9 GB of neighbourhood lists) or, when the grid geometry (origin / cell / dims) is unchanged,
// extends them: new points have larger ids than all old ones, so every list only ever gets entries APPENDED (lists hold the first P points of a cell
// in id order), occupancy and the dilated mask only gain bits, and
//   * the brick records (bits + prefix) and the two slot-indexed range tables are rewritten whole (a few MB: slots shift when cells are inserted);
//   * a cell whose list changed gets its new list at the END of `pts` (old entries copied, new points behind them), every other cell keeps its range;
//   * a cell of the dilated mask whose 3x3x3 neighbourhood holds a changed cell (or that is newly dilated) gets its run rebuilt at the END of `nb_pts`,
//     every other run stays where it is.
// The result is LOGICALLY what hnr_grid_build produces for the grown cloud -- same mask, same lists and runs in the same order, same counters (the
// tests compare them table by table and query by query) -- in a layout that is no longer compact: the superseded lists / runs stay behind as holes in
// the slack the build left, and when the slack is used up (or max_o would be exceeded) the call reports HNR_NEED_REBUILD and changes nothing.
struct hnr_grid_upd_t {
    unsigned long long *bits = nullptr, *dil = nullptr, *touched = nullptr, *ddirty = nullptr;     // [n_words]
    uint32_t *cnt = nullptr, *prefix = nullptr, *dcnt = nullptr, *dprefix = nullptr;                // [n_words]
    uint4 *occ_rec = nullptr, *dil_rec = nullptr;                                                   // spare tables [n_words]
    int2 *cell_rng = nullptr; uint32_t *cell_total = nullptr, *add_len = nullptr, *add_off = nullptr; int32_t *old_of_new = nullptr;   // [occ_cap]
    uint2 *nb_rng = nullptr; uint32_t *run_len = nullptr, *run_off = nullptr;                       // [dil_cap]
    uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr, *rank = nullptr; // [new_cap]
    char *tmp = nullptr; size_t tmp_bytes = 0;
    char *arena = nullptr;                                                                          // one allocation behind the scratch arrays (not the spare tables)
    unsigned long long *scal = nullptr;                                                             // device scalars [8]
    int new_cap = 0;
};
}  // namespace hnr
struct hnr_grid_upd : hnr::hnr_grid_upd_t {};
namespace hnr {

void grid_upd_free(hnr_grid_upd *u)
{
    if (!u) return;
    void *all[] = {u->arena, u->dil, u->occ_rec, u->dil_rec, u->cell_rng, u->cell_total, u->nb_rng, u->keys, u->vals, u->keys2, u->vals2, u->rank, u->tmp};
    for (void *q : all) if (q) (void)hipFree(q);
    delete u;
}

enum { US_INB = 0, US_OVERP, US_NOCC, US_NDIL, US_PTS_ADD, US_NB_ADD, US_FIRST_NEW, US_N };

// working copies of the two masks; nothing touched yet
__global__ void upd_init_kernel(const uint4 *__restrict__ occ_rec, const unsigned long long *__restrict__ dil_old, uint32_t n_words, unsigned long long *bits,
                                unsigned long long *dil, unsigned long long *touched, unsigned long long *ddirty)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint4 r = occ_rec[w];
    bits[w] = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
    dil[w] = dil_old[w];
    touched[w] = 0ull; ddirty[w] = 0ull;
}

// claim_occ for the new points only: occupancy bits, the cells that received a point, in-bounds count
__global__ void upd_mark_kernel(const float *__restrict__ xyz, int n_old, int n, GridView g, unsigned long long *bits, unsigned long long *touched, unsigned long long *scal)
{
    const int i = n_old + blockIdx.x * blockDim.x + threadIdx.x;
    bool inb = false;
    if (i < n) {
        const int x = cell_coord(xyz[3 * (size_t)i], g.ox, g.cx), y = cell_coord(xyz[3 * (size_t)i + 1], g.oy, g.cy), z = cell_coord(xyz[3 * (size_t)i + 2], g.oz, g.cz);
        inb = in_bounds(g, x, y, z);
        if (inb) {
            const uint32_t w = brick_word(g, x, y, z);
            const unsigned long long m = 1ull << brick_bit(x, y, z);
            atomicOr(&bits[w], m);
            atomicOr(&touched[w], m);
        }
    }
    const unsigned long long b = __ballot(inb);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&scal[US_INB], (unsigned long long)__popcll(b));
}

// map_coor2occ's dilation for the NEWLY occupied cells (the others' neighbourhoods are set already)
__global__ void upd_dilate_kernel(GridView g, const uint4 *__restrict__ occ_old, const unsigned long long *__restrict__ bits, uint32_t n_words, int qx, int qy, int qz,
                                  unsigned long long *dil)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint4 r = occ_old[w];
    unsigned long long bb = bits[w] & ~((unsigned long long)r.x | ((unsigned long long)r.y << 32));
    if (!bb) return;
    const int wz = w % g.bz, wy = (w / g.bz) % g.by, wx = w / (g.bz * g.by);
    while (bb) {
        const int b = __ffsll((long long)bb) - 1;
        bb &= bb - 1;
        const int x = wx * 4 + (b >> 4), y = wy * 4 + ((b >> 2) & 3), z = wz * 4 + (b & 3);
        const int x0 = max(0, x - qx / 2), x1 = min(g.dx, x + (qx + 1) / 2), y0 = max(0, y - qy / 2), y1 = min(g.dy, y + (qy + 1) / 2);
        const int z0 = max(0, z - qz / 2), z1 = min(g.dz, z + (qz + 1) / 2);
        for (int xx = x0; xx < x1; ++xx)
            for (int yy = y0; yy < y1; ++yy)
                for (int zz = z0; zz < z1; ++zz) atomicOr(&dil[brick_word(g, xx, yy, zz)], 1ull << brick_bit(xx, yy, zz));
    }
}

// per occupied cell of the new numbering: its old slot (-1: new cell) and its old unclamped point count
__global__ void upd_cells_kernel(const uint4 *__restrict__ occ_old, const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ prefix, uint32_t n_words,
                                 const uint32_t *__restrict__ total_old, uint32_t occ_cap, uint32_t *total_new, int32_t *old_of_new)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned long long bb = bits[w];
    if (!bb) return;
    const uint4 r = occ_old[w];
    const unsigned long long bo = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
    uint32_t ns = prefix[w];
    while (bb) {
        const int b = __ffsll((long long)bb) - 1;
        bb &= bb - 1;
        if (ns < occ_cap) {
            const bool was = (bo >> b) & 1ull;
            const uint32_t os = r.z + (uint32_t)__popcll(bo & ((1ull << b) - 1ull));
            old_of_new[ns] = was ? (int32_t)os : -1;
            total_new[ns] = was ? total_old[os] : 0u;
        }
        ++ns;
    }
}

// sort keys of the new points: the (new) slot of their cell, or 0xFFFFFFFF outside the grid
__global__ void upd_keys_kernel(const float *__restrict__ xyz, int n_old, int n, GridView g, const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ prefix,
                                uint32_t *keys, uint32_t *vals)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_old + k >= n) return;
    uint32_t slot;
    keys[k] = slot_of_point(xyz, n_old + k, g, bits, prefix, slot) ? slot : 0xFFFFFFFFu;
    vals[k] = (uint32_t)(n_old + k);
}

// rank of a new point among the new points of its cell (id order: the sort is stable), capped at P; the cells' new totals
__global__ void upd_rank_kernel(const uint32_t *__restrict__ keys_sorted, int n_new, int P, uint32_t occ_cap, uint32_t *rank, uint32_t *total_new)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_new) return;
    const uint32_t k = keys_sorted[j];
    if (k == 0xFFFFFFFFu || k >= occ_cap) { rank[j] = 0xFFFFFFFFu; return; }
    uint32_t r = 0;
    while (r <= (uint32_t)P && (int)(j - r) > 0 && keys_sorted[j - r - 1] == k) ++r;
    rank[j] = r;
    atomicAdd(&total_new[k], 1u);
}

// new range table: unchanged lists keep their place, changed ones are sized here (placed by the scan that follows)
__global__ void upd_rng_kernel(const int32_t *__restrict__ old_of_new, const int2 *__restrict__ rng_old, const uint32_t *__restrict__ total_new, uint32_t n_occ, uint32_t occ_cap, int P,
                               int slot0_word, int slot0_bit, const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ prefix, int2 *rng_new, uint32_t *add_len,
                               unsigned long long *scal)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool over = false;
    if (s < n_occ && s < occ_cap) {
        uint32_t slot0 = 0xFFFFFFFFu;
        if (slot0_word >= 0) slot0 = prefix[slot0_word] + (uint32_t)__popcll(bits[slot0_word] & ((1ull << slot0_bit) - 1ull));
        const int32_t os = old_of_new[s];
        const int2 ro = os >= 0 ? rng_old[os] : make_int2(0, 0);
        const uint32_t tot = total_new[s];
        int c = tot > (uint32_t)P ? P : (int)tot;
        if (s == slot0) c = 0;
        over = tot > (uint32_t)P && s != slot0;
        const bool changed = c != ro.y;
        rng_new[s] = make_int2(changed ? -1 : ro.x, c);
        add_len[s] = changed ? (uint32_t)c : 0u;
    }
    const unsigned long long b = __ballot(over);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&scal[US_OVERP], (unsigned long long)__popcll(b));
}

// place the changed lists behind the used part of pts and copy their old entries
__global__ void upd_place_kernel(const int32_t *__restrict__ old_of_new, const int2 *__restrict__ rng_old, const uint32_t *__restrict__ add_len, const uint32_t *__restrict__ add_off,
                                 uint32_t n_occ, uint32_t pts_used, int2 *rng_new, float4 *pts)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_occ || add_len[s] == 0u) return;
    const uint32_t start = pts_used + add_off[s];
    rng_new[s].x = (int)start;
    const int32_t os = old_of_new[s];
    if (os < 0) return;
    const int2 ro = rng_old[os];
    for (int j = 0; j < ro.y; ++j) pts[start + j] = pts[ro.x + j];
}

// the new points behind the old entries of their (changed) cells
__global__ void upd_append_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ vals_sorted, const uint32_t *__restrict__ rank,
                                  int n_new, const int32_t *__restrict__ old_of_new, const int2 *__restrict__ rng_old, const int2 *__restrict__ rng_new,
                                  const uint32_t *__restrict__ add_len, float4 *pts)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_new) return;
    const uint32_t s = keys_sorted[j], r = rank[j];
    if (r == 0xFFFFFFFFu || add_len[s] == 0u) return;
    const int32_t os = old_of_new[s];
    const int oc = os >= 0 ? rng_old[os].y : 0;
    const int2 rn = rng_new[s];
    const uint32_t pos = (uint32_t)oc + r;
    if (pos >= (uint32_t)rn.y) return;
    const uint32_t i = vals_sorted[j];
    pts[(uint32_t)rn.x + pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float((int)i));
}

// cells of the dilated mask whose run must be rebuilt: the 3x3x3 neighbourhood of every cell whose list changed
__global__ void upd_dirty_kernel(GridView g, const unsigned long long *__restrict__ touched, const unsigned long long *__restrict__ bits, const uint32_t *__restrict__ prefix,
                                 uint32_t n_words, const uint32_t *__restrict__ add_len, uint32_t occ_cap, unsigned long long *ddirty)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    unsigned long long tt = touched[w];
    if (!tt) return;
    const unsigned long long bb = bits[w];
    const int wz = w % g.bz, wy = (w / g.bz) % g.by, wx = w / (g.bz * g.by);
    while (tt) {
        const int b = __ffsll((long long)tt) - 1;
        tt &= tt - 1;
        if (!((bb >> b) & 1ull)) continue;                                    // (cannot happen: a touched cell is occupied)
        const uint32_t s = prefix[w] + (uint32_t)__popcll(bb & ((1ull << b) - 1ull));
        if (s >= occ_cap || add_len[s] == 0u) continue;                         // its list did not change (it was full already, or it is the slot-0 cell)
        const int x = wx * 4 + (b >> 4), y = wy * 4 + ((b >> 2) & 3), z = wz * 4 + (b & 3);
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz)
                    if (in_bounds(g, x + dx, y + dy, z + dz)) atomicOr(&ddirty[brick_word(g, x + dx, y + dy, z + dz)], 1ull << brick_bit(x + dx, y + dy, z + dz));
    }
}

// a cell's neighbourhood through the NEW tables
__device__ __forceinline__ bool upd_nb_cell(const GridView &g, const unsigned long long *bits, const uint32_t *prefix, const int2 *rng, int vx, int vy, int vz, int2 &rg)
{
    if (!in_bounds(g, vx, vy, vz)) return false;
    const uint32_t w = brick_word(g, vx, vy, vz);
    const unsigned long long bb = bits[w];
    const int b = brick_bit(vx, vy, vz);
    if (!((bb >> b) & 1ull)) return false;
    rg = rng[prefix[w] + (uint32_t)__popcll(bb & ((1ull << b) - 1ull))];
    return true;
}

// new run table: clean runs keep their place; dirty ones are measured (FILL = false) and then written (FILL = true) -- over their old run when the new
// one fits its two padded parts (a point or two added to a neighbourhood usually do), behind the used part of nb_pts otherwise
template <bool FILL>
__global__ void upd_runs_kernel(GridView g, const unsigned long long *__restrict__ dil_old, const uint4 *__restrict__ drec_old, const unsigned long long *__restrict__ dil,
                                const uint32_t *__restrict__ dprefix, const unsigned long long *__restrict__ ddirty, uint32_t n_words, const unsigned long long *__restrict__ bits,
                                const uint32_t *__restrict__ prefix, const int2 *__restrict__ rng, const float4 *__restrict__ pts, const uint2 *__restrict__ nb_rng_old,
                                uint32_t dil_cap, uint32_t nb_used, const uint32_t *__restrict__ run_off, uint2 *nb_rng, uint32_t *run_len, float4 *nb_pts)
{
    const uint32_t w = (uint32_t)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n_words) return;
    const unsigned long long bb = dil[w];
    const int b = threadIdx.x & 63;
    if (!((bb >> b) & 1ull)) return;
    const uint32_t ds = dprefix[w] + (uint32_t)__popcll(bb & ((1ull << b) - 1ull));
    if (ds >= dil_cap) return;
    const unsigned long long bo = dil_old[w];
    const bool dirty = ((ddirty[w] >> b) & 1ull) || !((bo >> b) & 1ull);
    if (!dirty) {
        if (!FILL) { nb_rng[ds] = nb_rng_old[drec_old[w].z + (uint32_t)__popcll(bo & ((1ull << b) - 1ull))]; run_len[ds] = 0u; }
        return;
    }
    const int wz = (int)(w % (uint32_t)g.bz), wy = (int)((w / (uint32_t)g.bz) % (uint32_t)g.by), wx = (int)(w / ((uint32_t)g.bz * (uint32_t)g.by));
    const int x = wx * 4 + (b >> 4), y = wy * 4 + ((b >> 2) & 3), z = wz * 4 + (b & 3);
    uint32_t c0 = 0, total = 0, cells1 = 0, cell0 = 0;
    // FILL: run_len[ds] > 0: appended at nb_used + run_off[ds]; == 0: in place (the measuring pass left the old start in nb_rng[ds].x)
    // (an empty new run has no old place either: it points at the append position -- readers fetch the first records of a run before they test its length)
    const uint32_t start = FILL ? ((run_len[ds] || nb_rng[ds].x == 0xFFFFFFFFu) ? nb_used + run_off[ds] : nb_rng[ds].x) : 0u;
    uint32_t out = start;
    int2 rg;
    if (upd_nb_cell(g, bits, prefix, rng, x, y, z, rg)) {
        cell0 = 1; c0 = (uint32_t)rg.y; total = c0;
        if (FILL) for (int j = 0; j < rg.y; ++j) nb_pts[out++] = pts[rg.x + j];
    }
    const uint32_t c0p = (c0 + 3u) & ~3u;
    if (FILL) out = start + c0p;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                if (dx == 0 && dy == 0 && dz == 0) continue;
                if (!upd_nb_cell(g, bits, prefix, rng, x + dx, y + dy, z + dz, rg)) continue;
                ++cells1; total += (uint32_t)rg.y;
                if (FILL) for (int j = 0; j < rg.y; ++j) nb_pts[out++] = pts[rg.x + j];
            }
    if (FILL) nb_rng[ds] = make_uint2(start, c0 | (total << 6) | (cells1 << 17) | (cell0 << 22));
    else {
        const uint32_t len = c0p + ((total - c0 + 3u) & ~3u);
        uint32_t keep = 0xFFFFFFFFu;                                           // old start when the new run fits the old one's padded parts
        if ((bo >> b) & 1ull) {
            const uint2 ro = nb_rng_old[drec_old[w].z + (uint32_t)__popcll(bo & ((1ull << b) - 1ull))];
            const uint32_t c0o = ro.y & 63u, toto = (ro.y >> 6) & 2047u;
            if (c0p == ((c0o + 3u) & ~3u) && ((total - c0 + 3u) & ~3u) <= ((toto - c0o + 3u) & ~3u)) keep = ro.x;
        }
        run_len[ds] = keep != 0xFFFFFFFFu ? 0u : len;
        nb_rng[ds] = make_uint2(keep, 0u);
    }
}

// totals the host needs before anything visible is written: cells, dilated cells, appended entries
__global__ void upd_totals_kernel(const uint32_t *cnt, const uint32_t *prefix, const uint32_t *dcnt, const uint32_t *dprefix, uint32_t n_words, const uint32_t *add_len,
                                  const uint32_t *add_off, uint32_t occ_cap, const uint32_t *run_len, const uint32_t *run_off, uint32_t dil_cap, unsigned long long *scal, int phase)
{
    if (phase == 0) {
        scal[US_NOCC] = (unsigned long long)prefix[n_words - 1] + cnt[n_words - 1];
        scal[US_NDIL] = (unsigned long long)dprefix[n_words - 1] + dcnt[n_words - 1];
    } else if (phase == 1) {
        const uint32_t n = (uint32_t)scal[US_NOCC];
        scal[US_PTS_ADD] = (n == 0 || n > occ_cap) ? 0ull : (unsigned long long)add_off[n - 1] + add_len[n - 1];
    } else {
        const uint32_t n = (uint32_t)scal[US_NDIL];
        scal[US_NB_ADD] = (n == 0 || n > dil_cap) ? 0ull : (unsigned long long)run_off[n - 1] + run_len[n - 1];
    }
}

#define GU_CHECK(e)                                                                       \
    do {                                                                                  \
        hipError_t _e = (e);                                                              \
        if (_e != hipSuccess) {                                                           \
            set_error("hnr_grid_grow: %s -> %s", #e, hipGetErrorString(_e));              \
            return (_e == hipErrorOutOfMemory) ? HNR_ERR_NOMEM : HNR_ERR_HIP;             \
        }                                                                                 \
    } while (0)

template <class T> static hipError_t upd_alloc(T *&ptr, size_t n) { return ptr ? hipSuccess : hipMalloc((void **)&ptr, (n ? n : 1) * sizeof(T)); }

static int grow_impl(hnr_grid *g, const float *d_xyz, int n, hipStream_t st)
{
    const int TB = 256;
    const uint32_t n_words = g->n_words;
    const int n_old = (int)g->st.n_points, n_new = n - n_old, P = g->p.P;
    if (!g->upd) g->upd = new hnr_grid_upd();
    hnr_grid_upd &u = *g->upd;
    if (!u.arena) {
        // the table-sized scratch arrays from ONE allocation (two dozen hipMallocs cost 0.3 ms on the first call); the six SPARE tables that are swapped
        // with the grid's own (and freed through it) stay individual allocations
        const size_t nw = n_words, oc = g->occ_cap, dc = g->dil_cap;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t total = 3 * up(nw * 8) + 4 * up(nw * 4) + 3 * up(oc * 4) + 2 * up(dc * 4) + up(US_N * 8);
        GU_CHECK(hipMalloc((void **)&u.arena, total));
        char *base = u.arena;
        auto take = [&](size_t b) { char *q = base; base += up(b); return q; };
        u.bits = (unsigned long long *)take(nw * 8); u.touched = (unsigned long long *)take(nw * 8); u.ddirty = (unsigned long long *)take(nw * 8);
        u.cnt = (uint32_t *)take(nw * 4); u.prefix = (uint32_t *)take(nw * 4); u.dcnt = (uint32_t *)take(nw * 4); u.dprefix = (uint32_t *)take(nw * 4);
        u.add_len = (uint32_t *)take(oc * 4); u.add_off = (uint32_t *)take(oc * 4); u.old_of_new = (int32_t *)take(oc * 4);
        u.run_len = (uint32_t *)take(dc * 4); u.run_off = (uint32_t *)take(dc * 4);
        u.scal = (unsigned long long *)take(US_N * 8);
    }
    // (no-ops once allocated; a call that failed half-way leaves the rest for the next one)
    GU_CHECK(upd_alloc(u.dil, n_words)); GU_CHECK(upd_alloc(u.occ_rec, n_words)); GU_CHECK(upd_alloc(u.dil_rec, n_words));
    GU_CHECK(upd_alloc(u.cell_rng, g->occ_cap)); GU_CHECK(upd_alloc(u.cell_total, g->occ_cap)); GU_CHECK(upd_alloc(u.nb_rng, g->dil_cap));
    if (n_new > u.new_cap) {
        for (uint32_t **q : {&u.keys, &u.vals, &u.keys2, &u.vals2, &u.rank}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        u.new_cap = 0;
        const int cap = n_new + n_new / 2 + 1024;
        GU_CHECK(upd_alloc(u.keys, cap)); GU_CHECK(upd_alloc(u.vals, cap)); GU_CHECK(upd_alloc(u.keys2, cap)); GU_CHECK(upd_alloc(u.vals2, cap));
        GU_CHECK(upd_alloc(u.rank, cap));
        u.new_cap = cap;
    }
    size_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    GU_CHECK(rocprim::exclusive_scan(nullptr, s1, u.cnt, u.prefix, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st));
    GU_CHECK(rocprim::exclusive_scan(nullptr, s2, u.add_len, u.add_off, 0u, (size_t)g->occ_cap, rocprim::plus<uint32_t>(), st));
    GU_CHECK(rocprim::exclusive_scan(nullptr, s3, u.run_len, u.run_off, 0u, (size_t)g->dil_cap, rocprim::plus<uint32_t>(), st));
    GU_CHECK(rocprim::radix_sort_pairs(nullptr, s4, u.keys, u.keys2, u.vals, u.vals2, (size_t)u.new_cap, 0, 32, st));
    size_t need = s1 > s2 ? s1 : s2; if (s3 > need) need = s3; if (s4 > need) need = s4;
    if (need > u.tmp_bytes) { if (u.tmp) (void)hipFree(u.tmp); u.tmp = nullptr; GU_CHECK(hipMalloc((void **)&u.tmp, need)); u.tmp_bytes = need; }

    const GridView v = g->view();
    GU_CHECK(hipMemsetAsync(u.scal, 0, US_N * 8, st));
    // ---- masks, prefixes
    upd_init_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(g->occ_rec, g->dil, n_words, u.bits, u.dil, u.touched, u.ddirty);
    upd_mark_kernel<<<cdiv(n_new, TB), TB, 0, st>>>(d_xyz, n_old, n, v, u.bits, u.touched, u.scal);
    upd_dilate_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(v, g->occ_rec, u.bits, n_words, g->p.query_size[0], g->p.query_size[1], g->p.query_size[2], u.dil);
    popc_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(u.bits, n_words, u.cnt);
    popc_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(u.dil, n_words, u.dcnt);
    { size_t sz = u.tmp_bytes; GU_CHECK(rocprim::exclusive_scan((void *)u.tmp, sz, u.cnt, u.prefix, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st)); }
    { size_t sz = u.tmp_bytes; GU_CHECK(rocprim::exclusive_scan((void *)u.tmp, sz, u.dcnt, u.dprefix, 0u, (size_t)n_words, rocprim::plus<uint32_t>(), st)); }
    upd_totals_kernel<<<1, 1, 0, st>>>(u.cnt, u.prefix, u.dcnt, u.dprefix, n_words, nullptr, nullptr, g->occ_cap, nullptr, nullptr, g->dil_cap, u.scal, 0);
    // ---- cells: old slots, totals, the new points sorted by cell
    GU_CHECK(hipMemsetAsync(u.add_len, 0, (size_t)g->occ_cap * 4, st));
    GU_CHECK(hipMemsetAsync(u.run_len, 0, (size_t)g->dil_cap * 4, st));
    upd_cells_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(g->occ_rec, u.bits, u.prefix, n_words, g->cell_total, g->occ_cap, u.cell_total, u.old_of_new);
    upd_keys_kernel<<<cdiv(n_new, TB), TB, 0, st>>>(d_xyz, n_old, n, v, u.bits, u.prefix, u.keys, u.vals);
    { size_t sz = u.tmp_bytes; GU_CHECK(rocprim::radix_sort_pairs((void *)u.tmp, sz, u.keys, u.keys2, u.vals, u.vals2, (size_t)n_new, 0, 32, st)); }
    upd_rank_kernel<<<cdiv(n_new, TB), TB, 0, st>>>(u.keys2, n_new, P, g->occ_cap, u.rank, u.cell_total);
    GU_CHECK(hipGetLastError());
    // the host needs the number of cells to size the per-cell launches (and to bail out before anything visible changes)
    unsigned long long h[US_N];
    GU_CHECK(hipMemcpyAsync(h, u.scal, sizeof(h), hipMemcpyDeviceToHost, st));
    GU_CHECK(hipStreamSynchronize(st));
    const uint64_t n_occ = h[US_NOCC], n_dil = h[US_NDIL];
    if (n_occ > g->occ_cap || n_dil > g->dil_cap || n_occ > (uint64_t)g->p.max_o) {
        set_error("hnr_grid_grow: rebuild needed (%llu cells / capacity %u, %llu dilated cells / capacity %u, max_o %d)", (unsigned long long)n_occ, g->occ_cap,
                  (unsigned long long)n_dil, g->dil_cap, g->p.max_o);
        return HNR_NEED_REBUILD;
    }
    upd_rng_kernel<<<cdiv(n_occ, TB), TB, 0, st>>>(u.old_of_new, g->cell_rng, u.cell_total, (uint32_t)n_occ, g->occ_cap, P, g->slot0_word, g->slot0_bit, u.bits, u.prefix,
                                                  u.cell_rng, u.add_len, u.scal);
    { size_t sz = u.tmp_bytes; GU_CHECK(rocprim::exclusive_scan((void *)u.tmp, sz, u.add_len, u.add_off, 0u, (size_t)n_occ, rocprim::plus<uint32_t>(), st)); }
    upd_totals_kernel<<<1, 1, 0, st>>>(u.cnt, u.prefix, u.dcnt, u.dprefix, n_words, u.add_len, u.add_off, g->occ_cap, nullptr, nullptr, g->dil_cap, u.scal, 1);
    upd_dirty_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(v, u.touched, u.bits, u.prefix, n_words, u.add_len, g->occ_cap, u.ddirty);
    const int run_blocks = cdiv((int64_t)n_words * 64, 256);
    upd_runs_kernel<false><<<run_blocks, 256, 0, st>>>(v, g->dil, g->dil_rec, u.dil, u.dprefix, u.ddirty, n_words, u.bits, u.prefix, u.cell_rng, g->pts, g->nb_rng, g->dil_cap,
                                                      g->nb_used, nullptr, u.nb_rng, u.run_len, nullptr);
    { size_t sz = u.tmp_bytes; GU_CHECK(rocprim::exclusive_scan((void *)u.tmp, sz, u.run_len, u.run_off, 0u, (size_t)n_dil, rocprim::plus<uint32_t>(), st)); }
    upd_totals_kernel<<<1, 1, 0, st>>>(u.cnt, u.prefix, u.dcnt, u.dprefix, n_words, u.add_len, u.add_off, g->occ_cap, u.run_len, u.run_off, g->dil_cap, u.scal, 2);
    GU_CHECK(hipGetLastError());
    GU_CHECK(hipMemcpyAsync(h, u.scal, sizeof(h), hipMemcpyDeviceToHost, st));
    GU_CHECK(hipStreamSynchronize(st));
    const uint64_t pts_add = h[US_PTS_ADD], nb_add = h[US_NB_ADD];
    if ((uint64_t)g->pts_used + pts_add > g->pts_cap || (uint64_t)g->nb_used + nb_add + 4 > g->nb_cap) {
        set_error("hnr_grid_grow: rebuild needed (list entries %u + %llu of %u, run entries %u + %llu of %u: the slack of the build is used up)", g->pts_used,
                  (unsigned long long)pts_add, g->pts_cap, g->nb_used, (unsigned long long)nb_add, g->nb_cap);
        return HNR_NEED_REBUILD;
    }
    // ---- from here on the live arrays are written: the appended parts of pts / nb_pts first (no reader sees them yet), then the small tables are swapped in
    upd_place_kernel<<<cdiv(n_occ, TB), TB, 0, st>>>(u.old_of_new, g->cell_rng, u.add_len, u.add_off, (uint32_t)n_occ, g->pts_used, u.cell_rng, g->pts);
    upd_append_kernel<<<cdiv(n_new, TB), TB, 0, st>>>(d_xyz, u.keys2, u.vals2, u.rank, n_new, u.old_of_new, g->cell_rng, u.cell_rng, u.add_len, g->pts);
    upd_runs_kernel<true><<<run_blocks, 256, 0, st>>>(v, g->dil, g->dil_rec, u.dil, u.dprefix, u.ddirty, n_words, u.bits, u.prefix, u.cell_rng, g->pts, g->nb_rng, g->dil_cap,
                                                     g->nb_used, u.run_off, u.nb_rng, u.run_len, g->nb_pts);
    GU_CHECK(hipMemsetAsync(g->nb_pts + g->nb_used + nb_add, 0, 4 * sizeof(float4), st));
    pack_rec_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(u.bits, u.prefix, n_words, u.occ_rec);
    pack_dil_rec_kernel<<<cdiv(n_words, TB), TB, 0, st>>>(u.dil, u.dprefix, n_words, u.dil_rec);
    GU_CHECK(hipGetLastError());
    // (kernels already queued on `st` keep reading the old tables: the swap below only changes what LATER launches are handed; the old arrays become the
    //  next update's spares and are not written before that update's kernels, which run behind everything queued now)
    std::swap(g->occ_rec, u.occ_rec); std::swap(g->dil_rec, u.dil_rec); std::swap(g->dil, u.dil); std::swap(g->cell_rng, u.cell_rng);
    std::swap(g->cell_total, u.cell_total); std::swap(g->nb_rng, u.nb_rng);
    g->n_occ = (uint32_t)n_occ; g->n_dil = (uint32_t)n_dil; g->pts_used += (uint32_t)pts_add; g->nb_used += (uint32_t)nb_add;
    g->st.n_points = n; g->st.n_inbounds += (int64_t)h[US_INB]; g->st.n_occ = (int64_t)n_occ; g->st.n_cells_over_P = (int64_t)h[US_OVERP];
    g->st.n_dilated = (int64_t)n_dil;
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
    if (g->cell_total) (void)hipFree(g->cell_total);
    grid_upd_free(g->upd);
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

extern "C" int hnr_grid_grow(hnr_grid *g, const float *d_xyz, int n_points, void *stream)
{
    if (!g || !d_xyz || n_points <= 0) { set_error("hnr_grid_grow: bad argument"); return HNR_ERR_BADARG; }
    if ((int64_t)n_points < g->st.n_points) { set_error("hnr_grid_grow: %d points, the grid was built from %lld (points can only be appended)", n_points, (long long)g->st.n_points); return HNR_ERR_BADARG; }
    if ((int64_t)n_points == g->st.n_points) return HNR_OK;
    // what the incremental form does not cover: no neighbourhood lists (P > 63), the two-level march's brick mask, a cloud that had no in-bounds point
    // (the first in-bounds point decides the slot-0 cell), voxels already dropped for max_o (the kept set is defined by first appearance)
    if (!g->nb_pts || !g->cell_total || g->brick_near || g->first_inb == 0x7fffffff || g->slot0_word < 0 || g->st.n_dropped_voxels > 0) return HNR_NEED_REBUILD;
    return grow_impl(g, d_xyz, n_points, (hipStream_t)stream);
}

// Test hook: the 3x3x3 neighbourhood run of every cell of the dilated mask in the dense layout: d_run_len [X*Y*Z] i32 (-1: cell not in the mask, else the
// candidates of the run) and d_run_hash [X*Y*Z] u64 (an order-dependent hash of the run's (point id, x, y, z) records and of its packed counts).
__global__ void export_runs_kernel(GridView g, int32_t *run_len, unsigned long long *run_hash)
{
    const int64_t vol = (int64_t)g.dx * g.dy * g.dz;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vol) return;
    const int z = (int)(i % g.dz), y = (int)((i / g.dz) % g.dy), x = (int)(i / ((int64_t)g.dz * g.dy));
    const uint4 rec = g.dil_rec[brick_word(g, x, y, z)];
    const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
    const int b = brick_bit(x, y, z);
    if (!((bb >> b) & 1ull)) { run_len[i] = -1; run_hash[i] = 0ull; return; }
    const uint2 rg = g.nb_rng[rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull))];
    const uint32_t c0 = rg.y & 63u, total = (rg.y >> 6) & 2047u, c0p = (c0 + 3u) & ~3u;
    unsigned long long hsh = 1469598103934665603ull ^ rg.y;
    for (uint32_t k = 0; k < total; ++k) {
        const float4 q = g.nb_pts[rg.x + (k < c0 ? k : c0p + (k - c0))];
        const unsigned long long w0 = ((unsigned long long)__float_as_uint(q.w) << 32) | __float_as_uint(q.x), w1 = ((unsigned long long)__float_as_uint(q.y) << 32) | __float_as_uint(q.z);
        hsh = (hsh ^ w0) * 1099511628211ull; hsh = (hsh ^ w1) * 1099511628211ull;
    }
    run_len[i] = (int32_t)total; run_hash[i] = hsh;
}

extern "C" int hnr_grid_export_runs(const hnr_grid *g, int32_t *d_run_len, uint64_t *d_run_hash, void *stream)
{
    if (!g || !d_run_len || !d_run_hash || !g->nb_pts) { set_error("hnr_grid_export_runs: bad argument / the grid has no neighbourhood lists"); return HNR_ERR_BADARG; }
    const int64_t vol = (int64_t)g->p.dims[0] * g->p.dims[1] * g->p.dims[2];
    export_runs_kernel<<<cdiv(vol, 256), 256, 0, (hipStream_t)stream>>>(g->view(), d_run_len, reinterpret_cast<unsigned long long *>(d_run_hash));
    HNR_LAUNCH_CHECK();
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
