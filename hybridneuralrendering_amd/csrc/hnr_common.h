// Shared device/host helpers for libhnr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hnr.h"

namespace hnr {

void set_error(const char *fmt, ...);

#define HNR_HIP_CHECK(expr)                                                             \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::hnr::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return HNR_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define HNR_LAUNCH_CHECK()  HNR_HIP_CHECK(hipGetLastError())

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Device view of the voxel grid (plain struct, passed by value to kernels).
//
// Cells are grouped in 4x4x4 bricks; one 64-bit word per brick, bit = (x&3)<<4 | (y&3)<<2 | (z&3).
//   occ_rec[w] = {bits_lo, bits_hi, prefix, 0}: which cells of brick w hold points, and how many
//                occupied cells precede brick w, so slot(cell) = prefix + popc(bits below the cell's bit);
//   dil[w]     = dilated occupancy (the reference's coor_occ) for the ray march: one bit per cell;
//   cell_rng[slot] = {first index into pts, min(P, points in the cell)};
//   pts[i]     = {x, y, z, bit-cast point id}, sorted by slot, point-id order inside a cell.
struct GridView {
    float ox, oy, oz;      // origin (d_coord_shift)
    float cx, cy, cz;      // cell size (d_voxel_size)
    int   dx, dy, dz;      // dims in cells
    int   by, bz;          // brick dims along y, z
    const uint4 *occ_rec;
    const unsigned long long *dil;
    const int2 *cell_rng;
    const float4 *pts;
    // 3x3x3 NEIGHBOURHOOD LISTS (round 5; built when P <= 63, NULL otherwise): for every cell of the dilated mask -- every cell a kept shading sample
    // can sit in -- the candidates of its 27-cell neighbourhood as ONE contiguous run in the reference's enumeration order (own cell first = shell 0,
    // then the other cells x-major / y / z = shell 1; query_point_indices_worldcoords.py:478-491), so the k-NN makes 2 dependent lookups per sample
    // instead of 27 x 2 and streams its candidates from consecutive addresses.  16 B x 27 x listed points: 0.86 GB at 2 M points -- HBM is what this GPU has.
    //   dil_rec[w]   = {dilated bits lo, hi, number of dilated cells before brick w, 0}
    //   run layout  = [own cell: c0 entries, padded to a multiple of 4][the other cells: padded to a multiple of 4]: both parts start on a 64-byte line
    //   nb_rng[slot] = {first index into nb_pts, own-cell candidates (6 bits) | all candidates << 6 (11 bits) | occupied shell-1 cells << 17 (5 bits) | own cell occupied << 22}
    const uint4 *dil_rec;
    const uint2 *nb_rng;
    const float4 *nb_pts;
    // brick_near[w] != 0: some cell of the dilated mask lies in brick w or within TWO bricks of it (5x5x5 bricks).  The march probes one depth in four
    // and skips the other three when the probed one sits in a brick with brick_near == 0 and the ray cannot leave the 5x5x5 block within three steps.
    const uint8_t *brick_near;
};

// floor((p - shift) / size) with fp32 subtract and IEEE fp32 divide, exactly as the reference
// kernels write it (query_point_indices_worldcoords.py:259-261, :400-402, :465-467).
// Returns INT_MIN for values that do not fit an int (C leaves that cast undefined).
// Host-side per-DEVICE caches (hipFuncSetAttribute and the CU count belong to the current device: a process that renders on a second GPU must
// not reuse the first one's).  Not thread-safe beyond "the same value is written twice".
inline int device_num_cus()
{
    static int n_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        hipDeviceProp_t prop;
        n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}
struct PerDeviceOnce {          // `static PerDeviceOnce once; if (once.first()) { ... set the kernel attributes ... }`
    bool done[64] = {};
    bool first()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// Publishes a kernel's running max |value| (mx >= 0) into a maximum word that per-tensor power-of-two scales are derived from (row_scale_exp
// reads the EXPONENT only): atomicMax on the bit pattern, skipped when the stored value already has the same or a larger exponent.  The
// stored word therefore has the exponent of the true maximum, not necessarily its mantissa (hnr_absmax itself stores the exact maximum).
// Same-address atomics retire one per ~23 ns: thousands of waves each publishing a slightly larger value cost more than the kernel.
__device__ __forceinline__ void absmax_publish(unsigned *dst, float mx)
{
    const unsigned b = __float_as_uint(mx);
    if (b == 0u) return;
    // (device-scope relaxed load: served by L2.  The filter only helps publishers that arrive one after another: waves that all finish
    // at the same moment all read the old value -- publish once per workgroup, from kernels whose workgroups end at different times)
    if ((b >> 23) > (__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 23)) atomicMax(dst, b);
}

// Correctly rounded fp32 division WITHOUT v_div_scale / v_div_fmas: the steps of the compiler's own expansion (reciprocal, one refinement of it,
// quotient, two residual corrections -- the same fused operations in the same order, hence the same bits) minus the operand scaling that only
// matters at the ends of the exponent range; operands out there take the compiler's `/`.  History: written when wrong quotients in lanes 48..63 under
// GPU contention were blamed on the lane mask v_div_fmas reads from VCC; the cause turned out to be packed fp32 arithmetic (hnr_h2.h).
// Kept: bit-identical to `/`, cheaper for loop-invariant divisors, and no state handed from one instruction to another in VCC.
__device__ __forceinline__ float hnr_div(float n, float d)
{
    // fast path: d normal with exponent in [-95, 95], n zero or normal with an exponent within 120 of d's (no overflow / underflow of the quotient or
    // the residuals).  Everything that depends on d alone comes first: with a loop-invariant divisor (cell sizes) the compiler hoists it.
    const int ed = (int)((__float_as_uint(d) >> 23) & 0xffu), en = (int)((__float_as_uint(n) >> 23) & 0xffu);
    const int lo = ed - 120 > 32 ? ed - 120 : 32, hi = ed + 120 < 222 ? ed + 120 : 222;
    const bool d_ok = (unsigned)(ed - 32) <= 190u;
    if (__builtin_expect(!(d_ok && ((unsigned)(en - lo) <= (unsigned)(hi - lo) || n == 0.f)), 0))
#ifdef HNR_DIV_NO_FALLBACK                                                  // tools/check_divisions.sh: with the fallback gone, no object may contain v_div_fmas
        return 0.f;
#else
        return n / d;                                                       // denormal / huge / inf / nan operands, quotients near the range's ends
#endif
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    const float q0 = __fmul_rn(n, r);
    const float q = fmaf(fmaf(-d, q0, n), r, q0);
    return n == 0.f ? q0 : fmaf(fmaf(-d, q, n), r, q);                      // (a zero keeps the sign of n x r: the corrections would turn -0 / d into +0)
}

// The same for fp64 (the loss kernels' scalar means): reciprocal refined twice, quotient, one residual correction -- the compiler's own steps.
__device__ __forceinline__ double hnr_div64(double n, double d)
{
    const unsigned en = (unsigned)((__double_as_longlong(n) >> 52) & 0x7ff), ed = (unsigned)((__double_as_longlong(d) >> 52) & 0x7ff);
    if (__builtin_expect(ed - 128u > 1790u || (en - 128u > 1790u && n != 0.0) || (int)en - (int)ed > 900 || (int)en - (int)ed < -900, 0))
#ifdef HNR_DIV_NO_FALLBACK
        return 0.0;
#else
        return n / d;
#endif
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// hnr_div for the cell index floor((p - shift) / size): only the divisor is checked (the cell size: uniform, normal, exponent in [-95, 95]), so the
// quotient costs five VALU instructions.  For a numerator whose exponent is more than 120 away from the divisor's the corrections may round
// differently from IEEE division, but such a quotient is either below 2^-24 in magnitude (its floor is decided by its sign, which n x r carries) or
// beyond the +-2e9 range test that follows (INT_MIN either way); inf / nan numerators give nan (INT_MIN, like inf / size).
__device__ __forceinline__ float hnr_div_cell(float n, float d)
{
    const int ed = (int)((__float_as_uint(d) >> 23) & 0xffu);
    if (__builtin_expect((unsigned)(ed - 32) > 190u, 0))
#ifdef HNR_DIV_NO_FALLBACK
        return 0.f;
#else
        return n / d;
#endif
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    const float q0 = __fmul_rn(n, r);
    const float q = fmaf(fmaf(-d, q0, n), r, q0);
    return fmaf(fmaf(-d, q, n), r, q);
}

// An integer the compiler knows nothing about (no range, no relation to other values): keeps index arithmetic on small values in 32-bit instructions.
__device__ __forceinline__ int hnr_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ int cell_coord(float p, float o, float c)
{
    float d = __fsub_rn(p, o);
    float q = hnr_div_cell(d, c);
    if (!(q > -2.0e9f && q < 2.0e9f)) return INT32_MIN;
    return (int)floorf(q);
}

__device__ __forceinline__ bool in_bounds(const GridView &g, int x, int y, int z)
{
    return x >= 0 && x < g.dx && y >= 0 && y < g.dy && z >= 0 && z < g.dz;
}

__device__ __forceinline__ uint32_t brick_word(const GridView &g, int x, int y, int z)
{
    return (uint32_t)(((x >> 2) * g.by + (y >> 2)) * g.bz + (z >> 2));
}

__device__ __forceinline__ int brick_bit(int x, int y, int z)
{
    return ((x & 3) << 4) | ((y & 3) << 2) | (z & 3);
}

// Reprojection of a world-space sample into reference view v + truncation to a pixel + bounds rule (models/neural_points_volumetric_model.py:248-255,
// models/aggregators/point_aggregators.py:1077-1088): w2c row-major 4x4, Kmat row-major 3x3, every fp32 operation rounded separately in this order
// (-ffp-contract=off), the divisions correctly rounded.  ONE text for every kernel that gathers reference-view pixels (merge stage fused / un-fused,
// its backward, the hnr_proj_pixels probe): a sample within an ulp of a pixel border must land on the same pixel everywhere.  Returns `invalid`
// (then px = py = 0, the zeroed pixel).
__device__ __forceinline__ bool hnr_project_pixel(float x, float y, float z, const float *m, const float *Kmat, int W, int H, int &px, int &py)
{
    float c[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) c[q] = x * m[4 * q] + y * m[4 * q + 1] + z * m[4 * q + 2] + m[4 * q + 3];
    float i3[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) i3[q] = c[0] * Kmat[3 * q] + c[1] * Kmat[3 * q + 1] + c[2] * Kmat[3 * q + 2];
    const float den = i3[2] + 1e-10f;
    const float fx = hnr_div(i3[0], den), fy = hnr_div(i3[1], den);          // (correctly rounded; not the v_div_fmas expansion: see hnr_div)
    // .to(torch.int32): truncation toward zero; out-of-range / NaN -> invalid
    px = (fx > -2.0e9f && fx < 2.0e9f) ? (int)fx : -1;
    py = (fy > -2.0e9f && fy < 2.0e9f) ? (int)fy : -1;
    const bool inval = px < 0 || px >= W || py < 0 || py >= H;
    if (inval) { px = 0; py = 0; }
    return inval;
}

}  // namespace hnr

struct hnr_grid_upd;             // scratch + spare tables of hnr_grid_grow (csrc/grid.hip), allocated by its first call
struct hnr_grid {
    hnr_grid_params p;
    hnr_grid_stats st;
    int bd[3];
    uint32_t n_words;
    uint4 *occ_rec;
    unsigned long long *dil;
    int2 *cell_rng;
    float4 *pts;
    uint4 *dil_rec;
    uint2 *nb_rng;
    float4 *nb_pts;
    uint8_t *brick_near;
    // what hnr_grid_grow needs to extend the tables in place: entries in use / allocated (the build leaves slack behind pts, nb_pts, cell_rng, nb_rng),
    // the unclamped number of points of every occupied cell, and the cell of the first in-bounds point (the `voxel_idx > 0` rule: it never lists points)
    uint32_t n_occ, n_dil, pts_used, pts_cap, nb_used, nb_cap, occ_cap, dil_cap;
    uint32_t *cell_total;
    int first_inb;               // id of the first in-bounds point, INT_MAX: none
    int slot0_word, slot0_bit;   // its cell (brick word, bit); -1: none
    hnr_grid_upd *upd;
    hnr::GridView view() const
    {
        hnr::GridView v;
        v.ox = p.origin[0]; v.oy = p.origin[1]; v.oz = p.origin[2];
        v.cx = p.cell[0]; v.cy = p.cell[1]; v.cz = p.cell[2];
        v.dx = p.dims[0]; v.dy = p.dims[1]; v.dz = p.dims[2];
        v.by = bd[1]; v.bz = bd[2];
        v.occ_rec = occ_rec; v.dil = dil; v.cell_rng = cell_rng; v.pts = pts;
        v.dil_rec = dil_rec; v.nb_rng = nb_rng; v.nb_pts = nb_pts; v.brick_near = brick_near;
        return v;
    }
};
