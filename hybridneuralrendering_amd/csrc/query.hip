// Ray march + first-SR compaction + layered k-NN over the brick/CSR voxel grid.
//
// Replaces, for one launch of R rays (reference: models/neural_points/query_point_indices_worldcoords.py):
//   near_far_linear_ray_generation's raypos tensor (models/rendering/diff_ray_marching.py:386) -- never
//     materialised: each lane recomputes campos + raydir * t_mid[d] (fp32 mul, then fp32 add);
//   mask_raypos (:384-408)            -> one bit test per marched sample in `dil`;
//   torch max/sum/masked_select/cumsum (:645-655) + get_shadingloc (:411-433)
//                                     -> wavefront ballot + popcount prefix, one wave per ray, no host sync;
//   query_neigh_along_ray_layered (:436-522) -> knn kernel over a device-side work list of kept samples.
//
// Results are bit-identical to oracle/query_oracle.c (same visiting order, same strict-< replacement
// and first-max scan), so sample_pidx matches slot for slot.
#include <stdlib.h>

#include <type_traits>

#include "hnr_common.h"

namespace hnr {

// ------------------------------------------------------------------------------------------------
// March: a wavefront per ray, 64 lanes test 64 consecutive depths per block; the ballot of occupied samples gives every hit its output slot; the
// first SR hits are kept.  Round 5: what bounded this kernel was neither its instructions (halving them changed nothing) nor its bytes but the life of
// 285 200 short waves -- launch, one round trip for the ray, one per block of depths for the table and one for the mask words, 4 - 7 blocks in a row,
// 24 - 32 waves per CU (0.21 ms per frame; 0.16 ms with every block skipped).  Now: PERSISTENT waves loop over rays; the depth table is loaded once per
// wave when all rays share it (no jitter); the next ray's direction is requested while the current ray is worked on; all blocks of a ray are probed
// together (all mask-word requests in flight at once: one round trip per ray); blocks that lie outside the grid's box are skipped by a slab test.
__global__ __launch_bounds__(256) void march_kernel(GridView g, const float *__restrict__ campos,
                                                    const float *__restrict__ raydir,
                                                    const float *__restrict__ tmid, int R, int D, int SR, int K,
                                                    int tmid_stride, int pad, int32_t *__restrict__ pidx,
                                                    float *__restrict__ loc, int32_t *__restrict__ ray_nsamp,
                                                    int8_t *__restrict__ ray_mask, int probe_mode)
{
    const int lane = threadIdx.x & 63;
    const int wave0 = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * (unsigned)blockDim.x) >> 6);
    if (wave0 >= R) return;
    const float px = campos[0], py = campos[1], pz = campos[2];
    const float fdx = (float)g.dx, fdy = (float)g.dy, fdz = (float)g.dz;
    // the grid's box widened by ONE CELL on every side: the widening covers the rounding of the slab test against the exact per-sample arithmetic by
    // orders of magnitude (a cell is ~1e4 ulps of a coordinate)
    const float blo[3] = {g.ox - g.cx, g.oy - g.cy, g.oz - g.cz};
    const float bhi[3] = {g.ox + (float)(g.dx + 1) * g.cx, g.oy + (float)(g.dy + 1) * g.cy, g.oz + (float)(g.dz + 1) * g.cz};
    const int nblk = (D + 63) >> 6;
    const bool fast = D <= 512 && !(probe_mode & 2);           // HNR_MARCH_PROBE=2 (tools): block after block with an early exit instead (fewer instructions, 3 % slower)
    float tv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int d = 64 * k + lane; tv[k] = (fast && tmid_stride == 0 && k < nblk && d < D) ? tmid[d] : NAN; }
    float ndx = raydir[3 * (size_t)wave0], ndy = raydir[3 * (size_t)wave0 + 1], ndz = raydir[3 * (size_t)wave0 + 2];
    for (int r = wave0; r < R; r += n_waves) {
        const float dx = ndx, dy = ndy, dz = ndz;
        {
            const int rn = r + n_waves < R ? r + n_waves : r;    // the next ray's direction: in flight across this ray's work
            ndx = raydir[3 * (size_t)rn]; ndy = raydir[3 * (size_t)rn + 1]; ndz = raydir[3 * (size_t)rn + 2];
        }
        const float *tt = tmid + (size_t)r * tmid_stride;
        float *loc_r = loc + (size_t)r * SR * 3;
        if (fast && tmid_stride != 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int d = 64 * k + lane; tv[k] = (k < nblk && d < D) ? tt[d] : NAN; }
        }
        // the part of the ray that can be inside the grid at all
        float t_in = -INFINITY, t_out = INFINITY;
        {
            const float pp[3] = {px, py, pz}, dd[3] = {dx, dy, dz};
            bool ok = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (dd[a] != 0.f) {
                    const float inv = __builtin_amdgcn_rcpf(dd[a]);
                    const float ta = (blo[a] - pp[a]) * inv, tb = (bhi[a] - pp[a]) * inv;
                    ok = ok && ta == ta && tb == tb;
                    const float t0 = fminf(ta, tb), t1 = fmaxf(ta, tb);
                    // (rcp is an approximation, ~1 ulp: relative slack on top of the one-cell widening)
                    t_in = fmaxf(t_in, t0 - 1e-5f * fabsf(t0)); t_out = fminf(t_out, t1 + 1e-5f * fabsf(t1));
                } else if (!(pp[a] >= blo[a] && pp[a] <= bhi[a])) {
                    if (pp[a] == pp[a]) { t_in = INFINITY; t_out = -INFINITY; } else ok = false;
                }
            }
            if (!ok || !(dx == dx && dy == dy && dz == dz)) { t_in = -INFINITY; t_out = INFINITY; }
        }
        if (probe_mode == 1) { t_in = INFINITY; t_out = -INFINITY; }    // tools: every block skipped (the kernel's fixed part alone; results are empty)
        // one lane, one depth: position (fp32 multiply, then add, like the reference), cell, and the REQUEST for the cell's word of the dilated mask
        auto probe = [&](float t, float &sx, float &sy, float &sz, unsigned long long &word, int &bit) -> bool {
            sx = __fadd_rn(px, __fmul_rn(dx, t));
            sy = __fadd_rn(py, __fmul_rn(dy, t));
            sz = __fadd_rn(pz, __fmul_rn(dz, t));
            // cell = floor((s - o) / c) is inside [0, dims) iff the quotient is in [0, dims) (dims < 2^24: exact as floats; NaN and the out-of-range
            // quotients cell_coord maps to INT_MIN fail the compares), and then truncation IS the floor
            const float qx = hnr_div_cell(__fsub_rn(sx, g.ox), g.cx), qy = hnr_div_cell(__fsub_rn(sy, g.oy), g.cy), qz = hnr_div_cell(__fsub_rn(sz, g.oz), g.cz);
            const bool in = t >= t_in && t <= t_out && qx >= 0.f && qx < fdx && qy >= 0.f && qy < fdy && qz >= 0.f && qz < fdz;
            const int cx = in ? (int)qx : 0, cy = in ? (int)qy : 0, cz = in ? (int)qz : 0;
            word = g.dil[brick_word(g, cx, cy, cz)];             // unconditional (cell 0 for the lanes outside): the load carries no branch
            bit = brick_bit(cx, cy, cz);
            return in;
        };
        int base = 0;
        auto keep = [&](bool occ, float sx, float sy, float sz) {
            const unsigned long long b = __ballot(occ);
            if (occ) {
                const int slot = base + __popcll(b & ((1ull << lane) - 1ull));
                if (slot < SR) { loc_r[3 * slot] = sx; loc_r[3 * slot + 1] = sy; loc_r[3 * slot + 2] = sz; }
            }
            base += __popcll(b);
        };
        if (g.brick_near && !(probe_mode & 4)) {
            // TWO LEVELS (only when the grid was built with HNR_MARCH_TWO_LEVEL=1; HNR_MARCH_PROBE=4 turns it off again).  Measured: same samples, not
            // faster (0.242 vs 0.233 ms) -- kept as a tested option.  The march is bound by its ~65 VALU instructions per probed depth, and most depths it probes lie in the
            // room's air.  Coarse level: one lane per GROUP of four consecutive depths probes the group's first depth exactly; the group is expanded -- its
            // four depths probed like before, in depth order -- unless that first position sits in a brick with nothing of the dilated mask within two bricks
            // (brick_near) AND the ray moves less than two bricks per axis (minus a margin) over the group's three steps, so that none of its positions can
            // reach a cell of the mask.  A first position outside the grid expands whenever the group overlaps the slab interval.  The kept samples are
            // the same by construction (every oracle comparison runs through this path).
            __shared__ int s_grp[4][64];
            int *grp = s_grp[(threadIdx.x >> 6) & 3];
            const float lim_x = 7.99f * g.cx, lim_y = 7.99f * g.cy, lim_z = 7.99f * g.cz;       // two bricks of four cells, minus a margin
            const float adx = fabsf(dx), ady = fabsf(dy), adz = fabsf(dz);
            const int n_groups = (D + 3) >> 2;
            for (int g0 = 0; g0 < n_groups && base < SR; g0 += 64) {
                const int gi = g0 + lane;
                bool expand = false;
                if (gi < n_groups) {
                    const int d0 = 4 * gi, d3 = min(d0 + 3, D - 1);
                    const float ta = tt[d0], tb = tt[d3], t1 = tt[min(d0 + 1, D - 1)], t2 = tt[min(d0 + 2, D - 1)];
                    const bool overlap = !(tb < t_in || ta > t_out);                        // (NaN depths: expand, the fine level sorts them out)
                    const bool mono = ta <= t1 && t1 <= t2 && t2 <= tb;                      // (depth tables ascend; anything else: expand)
                    const float span = mono ? tb - ta : NAN;
                    float sx, sy, sz;
                    unsigned long long word; int bit;
                    // position and cell of the group's first depth, exactly as the fine level computes them
                    sx = __fadd_rn(px, __fmul_rn(dx, ta)); sy = __fadd_rn(py, __fmul_rn(dy, ta)); sz = __fadd_rn(pz, __fmul_rn(dz, ta));
                    const float qx = hnr_div_cell(__fsub_rn(sx, g.ox), g.cx), qy = hnr_div_cell(__fsub_rn(sy, g.oy), g.cy), qz = hnr_div_cell(__fsub_rn(sz, g.oz), g.cz);
                    const bool inb = qx >= 0.f && qx < fdx && qy >= 0.f && qy < fdy && qz >= 0.f && qz < fdz;
                    const bool slow = !(span >= 0.f && adx * span <= lim_x && ady * span <= lim_y && adz * span <= lim_z);    // too far in three steps (or NaN)
                    bool near = true;
                    if (inb && !slow) near = g.brick_near[brick_word(g, (int)qx, (int)qy, (int)qz)] != 0;
                    expand = overlap && (slow || !inb || near);
                    (void)word; (void)bit;
                }
                const unsigned long long em = __ballot(expand);
                if (em == 0ull) continue;
                if (expand) grp[__popcll(em & ((1ull << lane) - 1ull))] = gi;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int n_exp = __popcll(em);
                for (int c0 = 0; c0 < 4 * n_exp && base < SR; c0 += 64) {
                    const int e = c0 + lane;
                    const bool on = e < 4 * n_exp;
                    const int d = on ? 4 * grp[e >> 2] + (e & 3) : D;
                    const float t = d < D ? tt[d] : NAN;
                    float sx, sy, sz;
                    unsigned long long word; int bit;
                    const bool in = probe(t, sx, sy, sz, word, bit);
                    keep(in && ((word >> bit) & 1ull), sx, sy, sz);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                                            // (grp is rewritten by the next block of groups)
            }
        } else if (fast) {
            // (the blocks behind the one that fills the SR-th slot are probed for nothing; their loads are in flight anyway)
            float sx[8], sy[8], sz[8];
            unsigned long long wd[8];
            int bt[8];
            bool in[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                in[k] = false; wd[k] = 0ull; bt[k] = 0; sx[k] = sy[k] = sz[k] = 0.f;
                if (k < nblk && __builtin_amdgcn_ballot_w64(tv[k] >= t_in && tv[k] <= t_out) != 0ull) in[k] = probe(tv[k], sx[k], sy[k], sz[k], wd[k], bt[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nblk && base < SR) keep(in[k] && ((wd[k] >> bt[k]) & 1ull), sx[k], sy[k], sz[k]);
        } else {
            for (int d0 = 0; d0 < D; d0 += 64) {
                const int d = d0 + lane;
                const float t = d < D ? tt[d] : NAN;
                if (__builtin_amdgcn_ballot_w64(t >= t_in && t <= t_out) == 0ull) continue;     // (a NaN depth passes no test here and none below)
                float sx, sy, sz;
                unsigned long long word;
                int bit;
                const bool in = probe(t, sx, sy, sz, word, bit);
                keep(in && ((word >> bit) & 1ull), sx, sy, sz);
                if (base >= SR) break;
            }
        }
        const int ns = base < SR ? base : SR;
        if (pad) {
            // pad: sample_loc zeros (torch.zeros, :647), sample_pidx -1 (torch.full, :648); kept slots are written by the k-NN
            for (int i = ns * 3 + lane; i < SR * 3; i += 64) loc_r[i] = 0.f;
            int32_t *pidx_r = pidx + (size_t)r * SR * K;
            for (int i = ns * K + lane; i < SR * K; i += 64) pidx_r[i] = -1;
        }
        if (lane == 0) {
            ray_nsamp[r] = ns;
            ray_mask[r] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Work list of kept samples in (ray, slot) order, by a two-level exclusive scan of ray_nsamp.
// No atomics: 285k same-address atomics cost ~6.5 ms on MI355X (measured, profiles/r01_query_v1).
__global__ __launch_bounds__(1024) void nsamp_block_sum_kernel(const int32_t *__restrict__ ray_nsamp, int R,
                                                               int32_t *__restrict__ block_sums)
{
    __shared__ int s_a[16], s_b[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int ns = i < R ? ray_nsamp[i] : 0;
    int hit = ns > 0;
    for (int o = 32; o > 0; o >>= 1) { ns += __shfl_xor(ns, o); hit += __shfl_xor(hit, o); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = ns; s_b[threadIdx.x >> 6] = hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; }
        block_sums[2 * blockIdx.x] = a;
        block_sums[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(1024) void worklist_kernel(const int32_t *__restrict__ ray_nsamp, int R, int SR,
                                                        const int32_t *__restrict__ block_sums, int nblocks,
                                                        int32_t *__restrict__ work, unsigned long long *__restrict__ counts)
{
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int part = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) part += block_sums[2 * k];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) s_w[wid] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += s_w[k];
        s_base = t;
    }
    __syncthreads();
    const int ns = i < R ? ray_nsamp[i] : 0;
    int inc = ns;                                   // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    int off = s_base + inc - ns;
    for (int k = 0; k < wid; ++k) off += s_w[k];
    for (int j = 0; j < ns; ++j) work[off + j] = i * SR + j;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int k = 0; k < nblocks; ++k) { a += (unsigned)block_sums[2 * k]; b += (unsigned)block_sums[2 * k + 1]; }
        counts[HNR_CNT_SAMPLES] = a;
        counts[HNR_CNT_RAYS_HIT] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// k-NN: one lane per kept shading sample, grid-stride over the work list.
// The K-entry buffer lives in registers (K is a template parameter; all indexing is unrolled).
template <int K>
struct KBuf {
    float d2[K];
    int id[K];
    int kid, far_ind;
    float far2;
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < K; ++i) { d2[i] = 0.f; id[i] = -1; }
        kid = 0; far_ind = 0; far2 = 0.f;
    }
    // The same rule without divergent branches (selects only; the re-scan for the new farthest entry runs when any lane of
    // the wave replaced one): `ok` = this lane really has a candidate inside the radius.
    __device__ __forceinline__ void offer_sel(float v, int p, bool ok)
    {
        const bool full = kid >= K;
        const bool ins = ok && (!full || v < far2);
        const int wp = full ? far_ind : kid;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const bool sel = ins && i == wp;
            d2[i] = sel ? v : d2[i];
            id[i] = sel ? p : id[i];
        }
        const bool grow = ok && !full && v > far2;             // filling phase: track the farthest entry (:497-500)
        far2 = grow ? v : far2;
        far_ind = grow ? kid : far_ind;
        const bool repl = ins && full;
        if (__builtin_amdgcn_ballot_w64(repl) != 0ull) {        // wave-uniform
            // (:506-511) far2 = v, then the first strictly larger entry wins, progressively: the result is the FIRST position of
            // the maximum if the maximum exceeds v, else the replaced slot itself (the entries include v at far_ind)
            float mx = d2[0];
#pragma unroll
            for (int i = 1; i < K; ++i) mx = fmaxf(mx, d2[i]);
            int ni = K - 1;
#pragma unroll
            for (int i = K - 2; i >= 0; --i) ni = d2[i] == mx ? i : ni;
            ni = mx > v ? ni : far_ind;
            far2 = repl ? mx : far2;
            far_ind = repl ? ni : far_ind;
        }
        kid += ok ? 1 : 0;
    }
    __device__ __forceinline__ void sync_kid() {}
    // reference :494-513
    __device__ __forceinline__ void offer(float v, int p)
    {
        if (kid < K) {
#pragma unroll
            for (int i = 0; i < K; ++i) if (i == kid) { d2[i] = v; id[i] = p; }
            if (v > far2) { far2 = v; far_ind = kid; }
            ++kid;
        } else {
            ++kid;
            if (v < far2) {
#pragma unroll
                for (int i = 0; i < K; ++i) if (i == far_ind) { d2[i] = v; id[i] = p; }
                far2 = v;
#pragma unroll
                for (int i = 0; i < K; ++i) if (d2[i] > far2) { far2 = d2[i]; far_ind = i; }
            }
        }
    }
};

template <int K>
__global__ __launch_bounds__(256) void knn_kernel(GridView g, const int32_t *__restrict__ work,
                                                  const float *__restrict__ loc, int SR, float radius2, int layers,
                                                  int32_t *__restrict__ pidx, int8_t *__restrict__ ray_mask,
                                                  const unsigned long long *__restrict__ counts,
                                                  unsigned long long *__restrict__ block_stats)
{
    __shared__ unsigned long long s_st[4][4];
    const int n = (int)counts[HNR_CNT_SAMPLES];
    unsigned long long n_cells = 0, n_cand = 0, n_nb = 0, n_sv = 0;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
        const int item = work[w];
        const float cx = loc[3 * (size_t)item], cy = loc[3 * (size_t)item + 1], cz = loc[3 * (size_t)item + 2];
        const int fx = cell_coord(cx, g.ox, g.cx), fy = cell_coord(cy, g.oy, g.cy), fz = cell_coord(cz, g.oz, g.cz);
        KBuf<K> kb;
        kb.init();
        for (int layer = 0; layer < layers; ++layer) {
            const int xlo = max(-fx, -layer), xhi = min(g.dx - fx, layer + 1);
            const int ylo = max(-fy, -layer), yhi = min(g.dy - fy, layer + 1);
            const int zlo = max(-fz, -layer), zhi = min(g.dz - fz, layer + 1);
            for (int x = xlo; x < xhi; ++x)
                for (int y = ylo; y < yhi; ++y)
                    for (int z = zlo; z < zhi; ++z) {
                        if (max(abs(z), max(abs(x), abs(y))) != layer) continue;
                        const int vx = fx + x, vy = fy + y, vz = fz + z;
                        const uint4 rec = g.occ_rec[brick_word(g, vx, vy, vz)];
                        const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
                        const int b = brick_bit(vx, vy, vz);
                        if (!((bb >> b) & 1ull)) continue;
                        const int2 rg = g.cell_rng[rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull))];
                        ++n_cells;
                        n_cand += (unsigned)rg.y;
                        for (int j = 0; j < rg.y; ++j) {
                            const float4 p = g.pts[rg.x + j];
                            const float xv = __fsub_rn(p.x, cx), yv = __fsub_rn(p.y, cy), zv = __fsub_rn(p.z, cz);
                            const float v = __fadd_rn(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv)), __fmul_rn(zv, zv));
                            if (radius2 == 0.f || v <= radius2) kb.offer(v, __float_as_int(p.w));
                        }
                    }
            if (kb.kid >= K) break;
        }
        {   // every kept sample gets its K ids (-1 where empty): the march kernel pads only the unused slots
            int32_t *o = pidx + (size_t)item * K;
            if constexpr ((K & 3) == 0) {
#pragma unroll
                for (int i = 0; i < K; i += 4)
                    reinterpret_cast<int4 *>(o)[i >> 2] = make_int4(kb.id[i], kb.id[i + 1], kb.id[i + 2], kb.id[i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) o[i] = kb.id[i];
            }
        }
        if (kb.kid > 0) {
            ray_mask[item / SR] = 1;
            n_nb += (unsigned)(kb.kid < K ? kb.kid : K);
            ++n_sv;
        }
    }
    // one atomic per wave and counter
    for (int o = 32; o > 0; o >>= 1) {
        n_cells += __shfl_xor(n_cells, o);
        n_cand += __shfl_xor(n_cand, o);
        n_nb += __shfl_xor(n_nb, o);
        n_sv += __shfl_xor(n_sv, o);
    }
    // per-block partial sums (plain stores); knn_finalize_kernel adds them up -- no same-address atomics
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_st[wv][0] = n_cells; s_st[wv][1] = n_cand; s_st[wv][2] = n_nb; s_st[wv][3] = n_sv;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        block_stats[4 * (size_t)blockIdx.x + threadIdx.x] =
            s_st[0][threadIdx.x] + s_st[1][threadIdx.x] + s_st[2][threadIdx.x] + s_st[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// k-NN, pipelined (K = 8, 3x3x3, the shipped configuration).  Same visiting order and insertion rule as knn_kernel /
// the oracle -- results are bit-identical -- but the candidate records are no longer fetched one dependent
// load per loop trip (rocprofv3: ~1 us per trip and wave slot = one exposed L2/HBM round trip, 0.9 ms per frame):
//  pass 1: per x-plane, 9 brick records in flight, then the 9 {start,count} records of the occupied cells in flight; each
//          cell's list is parked in LDS as one word (start << 6 | count; P <= 63, N < 2^26 checked by the launcher);
// Set-exact mode (hnr_query_params.knn_order = 1): the same neighbour SET as the rule above -- the K smallest candidates by (d^2, enumeration
// order), because a later candidate replaces the farthest entry only if it is STRICTLY closer -- kept as a sorted list, so the output
// is in the canonical order ascending (d^2, enumeration order) instead of the reference's insertion-history order (the
// consumers are order-free: every use is a sum over the K slots).  Sorted insertion of v into d[0] <= ... <= d[K-1]:
//   d'[i] = median(d[i-1], d[i], v) = v_med3_f32 (one instruction per slot),   id'[i] = v < d[i-1] ? id[i-1] : v < d[i] ? p : id[i]
// = 4 VALU per slot and no re-scan for the farthest entry, against ~9 per slot for the slot-exact rule.
template <int K>
struct KSorted {
    float d2[K];
    int id[K];
    int kid;
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < K; ++i) { d2[i] = __int_as_float(0x7f800000); id[i] = -1; }
        kid = 0;
    }
    __device__ __forceinline__ void offer_sel(float v, int p, bool ok)
    {
        const float vv = ok ? v : __int_as_float(0x7f800000);       // a rejected candidate is +inf: it is never < an entry
        if constexpr (K == 8) {
            // the 4-instructions-per-slot form spelled out: left to itself the compiler turns the selects into exec-masked branches and
            // copies (~105 instructions per candidate in the inner loop).  The eight compares come first (farthest slot first) so that
            // every lane mask is at least two instructions old when a v_cndmask reads it (VALU-writes-SGPR -> VALU-reads-SGPR hazard:
            // the assembler does not pad inline asm).  Slots are updated from the far end: slot i reads the OLD slot i - 1.
            unsigned long long c0, c1, c2, c3, c4, c5, c6, c7;
            asm volatile(
                "v_cmp_lt_f32_e64 %[c7], %[vv], %[d7]\n\tv_cmp_lt_f32_e64 %[c6], %[vv], %[d6]\n\t"
                "v_cmp_lt_f32_e64 %[c5], %[vv], %[d5]\n\tv_cmp_lt_f32_e64 %[c4], %[vv], %[d4]\n\t"
                "v_cmp_lt_f32_e64 %[c3], %[vv], %[d3]\n\tv_cmp_lt_f32_e64 %[c2], %[vv], %[d2]\n\t"
                "v_cmp_lt_f32_e64 %[c1], %[vv], %[d1]\n\tv_cmp_lt_f32_e64 %[c0], %[vv], %[d0]\n\t"
                "v_cndmask_b32_e64 %[i7], %[i7], %[p], %[c7]\n\tv_med3_f32 %[d7], %[d6], %[d7], %[vv]\n\tv_cndmask_b32_e64 %[i7], %[i7], %[i6], %[c6]\n\t"
                "v_cndmask_b32_e64 %[i6], %[i6], %[p], %[c6]\n\tv_med3_f32 %[d6], %[d5], %[d6], %[vv]\n\tv_cndmask_b32_e64 %[i6], %[i6], %[i5], %[c5]\n\t"
                "v_cndmask_b32_e64 %[i5], %[i5], %[p], %[c5]\n\tv_med3_f32 %[d5], %[d4], %[d5], %[vv]\n\tv_cndmask_b32_e64 %[i5], %[i5], %[i4], %[c4]\n\t"
                "v_cndmask_b32_e64 %[i4], %[i4], %[p], %[c4]\n\tv_med3_f32 %[d4], %[d3], %[d4], %[vv]\n\tv_cndmask_b32_e64 %[i4], %[i4], %[i3], %[c3]\n\t"
                "v_cndmask_b32_e64 %[i3], %[i3], %[p], %[c3]\n\tv_med3_f32 %[d3], %[d2], %[d3], %[vv]\n\tv_cndmask_b32_e64 %[i3], %[i3], %[i2], %[c2]\n\t"
                "v_cndmask_b32_e64 %[i2], %[i2], %[p], %[c2]\n\tv_med3_f32 %[d2], %[d1], %[d2], %[vv]\n\tv_cndmask_b32_e64 %[i2], %[i2], %[i1], %[c1]\n\t"
                "v_cndmask_b32_e64 %[i1], %[i1], %[p], %[c1]\n\tv_med3_f32 %[d1], %[d0], %[d1], %[vv]\n\tv_cndmask_b32_e64 %[i1], %[i1], %[i0], %[c0]\n\t"
                "v_cndmask_b32_e64 %[i0], %[i0], %[p], %[c0]\n\tv_min_f32 %[d0], %[d0], %[vv]"
                : [d0] "+v"(d2[0]), [d1] "+v"(d2[1]), [d2] "+v"(d2[2]), [d3] "+v"(d2[3]), [d4] "+v"(d2[4]), [d5] "+v"(d2[5]), [d6] "+v"(d2[6]),
                  [d7] "+v"(d2[7]), [i0] "+v"(id[0]), [i1] "+v"(id[1]), [i2] "+v"(id[2]), [i3] "+v"(id[3]), [i4] "+v"(id[4]), [i5] "+v"(id[5]),
                  [i6] "+v"(id[6]), [i7] "+v"(id[7]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [c4] "=&s"(c4),
                  [c5] "=&s"(c5), [c6] "=&s"(c6), [c7] "=&s"(c7)
                : [vv] "v"(vv), [p] "v"(p));
        } else {
            bool c[K];
#pragma unroll
            for (int i = 0; i < K; ++i) c[i] = vv < d2[i];              // on the OLD list; monotone in i
#pragma unroll
            for (int i = K - 1; i >= 1; --i) {
                id[i] = c[i - 1] ? id[i - 1] : (c[i] ? p : id[i]);
                d2[i] = __builtin_amdgcn_fmed3f(d2[i - 1], d2[i], vv);
            }
            id[0] = c[0] ? p : id[0];
            d2[0] = fminf(d2[0], vv);
        }
    }
    // entries held (<= K): the list itself says it (an empty slot is +inf), so the insertion carries no counter
    __device__ __forceinline__ void sync_kid()
    {
        kid = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) kid += d2[i] < __int_as_float(0x7f800000) ? 1 : 0;
    }
};

//  pass 2: per shell, an address generator walks the occupied cells and a 4-deep register ring keeps four candidate
//          loads in flight ahead of the insertion; stepping to the next cell costs no loop trip; the insertion is select-only.
// After this the kernel is VALU-bound on the insertion rule (rocprofv3 --pmc: 303 M VALU wave-instructions per frame = 0.56 ms of
// issue time on 1024 SIMDs, SQ_WAIT_INST_ANY 17 % of the wave cycles).  Measured and dropped on top of it (profiles/README.md):
// re-dealing a block's samples to lanes by candidate count + one stream over both shells (lane slots 215 M -> 132 M, same
// time: the extra bookkeeping costs what the idle lanes did); 4-record chunks per trip (1.2x slower: more slots, more loads).
template <int K, int SORTED = 0>
__global__ __launch_bounds__(256) void knn3_kernel(GridView g, const int32_t *__restrict__ work, const float *__restrict__ loc,
                                                   int SR, float radius2, int layers, int32_t *__restrict__ pidx,
                                                   int8_t *__restrict__ ray_mask, const unsigned long long *__restrict__ counts,
                                                   unsigned long long *__restrict__ block_stats)
{
    __shared__ unsigned long long s_st[4][4];
    __shared__ uint32_t s_cell[27][256];                      // [cell][thread]: conflict-free (consecutive lanes, consecutive banks)
    const int n = (int)counts[HNR_CNT_SAMPLES];
    unsigned n_cells = 0, n_cand = 0, n_nb = 0, n_sv = 0;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
        const int item = work[w];
        const float cx = loc[3 * (size_t)item], cy = loc[3 * (size_t)item + 1], cz = loc[3 * (size_t)item + 2];
        const int fx = cell_coord(cx, g.ox, g.cx), fy = cell_coord(cy, g.oy, g.cy), fz = cell_coord(cz, g.oz, g.cz);
        uint32_t occ = 0, occ_nz = 0;                          // occupied cells / those that list at least one point
        unsigned cand0 = 0, cand1 = 0;
#pragma unroll 1
        for (int xp = 0; xp < 3; ++xp) {
            uint32_t slot[9];
            uint32_t ob = 0, onz = 0;
#pragma unroll
            for (int yz = 0; yz < 9; ++yz) {
                const int x = xp - 1, y = yz / 3 - 1, z = yz % 3 - 1;
                const int vx = fx + x, vy = fy + y, vz = fz + z;
                const bool inb = in_bounds(g, vx, vy, vz);
                const int qx = inb ? vx : fx, qy = inb ? vy : fy, qz = inb ? vz : fz;     // clamp: the load is unconditional
                const uint4 rec = g.occ_rec[brick_word(g, qx, qy, qz)];
                const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
                const int b = brick_bit(qx, qy, qz);
                const bool o = inb && ((bb >> b) & 1ull);
                ob |= (o ? 1u : 0u) << yz;
                slot[yz] = o ? rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull)) : 0u;
            }
#pragma unroll
            for (int yz = 0; yz < 9; ++yz) {
                const int c = xp * 9 + yz;
                const int2 rg = g.cell_rng[slot[yz]];                                     // slot 0 for empty cells: a valid record, unused
                const bool o = (ob >> yz) & 1u;
                s_cell[c][threadIdx.x] = o ? ((uint32_t)rg.x << 6) | (uint32_t)rg.y : 0u;
                onz |= (o && rg.y > 0 ? 1u : 0u) << yz;         // (the slot-0 voxel is occupied but lists nothing, :366)
                if (c == 13) cand0 = o ? (unsigned)rg.y : 0u; else cand1 += o ? (unsigned)rg.y : 0u;
            }
            occ |= ob << (xp * 9);
            occ_nz |= onz << (xp * 9);
        }
        typename std::conditional<SORTED != 0, KSorted<K>, KBuf<K>>::type kb;
        kb.init();
        auto run_shell = [&](uint32_t mg) {
            int st = 0, cn = 0, jg = 0;
            auto gen = [&]() -> int {                            // next candidate index of this lane, -1 when the shell is exhausted
                const bool need = jg >= cn;
                if (__builtin_amdgcn_ballot_w64(need) != 0ull) { // wave-uniform; every cell in the mask lists >= 1 point
                    const int c = mg ? __ffs((int)mg) - 1 : 0;
                    const uint32_t u = s_cell[c][threadIdx.x];
                    st = need ? (int)(u >> 6) : st;
                    cn = need ? (mg ? (int)(u & 63u) : 0) : cn;
                    jg = need ? 0 : jg;
                    mg = need ? (mg & (mg - 1)) : mg;
                }
                const int a = jg < cn ? st + jg : -1;
                ++jg;
                return a;
            };
            auto consume = [&](const float4 &p, bool have) {
                const float xv = __fsub_rn(p.x, cx), yv = __fsub_rn(p.y, cy), zv = __fsub_rn(p.z, cz);
                const float v = __fadd_rn(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv)), __fmul_rn(zv, zv));
                kb.offer_sel(v, __float_as_int(p.w), have && (radius2 == 0.f || v <= radius2));
            };
            // loads are UNCONDITIONAL (an exhausted lane re-reads record 0): no branch around them, so the four requests stay
            // in flight across the insertion code instead of being waited for one by one
            int a0 = gen(), a1 = gen(), a2 = gen(), a3 = gen();
            float4 p0 = g.pts[a0 < 0 ? 0 : a0], p1 = g.pts[a1 < 0 ? 0 : a1], p2 = g.pts[a2 < 0 ? 0 : a2], p3 = g.pts[a3 < 0 ? 0 : a3];
            if (__builtin_amdgcn_ballot_w64(a0 >= 0) != 0ull) {
                do {                                             // bottom-tested: a top test makes the compiler keep two copies of the list
                    consume(p0, a0 >= 0); a0 = gen(); p0 = g.pts[a0 < 0 ? 0 : a0];
                    consume(p1, a1 >= 0); a1 = gen(); p1 = g.pts[a1 < 0 ? 0 : a1];
                    consume(p2, a2 >= 0); a2 = gen(); p2 = g.pts[a2 < 0 ? 0 : a2];
                    consume(p3, a3 >= 0); a3 = gen(); p3 = g.pts[a3 < 0 ? 0 : a3];
                } while (__builtin_amdgcn_ballot_w64(a0 >= 0) != 0ull);
            }
        };
        const uint32_t m0 = occ & (1u << 13), m1 = occ & ~(1u << 13);
        if (layers > 0) run_shell(occ_nz & (1u << 13));         // shell 0 = the sample's own cell (layers = 0: probe of pass 1 alone)
        kb.sync_kid();
        n_cells += m0 ? 1u : 0u;
        n_cand += cand0;
        if (layers > 1 && kb.kid < K) {                         // reference: `if (kid >= K) break;` after a layer
            run_shell(occ_nz & ~(1u << 13));
            kb.sync_kid();
            n_cells += (unsigned)__popc(m1);
            n_cand += cand1;
        }
        {
            int32_t *o = pidx + (size_t)item * K;
            if constexpr ((K & 3) == 0) {
#pragma unroll
                for (int i = 0; i < K; i += 4)
                    reinterpret_cast<int4 *>(o)[i >> 2] = make_int4(kb.id[i], kb.id[i + 1], kb.id[i + 2], kb.id[i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < K; ++i) o[i] = kb.id[i];
            }
        }
        if (kb.kid > 0) {
            ray_mask[item / SR] = 1;
            n_nb += (unsigned)(kb.kid < K ? kb.kid : K);
            ++n_sv;
        }
    }
    unsigned long long t_cells = n_cells, t_cand = n_cand, t_nb = n_nb, t_sv = n_sv;
    for (int o = 32; o > 0; o >>= 1) {
        t_cells += __shfl_xor(t_cells, o);
        t_cand += __shfl_xor(t_cand, o);
        t_nb += __shfl_xor(t_nb, o);
        t_sv += __shfl_xor(t_sv, o);
    }
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_st[wv][0] = t_cells; s_st[wv][1] = t_cand; s_st[wv][2] = t_nb; s_st[wv][3] = t_sv;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        block_stats[4 * (size_t)blockIdx.x + threadIdx.x] =
            s_st[0][threadIdx.x] + s_st[1][threadIdx.x] + s_st[2][threadIdx.x] + s_st[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// k-NN over the 3x3x3 NEIGHBOURHOOD LISTS of the grid (round 5; K = 8, GridView::nb_*): the candidates of a sample's 27 cells are one contiguous
// run in the reference's enumeration order (own cell first), so pass 1 of knn3_kernel -- 27 brick records + 27 {start, count} records per sample,
// parked in LDS -- becomes TWO lookups and the 23-instruction address generator becomes a counter (54 instead of 72 instructions per candidate).
// Same visiting order and the same insertion rules (KBuf slot-exact / KSorted set-exact): bit-identical results (tests/test_query_gpu.py).
// BIN: one lane per sample is 44 % lane-efficient when the 64 samples of a wave are taken in work-list order (a wave runs as long as its longest
// candidate list).  With the list length one lookup away, a workgroup takes NB_CHUNK consecutive samples, looks their lists up once, counting-sorts them
// in LDS by the number of loop trips they need (4 candidates per trip, the two shells separately) and deals them to its lanes in that order: the
// samples of a wave need the same number of trips up to +-1.  (Round 4 tried the same re-dealing over the 27-cell walk and gained nothing: there the
// list length cost 54 lookups per sample.)  The order samples are taken in changes nothing a sample computes; the counters are integer sums.
constexpr int NB_CHUNK = 1024;        // most samples sorted per workgroup and round (4 per thread)
// Samples per workgroup round: 1024 when there is enough work for every workgroup of the launch, fewer for small batches (a training batch has ~38 k
// samples: 38 rounds of 1024 would keep 38 of 2048 workgroups busy).  (Round 5 also tried to size the rounds so that the LAST round of residency is
// full on a whole frame -- 832 samples per round on 2048 workgroups, 768 on the 1536 the LDS lets the chip hold: 0.645 / 0.69 ms for the query against
// 0.60 with rounds of 1024.  Fewer, larger rounds win: a round's cost is its chain of dependent lookups + three barriers, and the workgroups that wait
// for a slot fill the first ones' tail.)
__device__ __forceinline__ int nb_chunk_size(int n, int n_groups)
{
    int c = (n + n_groups - 1) / n_groups;
    c = (c + 63) & ~63;
    return c < 64 ? 64 : (c > NB_CHUNK ? NB_CHUNK : c);
}
constexpr int NB_BINS = 32;
// values of the other lanes of a quad (quad_perm DPP; CTRL = sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6)
template <int CTRL>
__device__ __forceinline__ float quad_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL>
__device__ __forceinline__ int quad_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// XP: the four lanes of a quad FETCH each other's candidates -- with instruction k every lane of the quad reads one record of quad-lane k's run (one 64-byte
// line per quad and instruction instead of four) -- and the records change hands through a per-wave LDS staging area (row stride 80 bytes): the address unit
// sees a quarter of the lines, and every lane still keeps its own list (31 instructions per candidate and LANE, not 14 per candidate and QUAD).
template <int K, int SORTED, int BIN, int XP = 0>
__global__ __launch_bounds__(256) void knn_nb_kernel(GridView g, const int32_t *__restrict__ work, const float *__restrict__ loc,
                                                     int SR, float radius2, int layers, int32_t *__restrict__ pidx, int8_t *__restrict__ ray_mask,
                                                     const unsigned long long *__restrict__ counts, unsigned long long *__restrict__ block_stats)
{
    __shared__ unsigned long long s_st[4][4];
    __shared__ int32_t s_item[BIN ? NB_CHUNK : 1];
    __shared__ uint2 s_rg[BIN ? NB_CHUNK : 1];
    __shared__ int s_hist[BIN == 2 ? NB_BINS * 32 : NB_BINS], s_base[BIN == 2 ? NB_BINS * 32 : NB_BINS], s_wsum[4];
    __shared__ float4 s_stage[XP ? 256 * 5 : 1];
    constexpr int NBH = 1024;                                      // BIN == 3: hash slots, one per distinct cell of a round
    __shared__ unsigned s_hkey[BIN == 3 ? NBH : 1];
    __shared__ int s_hcnt[BIN == 3 ? NBH : 1], s_hpos[BIN == 3 ? NBH : 1];
    const int n = (int)counts[HNR_CNT_SAMPLES];
    unsigned n_cells = 0, n_cand = 0, n_nb = 0, n_sv = 0;
    auto lookup = [&](int item) -> uint2 {
        const float cx = loc[3 * (size_t)item], cy = loc[3 * (size_t)item + 1], cz = loc[3 * (size_t)item + 2];
        const int fx = cell_coord(cx, g.ox, g.cx), fy = cell_coord(cy, g.oy, g.cy), fz = cell_coord(cz, g.oz, g.cz);
        // a kept sample sits in a cell of the dilated mask (the march tested that bit); anything else lists nothing
        const bool inb = in_bounds(g, fx, fy, fz);
        const uint4 rec = g.dil_rec[inb ? brick_word(g, fx, fy, fz) : 0u];
        const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
        const int b = inb ? brick_bit(fx, fy, fz) : 0;
        const bool have = inb && ((bb >> b) & 1ull);
        const uint2 rg = g.nb_rng[have ? rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull)) : 0u];
        return have ? rg : make_uint2(0u, 0u);
    };
    auto one = [&](int item, uint2 rg) {                          // XP: item < 0 = an idle lane that only helps its quad fetch (rg = {0, 0})
        const size_t lo3 = 3 * (size_t)(item < 0 ? 0 : item);
        const float cx = loc[lo3], cy = loc[lo3 + 1], cz = loc[lo3 + 2];
        const int start = (int)rg.x;
        const int c0 = (int)(rg.y & 63u), tot = (int)((rg.y >> 6) & 2047u);
        typename std::conditional<SORTED != 0, KSorted<K>, KBuf<K>>::type kb;
        kb.init();
        auto run = [&](int lo, int hi) {
            int j = lo;
            auto gen = [&]() -> int { const int a = j < hi ? j : -1; ++j; return a; };
            auto consume = [&](const float4 &p, bool ok) {
                const float xv = __fsub_rn(p.x, cx), yv = __fsub_rn(p.y, cy), zv = __fsub_rn(p.z, cz);
                const float v = __fadd_rn(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv)), __fmul_rn(zv, zv));
                kb.offer_sel(v, __float_as_int(p.w), ok && (radius2 == 0.f || v <= radius2));
            };
            if constexpr (XP != 0) {
                // lo is a multiple of 4 (runs and both of their parts start on a line); a lane that has nothing left keeps fetching its first line
                const int q = threadIdx.x & 3;
                float4 *wr = s_stage + (size_t)(threadIdx.x & ~3) * 5 + q, *rd = s_stage + (size_t)threadIdx.x * 5;
                auto fetch = [&](int jj, float4 (&rec)[4]) {
                    rec[0] = g.nb_pts[quad_i<0x00>(jj) + q]; rec[1] = g.nb_pts[quad_i<0x55>(jj) + q];
                    rec[2] = g.nb_pts[quad_i<0xaa>(jj) + q]; rec[3] = g.nb_pts[quad_i<0xff>(jj) + q];
                };
                bool live = j < hi;
                float4 rec[4];
                fetch(live ? j : lo, rec);
                while (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    float4 p[4];
                    wr[0] = rec[0]; wr[5] = rec[1]; wr[10] = rec[2]; wr[15] = rec[3];      // record q of quad-lane k's line -> row of lane k, slot q
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    p[0] = rd[0]; p[1] = rd[1]; p[2] = rd[2]; p[3] = rd[3];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();                                        // (the rows are rewritten by the next trip)
                    const int jn = j + 4;
                    const bool ln = live && jn < hi;
                    fetch(ln ? jn : lo, rec);
                    consume(p[0], live); consume(p[1], live && j + 1 < hi); consume(p[2], live && j + 2 < hi); consume(p[3], live && j + 3 < hi);
                    j = jn; live = ln;
                }
                return;
            }
            // unconditional loads (an exhausted lane re-reads record 0): four requests stay in flight across the insertion code
            int a0 = gen(), a1 = gen(), a2 = gen(), a3 = gen();
            float4 p0 = g.nb_pts[a0 < 0 ? 0 : a0], p1 = g.nb_pts[a1 < 0 ? 0 : a1], p2 = g.nb_pts[a2 < 0 ? 0 : a2], p3 = g.nb_pts[a3 < 0 ? 0 : a3];
            if (__builtin_amdgcn_ballot_w64(a0 >= 0) != 0ull) {
                do {
                    consume(p0, a0 >= 0); a0 = gen(); p0 = g.nb_pts[a0 < 0 ? 0 : a0];
                    consume(p1, a1 >= 0); a1 = gen(); p1 = g.nb_pts[a1 < 0 ? 0 : a1];
                    consume(p2, a2 >= 0); a2 = gen(); p2 = g.nb_pts[a2 < 0 ? 0 : a2];
                    consume(p3, a3 >= 0); a3 = gen(); p3 = g.nb_pts[a3 < 0 ? 0 : a3];
                } while (__builtin_amdgcn_ballot_w64(a0 >= 0) != 0ull);
            }
        };
        if (layers > 0) run(start, start + c0);                  // shell 0 = the sample's own cell
        kb.sync_kid();
        n_cells += (rg.y >> 22) & 1u;
        n_cand += (unsigned)c0;
        const bool shell1 = layers > 1 && kb.kid < K;             // reference: `if (kid >= K) break;` after a layer
        if (XP != 0 || shell1) {                                 // (XP: every lane takes part in its quad's fetches, with an empty range if it is done)
            const int s1 = start + ((c0 + 3) & ~3);              // (the own-cell part of a run is padded to a multiple of 4 entries)
            run(s1, shell1 ? s1 + tot - c0 : s1);
        }
        if (shell1) {
            kb.sync_kid();
            n_cells += (rg.y >> 17) & 31u;
            n_cand += (unsigned)(tot - c0);
        }
        if (item < 0) return;
        {
            int32_t *o = pidx + (size_t)item * K;
#pragma unroll
            for (int i = 0; i < K; i += 4)
                reinterpret_cast<int4 *>(o)[i >> 2] = make_int4(kb.id[i], kb.id[i + 1], kb.id[i + 2], kb.id[i + 3]);
        }
        if (kb.kid > 0) {
            ray_mask[item / SR] = 1;
            n_nb += (unsigned)(kb.kid < K ? kb.kid : K);
            ++n_sv;
        }
    };
    if constexpr (BIN == 3) {
        // BIN == 3: samples of one CELL next to each other, cells in the order of their list length (longest first).  The samples of a round find their cell
        // in an LDS hash table keyed by the run address (one slot per distinct cell, the slot counts its samples and hands out ranks), the cells take their
        // place inside their list-length bin with one atomic each, and a sample lands at bin base + cell base + rank.  Neighbouring lanes then read the SAME
        // records: a load instruction of a wave touches fewer cache lines than with the cells interleaved (BIN == 2 groups by a 5-bit hash only: the bins of
        // the common list lengths still interleave ~5 cells).
        const int chunk = nb_chunk_size(n, (int)gridDim.x);
        for (int w0 = blockIdx.x * chunk; w0 < n; w0 += gridDim.x * chunk) {
            for (int i = threadIdx.x; i < NBH; i += 256) { s_hkey[i] = 0u; s_hcnt[i] = 0; }
            if (threadIdx.x < NB_BINS) s_hist[threadIdx.x] = 0;
            __syncthreads();
            int item[4], slot[4], rank[4];
            uint2 rg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                          // (all four lookups in flight)
                const int wi = q * 256 + (int)threadIdx.x, w = w0 + wi;
                item[q] = (wi < chunk && w < n) ? work[w] : -1;
                rg[q] = item[q] >= 0 ? lookup(item[q]) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                slot[q] = 0; rank[q] = 0;
                if (item[q] >= 0) {
                    const int c0 = (int)(rg[q].y & 63u), tot = (int)((rg[q].y >> 6) & 2047u);
                    const int trips = ((c0 + 3) >> 2) + ((tot - c0 + 3) >> 2);
                    const int bin = NB_BINS - 1 - (trips < NB_BINS - 1 ? trips : NB_BINS - 1);      // longest lists first
                    const unsigned key = rg[q].y ? rg[q].x + 1u : 0x7fffffffu;                      // (run addresses are unique per cell; no run: one slot for all)
                    int sl = (int)((key * 2654435761u) >> 22) & (NBH - 1);
                    for (;;) {
                        const unsigned old = atomicCAS(&s_hkey[sl], 0u, key);
                        if (old == 0u) { s_hpos[sl] = bin; break; }
                        if (old == key) break;
                        sl = (sl + 1) & (NBH - 1);
                    }
                    slot[q] = sl;
                    rank[q] = atomicAdd(&s_hcnt[sl], 1);
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < NBH; i += 256)
                if (s_hkey[i] != 0u) { const int bin = s_hpos[i]; s_hpos[i] = atomicAdd(&s_hist[bin], s_hcnt[i]) | (bin << 16); }
            __syncthreads();
            if (threadIdx.x < 64) {                                // exclusive scan of the 32 bin totals
                const int t = threadIdx.x;
                const int v = t < NB_BINS ? s_hist[t] : 0;
                int inc = v;
                for (int o = 1; o < NB_BINS; o <<= 1) { const int u = __shfl_up(inc, o); if (t >= o) inc += u; }
                if (t < NB_BINS) s_base[t] = inc - v;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (item[q] >= 0) { const int hp = s_hpos[slot[q]]; const int pos = s_base[hp >> 16] + (hp & 0xffff) + rank[q]; s_item[pos] = item[q]; s_rg[pos] = rg[q]; }
            __syncthreads();
            const int m = min(chunk, n - w0);
            for (int t = threadIdx.x; t < m; t += 256) one(s_item[t], s_rg[t]);
            __syncthreads();
        }
    } else if constexpr (BIN != 0) {
        // BIN == 2: buckets = (loop trips, 5 bits of the cell's run address): samples of one cell share a bucket and end up in neighbouring lanes, so the
        // four lanes of a quad mostly read the SAME record with a load instruction (one cache line per quad instead of four)
        constexpr int NBK = BIN == 2 ? NB_BINS * 32 : NB_BINS;
        const int chunk = nb_chunk_size(n, (int)gridDim.x);
        for (int w0 = blockIdx.x * chunk; w0 < n; w0 += gridDim.x * chunk) {
            for (int i = threadIdx.x; i < NBK; i += 256) s_hist[i] = 0;
            __syncthreads();
            int item[4], key[4], rank[4];
            uint2 rg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                          // (all four lookups in flight)
                const int wi = q * 256 + (int)threadIdx.x, w = w0 + wi;
                item[q] = (wi < chunk && w < n) ? work[w] : -1;
                rg[q] = item[q] >= 0 ? lookup(item[q]) : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (int)(rg[q].y & 63u), tot = (int)((rg[q].y >> 6) & 2047u);
                const int trips = ((c0 + 3) >> 2) + ((tot - c0 + 3) >> 2);
                key[q] = trips < NB_BINS - 1 ? trips : NB_BINS - 1;
                key[q] = NB_BINS - 1 - key[q];                     // longest lists first
                if constexpr (BIN == 2) key[q] = key[q] * 32 + (int)((rg[q].x >> 2) & 31u);
                rank[q] = item[q] >= 0 ? atomicAdd(&s_hist[key[q]], 1) : 0;
            }
            __syncthreads();
            {   // exclusive scan of the bucket counts: 4 (or 1/8) per thread, wave scan, wave totals through LDS
                constexpr int PER = (NBK + 255) / 256;
                int v[PER], sum = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) { const int b = (int)threadIdx.x * PER + i; v[i] = b < NBK ? s_hist[b] : 0; sum += v[i]; }
                int inc = sum;
                const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
                for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
                if (lane == 63) s_wsum[wv] = inc;
                __syncthreads();
                int base = inc - sum;
                for (int k = 0; k < wv; ++k) base += s_wsum[k];
#pragma unroll
                for (int i = 0; i < PER; ++i) { const int b = (int)threadIdx.x * PER + i; if (b < NBK) s_base[b] = base; base += v[i]; }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (item[q] >= 0) { const int pos = s_base[key[q]] + rank[q]; s_item[pos] = item[q]; s_rg[pos] = rg[q]; }
            __syncthreads();
            const int m = min(chunk, n - w0);
            if constexpr (XP != 0) {
                for (int t0 = 0; t0 < m; t0 += 256) { const int t = t0 + (int)threadIdx.x; const bool ok = t < m; one(ok ? s_item[t] : -1, ok ? s_rg[t] : make_uint2(0u, 0u)); }
            } else {
                for (int t = threadIdx.x; t < m; t += 256) one(s_item[t], s_rg[t]);
            }
            __syncthreads();
        }
    } else {
        for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
            const int item = work[w];
            one(item, lookup(item));
        }
    }
    unsigned long long t_cells = n_cells, t_cand = n_cand, t_nb = n_nb, t_sv = n_sv;
    for (int o = 32; o > 0; o >>= 1) {
        t_cells += __shfl_xor(t_cells, o);
        t_cand += __shfl_xor(t_cand, o);
        t_nb += __shfl_xor(t_nb, o);
        t_sv += __shfl_xor(t_sv, o);
    }
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_st[wv][0] = t_cells; s_st[wv][1] = t_cand; s_st[wv][2] = t_nb; s_st[wv][3] = t_sv;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        block_stats[4 * (size_t)blockIdx.x + threadIdx.x] =
            s_st[0][threadIdx.x] + s_st[1][threadIdx.x] + s_st[2][threadIdx.x] + s_st[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// k-NN over the neighbourhood lists, FOUR LANES PER SAMPLE (set-exact order, K = 8): the production kernel.
// With the candidates contiguous, knn_nb_kernel (one lane per sample) stopped being bound by its instruction count (111 M VALU wave-instructions per frame
// against 279 M for the 27-cell walk) and became bound by the texture addresser: every lane reads its own 16-byte record, so a wave's load instruction
// touches 64 cache lines = 64 address cycles for 1 KiB (SQ_WAIT_INST_ANY 54 % of the wave cycles, rocprofv3 --pmc, profiles/r05_query_pmc.txt).  Here a
// QUAD owns a sample: its four lanes read four CONSECUTIVE candidates -- one 64-byte line (runs are padded to lines, grid.hip) -- so a wave instruction
// touches 16 lines for the same 1 KiB; each lane computes one distance, and the four candidates enter the sample's sorted list one after the other, the
// list itself spread over the quad (lane i holds entries 2 i and 2 i + 1) with the neighbours' entries and the broadcast candidate moving through
// quad_perm DPP -- 14 instructions per candidate and quad instead of 31 per candidate and lane.  Ascending (d2, enumeration order) exactly as KSorted:
// a candidate is placed before the entries it is STRICTLY smaller than.

struct QuadList {
    float a, b;           // entries 2 q and 2 q + 1 of the quad's ascending list (q = lane & 3)
    int ia, ib;
    __device__ __forceinline__ void init() { a = b = __int_as_float(0x7f800000); ia = ib = -1; }
    // The quad's four candidates (lane s holds value v / id p; +inf = none) enter the list one after the other.  Per candidate, 14 instructions without a
    // branch (left to itself the compiler turns the selects into exec-masked branches: 135 instructions per trip instead of 75):
    //   vb, pb = the candidate, broadcast;  L, iL = the entry just below this lane's pair (quad_perm [0,0,1,2]; lane 0: -inf)
    //   b' = med3(a, b, vb)   ib' = vb < a ? ia : vb < b ? pb : ib        a' = med3(L, a, vb)   ia' = vb < L ? iL : vb < a ? pb : ia
    // Spacing: a DPP operand must be two instructions old (the assembler does not pad inline asm) -- b / ib are rewritten >= 5 instructions before the next
    // step reads them through DPP -- and every compare result is >= 2 instructions old when a v_cndmask reads it.
    __device__ __forceinline__ void insert4(float v, int p, unsigned long long m_first)
    {
        float vb, L; int pb, iL;
        unsigned long long cL, ca, cb;
        const float ninf = __int_as_float(0xff800000);
#define HNR_QSTEP(S)                                                                                                   \
        "v_mov_b32_dpp %[vb], %[v] quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"                \
        "v_mov_b32_dpp %[pb], %[p] quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"                \
        "v_mov_b32_dpp %[L], %[b] quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"                                   \
        "v_mov_b32_dpp %[iL], %[ib] quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"                                 \
        "v_cmp_lt_f32_e64 %[cb], %[vb], %[b]\n\t"                                                                       \
        "v_cmp_lt_f32_e64 %[ca], %[vb], %[a]\n\t"                                                                       \
        "v_cndmask_b32_e64 %[L], %[L], %[ninf], %[mf]\n\t"                                                              \
        "v_cndmask_b32_e64 %[ib], %[ib], %[pb], %[cb]\n\t"                                                              \
        "v_cmp_lt_f32_e64 %[cL], %[vb], %[L]\n\t"                                                                       \
        "v_med3_f32 %[b], %[a], %[b], %[vb]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[ib], %[ib], %[ia], %[ca]\n\t"                                                              \
        "v_cndmask_b32_e64 %[ia], %[ia], %[pb], %[ca]\n\t"                                                              \
        "v_med3_f32 %[a], %[L], %[a], %[vb]\n\t"                                                                        \
        "v_cndmask_b32_e64 %[ia], %[ia], %[iL], %[cL]\n\t"
        asm volatile("s_nop 1\n\t" HNR_QSTEP(0) HNR_QSTEP(1) HNR_QSTEP(2) HNR_QSTEP(3)
                     : [a] "+v"(a), [b] "+v"(b), [ia] "+v"(ia), [ib] "+v"(ib), [vb] "=&v"(vb), [pb] "=&v"(pb), [L] "=&v"(L), [iL] "=&v"(iL),
                       [cL] "=&s"(cL), [ca] "=&s"(ca), [cb] "=&s"(cb)
                     : [v] "v"(v), [p] "v"(p), [ninf] "v"(ninf), [mf] "s"(m_first));
#undef HNR_QSTEP
    }
};

template <int BIN>
__global__ __launch_bounds__(256) void knn_quad_kernel(GridView g, const int32_t *__restrict__ work, const float *__restrict__ loc,
                                                       int SR, float radius2, int layers, int32_t *__restrict__ pidx, int8_t *__restrict__ ray_mask,
                                                       const unsigned long long *__restrict__ counts, unsigned long long *__restrict__ block_stats)
{
    constexpr int K = 8;
    __shared__ unsigned long long s_st[4][4];
    __shared__ float4 s_loc[NB_CHUNK + 64];                        // {x, y, z, item} in processing order (+ one idle round of zero records)
    __shared__ uint2 s_rg[NB_CHUNK + 64];
    __shared__ int s_hist[NB_BINS], s_base[NB_BINS];
    const int n = (int)counts[HNR_CNT_SAMPLES];
    const int q = threadIdx.x & 3, quad = threadIdx.x >> 2;
    unsigned n_cells = 0, n_cand = 0, n_nb = 0, n_sv = 0;          // counted by lane 0 of each quad
    const float inf = __int_as_float(0x7f800000);
    auto lookup = [&](float cx, float cy, float cz) -> uint2 {
        const int fx = cell_coord(cx, g.ox, g.cx), fy = cell_coord(cy, g.oy, g.cy), fz = cell_coord(cz, g.oz, g.cz);
        const bool inb = in_bounds(g, fx, fy, fz);
        const uint4 rec = g.dil_rec[inb ? brick_word(g, fx, fy, fz) : 0u];
        const unsigned long long bb = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
        const int b = inb ? brick_bit(fx, fy, fz) : 0;
        const bool have = inb && ((bb >> b) & 1ull);
        const uint2 rg = g.nb_rng[have ? rec.z + (uint32_t)__popcll(bb & ((1ull << b) - 1ull)) : 0u];
        return have ? rg : make_uint2(0u, 0u);
    };
    const int chunk = nb_chunk_size(n, (int)gridDim.x);
    for (int w0 = blockIdx.x * chunk; w0 < n; w0 += gridDim.x * chunk) {
        const int m = min(chunk, n - w0);
        // ---- phase A: the chunk's samples with their positions and run records into LDS, in the order they will be processed
        if (threadIdx.x < NB_BINS) s_hist[threadIdx.x] = 0;
        if (threadIdx.x < 64) { s_loc[m + threadIdx.x] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)); s_rg[m + threadIdx.x] = make_uint2(0u, 0u); }
        __syncthreads();
        {
            int item[4], key[4], rank[4];
            float4 lc[4];
            uint2 rg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wi = u * 256 + (int)threadIdx.x, w = w0 + wi;
                item[u] = (wi < chunk && w < n) ? work[w] : -1;
                const size_t o = 3 * (size_t)(item[u] < 0 ? 0 : item[u]);
                lc[u] = make_float4(loc[o], loc[o + 1], loc[o + 2], __int_as_float(item[u]));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) rg[u] = item[u] >= 0 ? lookup(lc[u].x, lc[u].y, lc[u].z) : make_uint2(0u, 0u);
            if constexpr (BIN != 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c0 = (int)(rg[u].y & 63u), tot = (int)((rg[u].y >> 6) & 2047u);
                    const int trips = ((c0 + 3) >> 2) + ((tot - c0 + 3) >> 2);
                    key[u] = trips < NB_BINS - 1 ? trips : NB_BINS - 1;
                    rank[u] = item[u] >= 0 ? atomicAdd(&s_hist[key[u]], 1) : 0;
                }
                __syncthreads();
                if (threadIdx.x < 64) {                            // exclusive scan of the bin counts, longest lists first
                    const int t = threadIdx.x;
                    int v = t < NB_BINS ? s_hist[NB_BINS - 1 - t] : 0, inc = v;
                    for (int o = 1; o < NB_BINS; o <<= 1) { const int u2 = __shfl_up(inc, o); if (t >= o) inc += u2; }
                    if (t < NB_BINS) s_base[NB_BINS - 1 - t] = inc - v;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (item[u] >= 0) { const int pos = s_base[key[u]] + rank[u]; s_loc[pos] = lc[u]; s_rg[pos] = rg[u]; }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (item[u] >= 0) { const int pos = u * 256 + (int)threadIdx.x; s_loc[pos] = lc[u]; s_rg[pos] = rg[u]; }
            }
        }
        __syncthreads();
        // ---- phase B: a quad per sample.  The first line of BOTH shells of the next sample is requested before the current one is worked on, and inside
        //      a shell the next line is in flight across the four insertions: no load is waited for right after its issue.
        int t = quad;
        float4 c = s_loc[t];
        uint2 rg = s_rg[t];
        int s1 = (int)rg.x + (((int)(rg.y & 63u) + 3) & ~3);
        float4 pA = g.nb_pts[rg.x + q], pB = g.nb_pts[s1 + q];       // (an empty run reads record 0..3 of run 0: harmless, never used)
        const int rounds = (m + 63) >> 6;
        for (int r = 0; r < rounds; ++r) {
            const int tn = min(t + 64, m + 63);                    // (one idle round of zero records follows the chunk; nothing beyond it is touched)
            const float4 cn = s_loc[tn];
            const uint2 rgn = s_rg[tn];
            const int s1n = (int)rgn.x + (((int)(rgn.y & 63u) + 3) & ~3);
            const float4 pAn = g.nb_pts[rgn.x + q], pBn = g.nb_pts[s1n + q];
            const int item = __float_as_int(c.w);
            const int start = (int)rg.x, c0 = (int)(rg.y & 63u), tot = (int)((rg.y >> 6) & 2047u);
            QuadList kl;
            kl.init();
            auto run = [&](int lo, int hi, float4 p) {             // lo: a multiple of 4 entries; p: record lo + q, already loaded
                int j = lo + q;
                bool live = j < hi;
                while (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    const int jn = j + 4;
                    const float4 pn = g.nb_pts[jn < hi ? jn : lo];
                    const float xv = __fsub_rn(p.x, c.x), yv = __fsub_rn(p.y, c.y), zv = __fsub_rn(p.z, c.z);
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(xv, xv), __fmul_rn(yv, yv)), __fmul_rn(zv, zv));
                    const float v = (live && (radius2 == 0.f || d2 <= radius2)) ? d2 : inf;
                    kl.insert4(v, __float_as_int(p.w), 0x1111111111111111ull);
                    j = jn; p = pn; live = j < hi;
                }
            };
            if (layers > 0) run(start, start + c0, pA);
            // the list is full after shell 0 iff its last entry (lane 3's b) is finite -- reference: `if (kid >= K) break;`
            const bool full0 = quad_f<0xff>(kl.b) < inf;
            unsigned cells = (rg.y >> 22) & 1u, cand = (unsigned)c0;
            if (layers > 1 && !full0) {
                run(s1, s1 + tot - c0, pB);
                cells += (rg.y >> 17) & 31u; cand += (unsigned)(tot - c0);
            }
            if (item >= 0) {
                reinterpret_cast<int2 *>(pidx + (size_t)item * K)[q] = make_int2(kl.ia, kl.ib);
                int mine = (kl.a < inf ? 1 : 0) + (kl.b < inf ? 1 : 0);          // neighbours found: finite entries over the quad
                mine += quad_i<0xb1>(mine);                        // quad_perm [1,0,3,2]
                mine += quad_i<0x4e>(mine);                        // quad_perm [2,3,0,1]
                if (q == 0) {
                    n_cells += cells; n_cand += cand;
                    if (mine > 0) { ray_mask[item / SR] = 1; n_nb += (unsigned)mine; ++n_sv; }
                }
            }
            t = tn; c = cn; rg = rgn; s1 = s1n; pA = pAn; pB = pBn;
        }
        __syncthreads();
    }
    unsigned long long t_cells = n_cells, t_cand = n_cand, t_nb = n_nb, t_sv = n_sv;
    for (int o = 32; o > 0; o >>= 1) {
        t_cells += __shfl_xor(t_cells, o);
        t_cand += __shfl_xor(t_cand, o);
        t_nb += __shfl_xor(t_nb, o);
        t_sv += __shfl_xor(t_sv, o);
    }
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_st[wv][0] = t_cells; s_st[wv][1] = t_cand; s_st[wv][2] = t_nb; s_st[wv][3] = t_sv;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        block_stats[4 * (size_t)blockIdx.x + threadIdx.x] =
            s_st[0][threadIdx.x] + s_st[1][threadIdx.x] + s_st[2][threadIdx.x] + s_st[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void knn_finalize_kernel(const unsigned long long *__restrict__ block_stats, int nblocks,
                                                           unsigned long long *__restrict__ counts)
{
    __shared__ unsigned long long s_st[4][4];
    unsigned long long v[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += block_stats[4 * (size_t)b + k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) s_st[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[HNR_CNT_CELLS_VISITED] = s_st[0][0] + s_st[1][0] + s_st[2][0] + s_st[3][0];
        counts[HNR_CNT_CANDIDATES] = s_st[0][1] + s_st[1][1] + s_st[2][1] + s_st[3][1];
        counts[HNR_CNT_NEIGHBOURS] = s_st[0][2] + s_st[1][2] + s_st[2][2] + s_st[3][2];
        counts[HNR_CNT_SAMPLES_VALID] = s_st[0][3] + s_st[1][3] + s_st[2][3] + s_st[3][3];
    }
}

// ------------------------------------------------------------------------------------------------
// Second compaction (:705-709): two small kernels, deterministic, no host sync inside.
__global__ __launch_bounds__(1024) void compact_count_kernel(const int8_t *__restrict__ mask, int R, int32_t *block_sums)
{
    __shared__ int s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool m = i < R && mask[i] != 0;
    const unsigned long long b = __ballot(m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += s_w[k];
        block_sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void compact_plan_kernel(const int8_t *__restrict__ mask, int R,
                                                            const int32_t *__restrict__ block_sums, int nblocks,
                                                            int32_t *__restrict__ ray_row, unsigned long long *counts)
{
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    // block offset: sum of the preceding block sums (R/1024 is small: a frame has ~280 blocks)
    int part = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) part += block_sums[k];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const bool m = i < R && mask[i] != 0;
    const unsigned long long b = __ballot(m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += s_w[k];
        s_base = t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += s_w[k];
    if (i < R) ray_row[i] = m ? off + __popcll(b & ((1ull << (threadIdx.x & 63)) - 1ull)) : -1;
    if (blockIdx.x == (unsigned)nblocks - 1 && threadIdx.x == 0) {
        int t = s_base;
        for (int k = 0; k < 16; ++k) t += s_w[k];
        counts[HNR_CNT_RAYS_VALID] = (unsigned long long)t;
    }
}

// one wave per input ray: copy its rows to the compact position, expand the ray direction over SR (:91)
__global__ __launch_bounds__(256) void compact_rows_kernel(const int32_t *__restrict__ ray_row, int R, int SR, int K,
                                                           const int32_t *__restrict__ pidx, const float *__restrict__ loc,
                                                           const float *__restrict__ raydir,
                                                           const float *__restrict__ campos, const float *__restrict__ camrot,
                                                           int32_t *__restrict__ o_pidx, float *__restrict__ o_loc,
                                                           float *__restrict__ o_pers, float *__restrict__ o_dir)
{
    const int lane = threadIdx.x & 63;
    const int r = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    if (r >= R) return;
    const int row = ray_row[r];
    if (row < 0) return;
    const int nk = SR * K;
    for (int i = lane; i < nk; i += 64) o_pidx[(size_t)row * nk + i] = pidx[(size_t)r * nk + i];
    for (int i = lane; i < SR * 3; i += 64) {
        o_loc[(size_t)row * SR * 3 + i] = loc[(size_t)r * SR * 3 + i];
        o_dir[(size_t)row * SR * 3 + i] = raydir[3 * (size_t)r + (i % 3)];
    }
    // w2pers (:96-103): xyz_c[j] = sum_i (p[i] - campos[i]) * camrot[i][j]; padded slots included, like the reference
    for (int s = lane; s < SR; s += 64) {
        const float *p = loc + ((size_t)r * SR + s) * 3;
        const float s0 = __fsub_rn(p[0], campos[0]), s1 = __fsub_rn(p[1], campos[1]), s2 = __fsub_rn(p[2], campos[2]);
        float c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[j] = __fadd_rn(__fadd_rn(__fmul_rn(s0, camrot[j]), __fmul_rn(s1, camrot[3 + j])), __fmul_rn(s2, camrot[6 + j]));
        float *o = o_pers + ((size_t)row * SR + s) * 3;
        o[0] = hnr_div(c[0], c[2]); o[1] = hnr_div(c[1], c[2]); o[2] = c[2];
    }
}

}  // namespace hnr

using namespace hnr;

static int knn_blocks(int max_items)
{
    int blocks = cdiv(max_items, 256);
    const int cap = 256 * 8;   // 256 CUs x 8 blocks of 256 threads: grid-stride beyond that
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

template <int K>
static void launch_knn(const GridView &v, const int32_t *work, const float *loc, int SR, float r2, int layers,
                       int32_t *pidx, int8_t *mask, unsigned long long *counts, unsigned long long *block_stats,
                       int max_items, hipStream_t st)
{
    const int blocks = knn_blocks(max_items);
    knn_kernel<K><<<blocks, 256, 0, st>>>(v, work, loc, SR, r2, layers, pidx, mask, counts, block_stats);
    knn_finalize_kernel<<<1, 256, 0, st>>>(block_stats, blocks, counts);
}

extern "C" int hnr_march_query(const hnr_grid *g, const float *d_campos, const float *d_raydir, const float *d_tmid,
                               const hnr_query_params *q, int32_t *d_sample_pidx, float *d_sample_loc_w,
                               int32_t *d_ray_nsamp, int8_t *d_ray_mask, int32_t *d_work, int64_t *d_counts,
                               void *stream)
{
    if (!g || !q || !d_counts) { set_error("hnr_march_query: NULL argument"); return HNR_ERR_BADARG; }
    if (q->R > 0 && (!d_campos || !d_raydir || !d_tmid || !d_sample_pidx || !d_sample_loc_w || !d_ray_nsamp ||
                     !d_ray_mask || !d_work)) {
        set_error("hnr_march_query: NULL argument"); return HNR_ERR_BADARG;
    }
    if (q->R < 0 || q->D <= 0 || q->SR <= 0 || q->K <= 0 || q->K > HNR_MAX_K ||
        (q->tmid_stride != 0 && q->tmid_stride != q->D) || q->kernel_size[0] <= 0 || (q->knn_order != 0 && q->knn_order != 1)) {
        set_error("hnr_march_query: bad sizes (R=%d D=%d SR=%d K=%d tmid_stride=%d)", q->R, q->D, q->SR, q->K, q->tmid_stride);
        return HNR_ERR_BADARG;
    }
    if ((int64_t)q->R * q->SR * q->K >= (1ll << 31)) { set_error("hnr_march_query: R*SR*K overflows int32"); return HNR_ERR_TOOBIG; }
    hipStream_t st = (hipStream_t)stream;
    HNR_HIP_CHECK(hipMemsetAsync(d_counts, 0, sizeof(int64_t) * HNR_NCOUNTS, st));
    if (q->R == 0) return HNR_OK;
    const GridView v = g->view();
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d_counts);
    static int march_probe = -1;                                 // HNR_MARCH_PROBE=1: time the march kernel's fixed part alone (tools; results are empty)
    if (march_probe < 0) { const char *e = getenv("HNR_MARCH_PROBE"); march_probe = e ? atoi(e) : 0; }
    // a wave works through ~8 rays (wave launches, not instructions, bounded the one-ray-per-wave form: 0.63 waves per cycle over the whole chip); enough
    // workgroups that the last, partial round of residency is a few per cent of the launch (ONE workgroup per resident slot left a third of the CUs idle in a second round)
    static int rays_per_wave = -1;
    if (rays_per_wave < 0) { const char *e = getenv("HNR_MARCH_RAYS_PER_WAVE"); rays_per_wave = e ? atoi(e) : 8; if (rays_per_wave < 1) rays_per_wave = 1; }
    // (small batches: one ray per wave until there are ~32 waves per CU)
    int rpw = q->R / (device_num_cus() * 32);
    rpw = rpw < 1 ? 1 : (rpw > rays_per_wave ? rays_per_wave : rpw);
    int march_blocks = cdiv((int64_t)cdiv(q->R, rpw) * 64, 256);
    if (march_blocks < 1) march_blocks = 1;
    march_kernel<<<march_blocks, 256, 0, st>>>(v, d_campos, d_raydir, d_tmid, q->R, q->D, q->SR, q->K,
                                                                q->tmid_stride, q->pad_outputs, d_sample_pidx, d_sample_loc_w,
                                                                d_ray_nsamp, d_ray_mask, march_probe);
    HNR_LAUNCH_CHECK();
    const int layers = (q->kernel_size[0] + 1) / 2;
    const int max_items = q->R * q->SR;
    // scratch layout inside d_work: [R*SR work items | 2*nb block sums | pad to 8 B | 4*knn_blocks u64 stats]
    const int nb = cdiv(q->R, 1024);
    int32_t *block_sums = d_work + (size_t)max_items;
    size_t stats_off = ((size_t)max_items + 2 * (size_t)nb + 1) & ~(size_t)1;
    unsigned long long *block_stats = reinterpret_cast<unsigned long long *>(d_work + stats_off);
    nsamp_block_sum_kernel<<<nb, 1024, 0, st>>>(d_ray_nsamp, q->R, block_sums);
    worklist_kernel<<<nb, 1024, 0, st>>>(d_ray_nsamp, q->R, q->SR, block_sums, nb, d_work, cnt);
    HNR_LAUNCH_CHECK();
    // K = 8 with a 3x3x3 neighbourhood (every shipped config): the pipelined kernel (HNR_KNN=1: the generic one-cell-at-a-time kernel)
    static int knn_sel = -1;
    if (knn_sel < 0) { const char *e = getenv("HNR_KNN"); knn_sel = e ? atoi(e) : 8; }
    if (q->knn_order == 1 && !(q->K == 8 && layers <= 2)) {
        set_error("hnr_march_query: knn_order = 1 (canonical neighbour order) is built for K = 8 with a 3x3x3 neighbourhood (K=%d)", q->K);
        return HNR_ERR_BADARG;
    }
    hnr_grid_params gp;
    hnr_grid_stats gs;
    hnr_grid_get_params(g, &gp);
    hnr_grid_get_stats(g, &gs);
    const bool packable = gp.P <= 63 && gs.n_points < (1ll << 26);      // the pipelined kernel parks a cell's list as one word (start << 6 | count)
    if (q->knn_order == 1 && !packable) {
        set_error("hnr_march_query: knn_order = 1 needs P <= 63 and < 2^26 points (P=%d)", gp.P);
        return HNR_ERR_BADARG;
    }
    if (q->K == 8 && layers <= 2 && v.nb_pts && knn_sel != 1 && knn_sel != 3) {
        // the grid carries 3x3x3 neighbourhood lists (P <= 63): two lookups per sample, contiguous candidates (HNR_KNN=3: the 27-cell walk below)
        int blocks = knn_blocks(max_items);
        const int bin = knn_sel == 5 ? 0 : (knn_sel == 8 ? 2 : (knn_sel == 10 ? 3 : 1));
        {   // workgroups per CU of the persistent k-NN kernels (HNR_KNN_WG_PER_CU; 8 = what knn_blocks caps at).  Measured for the quad kernel: 6 (what its
            // 26 KB of LDS let a CU hold at once) 0.423 ms, 8 0.35 ms -- the workgroups that wait for a slot fill the tail of the first ones
            static int wg_per_cu = -1;
            if (wg_per_cu < 0) { const char *e = getenv("HNR_KNN_WG_PER_CU"); wg_per_cu = e ? atoi(e) : 8; if (wg_per_cu < 1) wg_per_cu = 1; }
            const int cap = device_num_cus() * wg_per_cu;
            if (knn_sel != 5) blocks = cdiv(max_items, 256) < cap ? cdiv(max_items, 256) : cap;
            if (blocks > knn_blocks(max_items)) blocks = knn_blocks(max_items);          // (the per-workgroup counters' scratch is sized for that many)
            if (blocks < 1) blocks = 1;
        }  // HNR_KNN=5: work-list order; 7: sorted by list length; 8: ... and by cell
#define HNR_NB_LAUNCH(S_, B_) knn_nb_kernel<8, S_, B_><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats)
        if (knn_sel == 9) {                                       // one lane per sample, sorted by list length, candidates fetched quad-cooperatively
            if (q->knn_order == 1) knn_nb_kernel<8, 1, 1, 1><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
            else knn_nb_kernel<8, 0, 1, 1><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
        } else
        // HNR_KNN: 8 (default) one lane per sample, samples sorted by list length and by a hash of their cell; 10 ... exactly by cell; 7 by list length only; 5 work-list
        // order; 9 candidates fetched quad-cooperatively through LDS; 4 quad-per-sample kernel (set-exact order only), 6 the same in work-list order; 3 / 1 the 27-cell walks.
        // Same-run query times on the bench frame (ms): 8 0.586 - 0.606 | 4 0.603 - 0.623 | 10 0.609 | 7 0.614 - 0.640 | 5 0.625 | 9 1.18 | 3 0.72 - 0.75
        if (q->knn_order == 1 && (knn_sel == 4 || knn_sel == 6)) {
            if (knn_sel == 4) knn_quad_kernel<1><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
            else knn_quad_kernel<0><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
        } else if (q->knn_order == 1) { if (bin == 3) HNR_NB_LAUNCH(1, 3); else if (bin == 2) HNR_NB_LAUNCH(1, 2); else if (bin) HNR_NB_LAUNCH(1, 1); else HNR_NB_LAUNCH(1, 0); }
        else { if (bin == 3) HNR_NB_LAUNCH(0, 3); else if (bin == 2) HNR_NB_LAUNCH(0, 2); else if (bin) HNR_NB_LAUNCH(0, 1); else HNR_NB_LAUNCH(0, 0); }
#undef HNR_NB_LAUNCH
        knn_finalize_kernel<<<1, 256, 0, st>>>(block_stats, blocks, cnt);
        HNR_LAUNCH_CHECK();
        return HNR_OK;
    }
    if (q->K == 8 && layers <= 2 && packable && (knn_sel != 1 || q->knn_order == 1)) {
        const int blocks = knn_blocks(max_items);
        static int probe_pass1 = -1;                             // HNR_KNN_PROBE_PASS1=1: time the cell lookups alone (tools/probe_query.py; results are empty)
        if (probe_pass1 < 0) { const char *e = getenv("HNR_KNN_PROBE_PASS1"); probe_pass1 = e ? atoi(e) : 0; }
        if (q->knn_order == 1)
            knn3_kernel<8, 1><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, probe_pass1 ? 0 : layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
        else
            knn3_kernel<8><<<blocks, 256, 0, st>>>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats);
        knn_finalize_kernel<<<1, 256, 0, st>>>(block_stats, blocks, cnt);
        HNR_LAUNCH_CHECK();
        return HNR_OK;
    }
    switch (q->K) {
#define HNR_KCASE(KK) case KK: launch_knn<KK>(v, d_work, d_sample_loc_w, q->SR, q->radius2, layers, d_sample_pidx, d_ray_mask, cnt, block_stats, max_items, st); break;
        // every K up to HNR_MAX_K (the reference compiles `#define KN <K>` into its kernel for any value, query_point_indices_worldcoords.py:110)
        HNR_KCASE(1) HNR_KCASE(2) HNR_KCASE(3) HNR_KCASE(4) HNR_KCASE(5) HNR_KCASE(6) HNR_KCASE(7) HNR_KCASE(8)
        HNR_KCASE(9) HNR_KCASE(10) HNR_KCASE(11) HNR_KCASE(12) HNR_KCASE(13) HNR_KCASE(14) HNR_KCASE(15) HNR_KCASE(16)
        HNR_KCASE(17) HNR_KCASE(18) HNR_KCASE(19) HNR_KCASE(20) HNR_KCASE(21) HNR_KCASE(22) HNR_KCASE(23) HNR_KCASE(24)
        HNR_KCASE(25) HNR_KCASE(26) HNR_KCASE(27) HNR_KCASE(28) HNR_KCASE(29) HNR_KCASE(30) HNR_KCASE(31) HNR_KCASE(32)
#undef HNR_KCASE
        default:
            set_error("hnr_march_query: K=%d exceeds HNR_MAX_K = %d", q->K, HNR_MAX_K);
            return HNR_ERR_BADARG;
    }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int64_t hnr_query_work_elems(int R, int SR)
{
    if (R < 0 || SR <= 0) return 0;
    const int64_t items = (int64_t)R * SR;
    const int64_t nb = (R + 1023) / 1024;
    return ((items + 2 * nb + 1) & ~(int64_t)1) + 2 * 4 * (int64_t)knn_blocks((int)(items < (1ll << 30) ? items : (1ll << 30))) + 2;
}

extern "C" int hnr_ray_compact_plan(const int8_t *d_ray_mask, int R, int32_t *d_ray_row, int32_t *d_scratch,
                                    int64_t *d_counts, void *stream)
{
    if (!d_counts || R < 0 || (R > 0 && (!d_ray_mask || !d_ray_row || !d_scratch))) {
        set_error("hnr_ray_compact_plan: bad argument"); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d_counts);
    if (R == 0) { HNR_HIP_CHECK(hipMemsetAsync(d_counts + HNR_CNT_RAYS_VALID, 0, 8, st)); return HNR_OK; }
    const int nb = cdiv(R, 1024);
    compact_count_kernel<<<nb, 1024, 0, st>>>(d_ray_mask, R, d_scratch);
    compact_plan_kernel<<<nb, 1024, 0, st>>>(d_ray_mask, R, d_scratch, nb, d_ray_row, cnt);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_ray_compact(const int32_t *d_ray_row, int R, int SR, int K, const int32_t *d_sample_pidx,
                               const float *d_sample_loc_w, const float *d_raydir, const float *d_campos,
                               const float *d_camrotc2w, int32_t *d_out_pidx, float *d_out_loc_w,
                               float *d_out_loc_pers, float *d_out_raydir, void *stream)
{
    if (R < 0 || SR <= 0 || K <= 0) { set_error("hnr_ray_compact: bad argument"); return HNR_ERR_BADARG; }
    if (R == 0) return HNR_OK;
    if (!d_ray_row || !d_sample_pidx || !d_sample_loc_w || !d_raydir || !d_campos || !d_camrotc2w) {
        set_error("hnr_ray_compact: NULL argument"); return HNR_ERR_BADARG;
    }
    if (!d_out_pidx || !d_out_loc_w || !d_out_loc_pers || !d_out_raydir) { set_error("hnr_ray_compact: NULL output"); return HNR_ERR_BADARG; }
    compact_rows_kernel<<<cdiv((int64_t)R * 64, 256), 256, 0, (hipStream_t)stream>>>(
        d_ray_row, R, SR, K, d_sample_pidx, d_sample_loc_w, d_raydir, d_campos, d_camrotc2w, d_out_pidx, d_out_loc_w,
        d_out_loc_pers, d_out_raydir);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
