// Gather + row assembly + K-sum + image-feature merge + final colour: everything of
// PointAggregator.forward / viewmlp (models/aggregators/point_aggregators.py:1427-1522, :892-1338)
// and NeuralPoints' gather (models/neural_points/neural_points.py:709-720) that is not a dense layer
// (those run in linear.hip).  Restated for the shipped configuration only:
// viewmlp, agg_intrp_order=2, linear kernel, agg_dist_pers=20, 3/5/4 PE freqs, hybrid image branch.
//
// Row order everywhere is the reference's boolean-mask order: (ray, slot, k) ascending.
#include "hnr_common.h"

namespace hnr {

// ------------------------------------------------------------------------------------------------
// Plan: exclusive scans over the work list of kept samples -> compact lists of VALID samples (>= 1
// neighbour) and of neighbour rows.  Two-level scan, no atomics, deterministic.
__device__ __forceinline__ int count_neighbours(const int32_t *__restrict__ p, int K)
{
    int n = 0;                                        // valid ids are a prefix (reference :494-496)
    while (n < K && p[n] >= 0) ++n;
    return n;
}

__global__ __launch_bounds__(1024) void plan_block_sum_kernel(const int32_t *__restrict__ work, const int32_t *__restrict__ pidx,
                                                              const unsigned long long *__restrict__ counts, int K,
                                                              int32_t *__restrict__ block_sums)
{
    __shared__ int s_a[16], s_b[16];
    const int n_items = (int)counts[HNR_CNT_SAMPLES];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int nb = i < n_items ? count_neighbours(pidx + (size_t)work[i] * K, K) : 0;
    int v = nb > 0;
    for (int o = 32; o > 0; o >>= 1) { nb += __shfl_xor(nb, o); v += __shfl_xor(v, o); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = v; s_b[threadIdx.x >> 6] = nb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; }
        block_sums[2 * blockIdx.x] = a;
        block_sums[2 * blockIdx.x + 1] = b;
    }
}

// vs_item[s] = ray*SR+slot of valid sample s; vs_off[s] = first neighbour row; vs_cnt[s] = #neighbours
__global__ __launch_bounds__(1024) void plan_scan_kernel(const int32_t *__restrict__ work, const int32_t *__restrict__ pidx,
                                                         const unsigned long long *__restrict__ counts, int K,
                                                         const int32_t *__restrict__ block_sums,
                                                         int32_t *__restrict__ vs_item, int32_t *__restrict__ vs_off,
                                                         int32_t *__restrict__ vs_cnt, int cap_samples, int cap_rows,
                                                         int32_t *__restrict__ overflow)
{
    __shared__ int s_a[16], s_b[16];
    __shared__ int s_base[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n_items = (int)counts[HNR_CNT_SAMPLES];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int pa = 0, pb = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) { pa += block_sums[2 * k]; pb += block_sums[2 * k + 1]; }
    for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o); pb += __shfl_xor(pb, o); }
    if (lane == 0) { s_a[wid] = pa; s_b[wid] = pb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; }
        s_base[0] = a; s_base[1] = b;
    }
    __syncthreads();
    int item = 0, nb = 0;
    if (i < n_items) { item = work[i]; nb = count_neighbours(pidx + (size_t)item * K, K); }
    const int v = nb > 0;
    int ia = v, ib = nb;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(ia, o), tb = __shfl_up(ib, o);
        if (lane >= o) { ia += ta; ib += tb; }
    }
    __syncthreads();
    if (lane == 63) { s_a[wid] = ia; s_b[wid] = ib; }
    __syncthreads();
    int oa = s_base[0] + ia - v, ob = s_base[1] + ib - nb;
    for (int k = 0; k < wid; ++k) { oa += s_a[k]; ob += s_b[k]; }
    if (v) {
        if (oa < cap_samples && ob + nb <= cap_rows) {
            vs_item[oa] = item; vs_off[oa] = ob; vs_cnt[oa] = nb;
        } else {
            *overflow = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gather + geometry + row assembly.  One 256-thread block handles 32 valid samples (up to 32*K rows):
//  phase 1: one lane per (sample, k): fetch the point record, w2pers, dists6, inverse-distance weight;
//  phase 2: the block writes the 284-wide block1 rows and the 7 extra block3 columns, one lane per
//           column so that every row is stored as contiguous 16-B-aligned bursts.
// K <= 8 (lanes 8 per sample); larger K loops.
struct GatherArgs {
    const float *xyz, *emb, *conf, *pdir, *color;       // point buffers [N,3] [N,F] [N] [N,3] [N,3]
    int F;                                               // embedding width (32)
    const int32_t *pidx;                                 // [R,SR,K]
    const float *loc_w;                                  // [R,SR,3]
    const float *raydir;                                 // [R,3]
    const float *campos, *camrot;                        // [3], [3,3]
    const int32_t *vs_item, *vs_off, *vs_cnt;
    const unsigned long long *counts;
    int SR, K;
    float *X1; int ld1;                                  // [rows, ld1]  block1 input (F + 6F + 60)
    float *X3; int ld3;                                  // [rows, ld3]  block3 input; cols 256..262 written here
    float *wagg;                                         // [rows] normalised weight * clamp(conf)
    float *weight_out, *conf_out;                        // optional [R,SR,K] (reference outputs), may be NULL
    int32_t *row_pid;                                    // SPLIT only: point id of every neighbour row
};

constexpr int G_SAMPLES = 32;      // valid samples per block
constexpr int G_RAW = 40;          // emb32 + dists6 (+2 pad)

__device__ __forceinline__ void w2pers(const float *p, const float *campos, const float *camrot, float out[3])
{
    // neural_points.py:607-613 / query_point_indices_worldcoords.py:96-103: c[j] = sum_i shift[i] * camrot[i][j]
    const float s0 = __fsub_rn(p[0], campos[0]), s1 = __fsub_rn(p[1], campos[1]), s2 = __fsub_rn(p[2], campos[2]);
    float c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
        c[j] = __fadd_rn(__fadd_rn(__fmul_rn(camrot[j], s0), __fmul_rn(camrot[3 + j], s1)), __fmul_rn(camrot[6 + j], s2));
    out[0] = hnr_div(c[0], c[2]); out[1] = hnr_div(c[1], c[2]); out[2] = c[2];
}

// SPLIT = 1: X1 rows hold only the 60 distance-encoding columns (the point-only 224 columns of block1's input are folded
// into a per-point table, see hnr_linear_f32_gather_add) and row_pid[row] names the point.
template <int F, int SPLIT>
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a)
{
    __shared__ float s_raw[G_SAMPLES * 8][G_RAW];     // per row: emb[F], dists[6]
    __shared__ float s_ext[G_SAMPLES * 8][8];         // per row: color3, dir-view3, dir.view, (pad)
    __shared__ int s_row[G_SAMPLES * 8];              // global row or -1
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const int s0 = blockIdx.x * G_SAMPLES;
    if (s0 >= n_valid) return;
    const int tid = threadIdx.x;
    const int K = a.K;
    const float cp[3] = {a.campos[0], a.campos[1], a.campos[2]};
    float cr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cr[i] = a.camrot[i];

    for (int kbase = 0; kbase < K; kbase += 8) {
        // ---- phase 1: lane (ls, kk) ----
        const int ls = tid >> 3, kk = kbase + (tid & 7);
        const int s = s0 + ls;
        int row = -1;
        float wraw = 0.f, confc = 0.f;
        int item = 0;
        if (s < n_valid) {
            item = a.vs_item[s];
            const int cnt = a.vs_cnt[s];
            if (kk < K && kk < cnt) row = a.vs_off[s] + kk;
        }
        if (row >= 0) {
            const int pid = a.pidx[(size_t)item * K + kk];
            const float *lw = a.loc_w + (size_t)item * 3;
            float sp[3], pp[3];
            w2pers(lw, cp, cr, sp);
            const float px = a.xyz[3 * (size_t)pid], py = a.xyz[3 * (size_t)pid + 1], pz = a.xyz[3 * (size_t)pid + 2];
            const float pw[3] = {px, py, pz};
            w2pers(pw, cp, cr, pp);
            float *raw = s_raw[tid];
            const float4 *e4 = reinterpret_cast<const float4 *>(a.emb + (size_t)pid * F);
#pragma unroll
            for (int i = 0; i < F / 4; ++i) {
                const float4 v = e4[i];
                raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
            }
            // dists (point_aggregators.py:1472-1480)
            const float dx = __fsub_rn(px, lw[0]), dy = __fsub_rn(py, lw[1]), dz = __fsub_rn(pz, lw[2]);
            raw[F + 0] = dx; raw[F + 1] = dy; raw[F + 2] = dz;
            raw[F + 3] = __fsub_rn(__fmul_rn(pp[0], pp[2]), __fmul_rn(sp[0], sp[2]));
            raw[F + 4] = __fsub_rn(__fmul_rn(pp[1], pp[2]), __fmul_rn(sp[1], sp[2]));
            raw[F + 5] = __fsub_rn(pp[2], sp[2]);
            // linear kernel (:825-833)
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            wraw = hnr_div(1.0f, fmaxf(nrm, 1e-6f));
            const float cf = a.conf[pid];
            confc = fminf(fmaxf(cf, 0.0001f), 1.0f);                 // gradiant_clamp forward value (:1422-1424)
            // block3 extras (:957-971): colour, dir - viewdir, dir . viewdir (viewdir = raw ray direction)
            const int ray = item / a.SR;
            const float vx = a.raydir[3 * (size_t)ray], vy = a.raydir[3 * (size_t)ray + 1], vz = a.raydir[3 * (size_t)ray + 2];
            const float ddx = a.pdir[3 * (size_t)pid], ddy = a.pdir[3 * (size_t)pid + 1], ddz = a.pdir[3 * (size_t)pid + 2];
            float *ext = s_ext[tid];
            ext[0] = a.color[3 * (size_t)pid]; ext[1] = a.color[3 * (size_t)pid + 1]; ext[2] = a.color[3 * (size_t)pid + 2];
            ext[3] = __fsub_rn(ddx, vx); ext[4] = __fsub_rn(ddy, vy); ext[5] = __fsub_rn(ddz, vz);
            ext[6] = __fadd_rn(__fadd_rn(__fmul_rn(ddx, vx), __fmul_rn(ddy, vy)), __fmul_rn(ddz, vz));
        }
        s_row[tid] = row;
        // normalise over the sample's K neighbours (:1500-1501).  With K <= 8 the 8 lanes of a sample hold all
        // of them; for K > 8 the sum is carried across kbase passes below.
        float sum = wraw;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        if (K > 8) {
            // second pass over the other chunks to complete the sum (rare configuration)
            float extra = 0.f;
            if (s < n_valid) {
                const int cnt = a.vs_cnt[s];
                for (int k2 = 0; k2 < cnt; ++k2) {
                    if (k2 >= kbase && k2 < kbase + 8) continue;
                    const int pid2 = a.pidx[(size_t)item * K + k2];
                    const float *lw = a.loc_w + (size_t)item * 3;
                    const float ex = __fsub_rn(a.xyz[3 * (size_t)pid2], lw[0]), ey = __fsub_rn(a.xyz[3 * (size_t)pid2 + 1], lw[1]),
                                ez = __fsub_rn(a.xyz[3 * (size_t)pid2 + 2], lw[2]);
                    const float n2 = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)));
                    extra += hnr_div(1.0f, fmaxf(n2, 1e-6f));
                }
            }
            sum += extra;
        }
        if (row >= 0) {
            const float w = hnr_div(wraw, fmaxf(sum, 1e-8f));
            a.wagg[row] = __fmul_rn(w, confc);
            if (SPLIT) a.row_pid[row] = a.pidx[(size_t)item * K + kk];
            if (a.weight_out) { a.weight_out[(size_t)item * K + kk] = w; a.conf_out[(size_t)item * K + kk] = confc; }
        }
        __syncthreads();
        // ---- phase 2: block-wide row assembly.  Work item = one 16-B chunk of the raw embedding, or one
        // (input, frequency) pair whose sin AND cos come from one sincosf and are stored as one 8-B pair
        // (positional_encoding interleaves [sin, cos] per (dim, freq), networks.py:182-189). ----
        constexpr int NPAIR = SPLIT ? 30 : 3 * F + 30;        // (96 embedding pairs +) 30 distance pairs
        constexpr int NITEM = SPLIT ? NPAIR : F / 4 + NPAIR;  // per row
        for (int it = tid; it < G_SAMPLES * 8 * NITEM; it += 256) {
            const int r = it / NITEM, w = it - r * NITEM;
            const int grow = s_row[r];
            if (grow < 0) continue;
            const float *raw = s_raw[r];
            float *o1 = a.X1 + (size_t)grow * a.ld1;
            if (SPLIT) {
                const int d = w / 5, f = w - 5 * d;
                const float x = __fmul_rn(raw[F + d], (float)(1 << f));
                float sv, cv;
                sincosf(x, &sv, &cv);
                *reinterpret_cast<float2 *>(o1 + 2 * w) = make_float2(sv, cv);
            } else if (w < F / 4) {
                reinterpret_cast<float4 *>(o1)[w] = make_float4(raw[4 * w], raw[4 * w + 1], raw[4 * w + 2], raw[4 * w + 3]);
            } else {
                const int p = w - F / 4;
                float x;
                int col;
                if (p < 3 * F) {
                    const int d = p / 3, f = p - 3 * d;
                    x = __fmul_rn(raw[d], (float)(1 << f));
                    col = F + 2 * p;
                } else {
                    const int q = p - 3 * F, d = q / 5, f = q - 5 * d;
                    x = __fmul_rn(raw[F + d], (float)(1 << f));
                    col = 7 * F + 2 * q;
                }
                float sv, cv;
                sincosf(x, &sv, &cv);
                *reinterpret_cast<float2 *>(o1 + col) = make_float2(sv, cv);
            }
        }
        for (int r = tid >> 3; r < G_SAMPLES * 8; r += 32) {
            const int grow = s_row[r];
            const int e = tid & 7;
            if (grow >= 0 && e < 7) a.X3[(size_t)grow * a.ld3 + 256 + e] = s_ext[r][e];
        }
        __syncthreads();
    }
}

// Per-point half of block1's input row: E[p, 0:224] = [emb32 | PE3(emb) 192] (point_aggregators.py:931-938), one 8-lane group
// of work items per point; E feeds one dense layer that yields the per-point addend table of hnr_linear_f32_gather_add.
template <int F>
__global__ __launch_bounds__(256) void point_rows_kernel(const float *__restrict__ emb, const int32_t *__restrict__ ids, int n,
                                                         float *__restrict__ E, int lde, const long long *__restrict__ d_n = nullptr)
{
    constexpr int NITEM = F / 4 + 3 * F;
    if (d_n && *d_n < n) n = (int)*d_n;                              // device-count form (csrc/render_train.hip)
    const int64_t total = (int64_t)n * NITEM;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(it / NITEM), w = (int)(it - (int64_t)p * NITEM);
        const float *e = emb + (size_t)(ids ? ids[p] : p) * F;      // ids: rows for a list of points (train: the touched ones)
        float *o = E + (size_t)p * lde;
        if (w < F / 4) {
            reinterpret_cast<float4 *>(o)[w] = reinterpret_cast<const float4 *>(e)[w];
        } else {
            const int q = w - F / 4, d = q / 3, f = q - 3 * d;
            float sv, cv;
            sincosf(__fmul_rn(e[d], (float)(1 << f)), &sv, &cv);
            *reinterpret_cast<float2 *>(o + F + 2 * q) = make_float2(sv, cv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-weighted sum (:1005-1026) + alpha branch + colour-feature-branch input (:1028-1036).
// One wave per valid sample; 64 lanes x float4 = one 256-wide row per load.
struct KsumArgs {
    const float *H4; int ldh;                         // [rows, ldh] block3 output (256 wide)
    const float *wagg;                                // [rows]
    const float *alpha_w, *alpha_b;                   // alpha_branch.0: [256], [1]
    const int32_t *vs_item, *vs_off, *vs_cnt;
    const float *raydir;                              // [R,3]
    const unsigned long long *counts;
    int SR;
    float *X5; int ld5;                               // [S_v, ld5]: feat256 | sin(viewdir 2^f) 12 | cos 12
    float *sigma;                                     // [S_v]
};

__device__ __forceinline__ float softplus_m1(float x)
{
    const float y = __fsub_rn(x, 1.0f);                // raw2out_density: softplus(x - 1), beta=1, threshold=20
    return y > 20.f ? y : log1pf(expf(y));
}

__global__ __launch_bounds__(256) void ksum_kernel(KsumArgs a)
{
    const int lane = threadIdx.x & 63;
    const int s = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    if (s >= n_valid) return;
    const int off = a.vs_off[s], cnt = a.vs_cnt[s];
    const float4 aw = reinterpret_cast<const float4 *>(a.alpha_w)[lane];
    const float ab = a.alpha_b[0];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float sig = 0.f;
    for (int k = 0; k < cnt; ++k) {
        const float4 h = reinterpret_cast<const float4 *>(a.H4 + (size_t)(off + k) * a.ldh)[lane];
        const float w = a.wagg[off + k];
        float d = h.x * aw.x + h.y * aw.y + h.z * aw.z + h.w * aw.w;
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        sig += softplus_m1(d + ab) * w;
        acc.x += h.x * w; acc.y += h.y * w; acc.z += h.z * w; acc.w += h.w * w;
    }
    float *o = a.X5 + (size_t)s * a.ld5;
    reinterpret_cast<float4 *>(o)[lane] = acc;
    if (lane < 24) {
        // positional_encoding(viewdirs, 4, ori=True)[3:] = [sin(d*F+f) x12 | cos x12] (:909-913)
        const int ray = a.vs_item[s] / a.SR;
        const int j = lane % 12, d = j >> 2, f = j & 3;
        const float x = __fmul_rn(a.raydir[3 * (size_t)ray + d], (float)(1 << f));
        o[256 + lane] = lane < 12 ? sinf(x) : cosf(x);
    }
    if (lane == 0) a.sigma[s] = sig;
}

// ------------------------------------------------------------------------------------------------
// Reference-view image features: 3x3 conv pyramid (:1047-1063), bilinear upsample + concat (:1064-1067)
// into a channels-last [V,H,W,48] map (45 used), pixel (0,0) zeroed (:1089).  Once per frame.
__global__ void conv3x3_lrelu_kernel(const float *__restrict__ in, int Cin, int Hin, int Win, int in_cl, int in_cstride,
                                     const float *__restrict__ w, const float *__restrict__ b, int Cout, int stride,
                                     int Hout, int Wout, float slope, float *__restrict__ out, int V)
{
    // in: NCHW planar [V,Cin,Hin,Win] (in_cl = 0) or channels-last [V,Hin,Win,in_cstride] (in_cl = 1); out: NCHW planar
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)V * Cout * Hout * Wout;
    if (idx >= total) return;
    const int ox = (int)(idx % Wout), oy = (int)((idx / Wout) % Hout), co = (int)((idx / ((int64_t)Wout * Hout)) % Cout),
              v = (int)(idx / ((int64_t)Wout * Hout * Cout));
    float acc = b[co];
    for (int ci = 0; ci < Cin; ++ci)
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - 1 + ky;
            if (iy < 0 || iy >= Hin) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - 1 + kx;
                if (ix < 0 || ix >= Win) continue;
                const float x = in_cl ? in[(((size_t)v * Hin + iy) * Win + ix) * in_cstride + ci]
                                      : in[(((size_t)v * Cin + ci) * Hin + iy) * Win + ix];
                acc = fmaf(x, w[((co * Cin + ci) * 3 + ky) * 3 + kx], acc);
            }
        }
    out[idx] = acc > 0.f ? acc : acc * slope;
}

__device__ __forceinline__ float bilinear_at(const float *__restrict__ p, int Hs, int Ws, int H, int W, int y, int x)
{
    // F.interpolate(mode='bilinear', align_corners=False): src = (dst + 0.5) * (in/out) - 0.5, clamped at 0
    const float sy = hnr_div((float)Hs, (float)H), sx = hnr_div((float)Ws, (float)W);
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    if (fy < 0.f) fy = 0.f;
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * p[(size_t)y0 * Ws + x0] + lx * p[(size_t)y0 * Ws + x1]) + ly * (hx * p[(size_t)y1 * Ws + x0] + lx * p[(size_t)y1 * Ws + x1]);
}

// bilinear taps of one (pixel, level): computed once, applied to all channels of the level (same formula and operation order as bilinear_at)
struct BilinearTap { size_t i00, i01, i10, i11; float hy, hx, ly, lx; };
__device__ __forceinline__ BilinearTap bilinear_tap(int Hs, int Ws, int H, int W, int y, int x)
{
    const float sy = hnr_div((float)Hs, (float)H), sx = hnr_div((float)Ws, (float)W);
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    if (fy < 0.f) fy = 0.f;
    if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    BilinearTap t;
    t.ly = fy - (float)y0; t.lx = fx - (float)x0; t.hy = 1.f - t.ly; t.hx = 1.f - t.lx;
    t.i00 = (size_t)y0 * Ws + x0; t.i01 = (size_t)y0 * Ws + x1; t.i10 = (size_t)y1 * Ws + x0; t.i11 = (size_t)y1 * Ws + x1;
    return t;
}
__device__ __forceinline__ float bilinear_apply(const float *__restrict__ p, const BilinearTap &t)
{
    return t.hy * (t.hx * p[t.i00] + t.lx * p[t.i01]) + t.ly * (t.hx * p[t.i10] + t.lx * p[t.i11]);
}

// four threads per pixel, 12 channels (three 16-B stores) each: [img 3 | s1 6 | s2 0..2], [s2 3..11 | s3 0..2], [s3 3..14], [s3 15..23 | pad 3];
// the taps of a level are computed once per thread (one thread per (pixel, channel) recomputed them 45 times per pixel: 0.43 ms per frame)
// One block = 256 pixels of ONE part (blockIdx.y): the part decides which levels a thread samples, and with the four parts of a pixel in adjacent lanes
// every wave ran all four variants under exec masks.
template <int PART>
__device__ __forceinline__ void featmap_part(const float *__restrict__ img, const float *__restrict__ s1, const float *__restrict__ s2,
                                             const float *__restrict__ s3, int V, int H, int W, int H1, int W1, int H2, int W2, int H3, int W3,
                                             float *__restrict__ fm, int64_t pix)
{
    const int x = (int)(pix % W), y = (int)((pix / W) % H), v = (int)(pix / ((int64_t)W * H));
    float o[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = 0.f;
    if (!(x == 0 && y == 0)) {                     // aux_feature_output[:, :, 0, 0] *= 0  (:1089)
        constexpr bool need1 = PART == 0, need2 = PART <= 1, need3 = PART >= 1;
        BilinearTap t1 = {}, t2 = {}, t3 = {};
        if (need1) t1 = bilinear_tap(H1, W1, H, W, y, x);
        if (need2) t2 = bilinear_tap(H2, W2, H, W, y, x);
        if (need3) t3 = bilinear_tap(H3, W3, H, W, y, x);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            constexpr int c0 = 12 * PART;
            const int c = c0 + i;
            if (c < 3) o[i] = img[pix * 3 + c];
            else if (c < 9) o[i] = bilinear_apply(s1 + ((size_t)v * 6 + (c - 3)) * H1 * W1, t1);
            else if (c < 21) o[i] = bilinear_apply(s2 + ((size_t)v * 12 + (c - 9)) * H2 * W2, t2);
            else if (c < 45) o[i] = bilinear_apply(s3 + ((size_t)v * 24 + (c - 21)) * H3 * W3, t3);
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(fm + pix * 48 + 12 * PART);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]); dst[1] = make_float4(o[4], o[5], o[6], o[7]); dst[2] = make_float4(o[8], o[9], o[10], o[11]);
}

__global__ __launch_bounds__(256) void featmap_kernel(const float *__restrict__ img /*[V,H,W,3]*/, const float *__restrict__ s1, const float *__restrict__ s2,
                                                      const float *__restrict__ s3, int V, int H, int W, int H1, int W1, int H2, int W2, int H3, int W3,
                                                      float *__restrict__ fm /*[V,H,W,48]*/)
{
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (int64_t)V * H * W) return;
    switch (blockIdx.y) {
        case 0: featmap_part<0>(img, s1, s2, s3, V, H, W, H1, W1, H2, W2, H3, W3, fm, pix); break;
        case 1: featmap_part<1>(img, s1, s2, s3, V, H, W, H1, W1, H2, W2, H3, W3, fm, pix); break;
        case 2: featmap_part<2>(img, s1, s2, s3, V, H, W, H1, W1, H2, W2, H3, W3, fm, pix); break;
        default: featmap_part<3>(img, s1, s2, s3, V, H, W, H1, W1, H2, W2, H3, W3, fm, pix); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Merge-weight rows (:1074-1096, :1188-1199): 16 lanes per (view, valid sample): reprojection (w2iproject,
// neural_points_volumetric_model.py:248-255), truncation to a pixel, 45-float gather, delta view direction
// (:296-310), colour feature copy.  Row = [imgfeat45 | colfeat128 | delta_dir3] (176), lda = 176.
struct ProjArgs {
    const float *loc_w;                                 // [R,SR,3]
    const int32_t *vs_item;
    const unsigned long long *counts;
    const float *w2c;                                   // [V,4,4] = inverse(c2w_nearest[v]) row-major
    const float *Kmat;                                  // [3,3] intrinsic_nearest
    const float *campos, *campos_n;                     // [3], [V,3]
    const float *fm; int H, W;                          // [V,H,W,48]
    const float *CF; int ldcf;                          // [S_v, ldcf] colour feature (128)
    int V, cap;                                         // cap = row capacity per view (rows are v*cap + s)
    float *X6; int ld6;                                 // [V*cap, ld6]
    float *vmask;                                       // [V*cap]
    int32_t *row_sample;                                // SPLIT: [V*cap] sample index of the row (CF is then not copied)
};

__global__ __launch_bounds__(256) void proj_rows_kernel(ProjArgs a)
{
    // 16 lanes per (view, sample) row, 3 feature channels per lane: 4 rows per wave keep four dependent
    // index -> position -> pixel -> feature load chains in flight (one row per wave was latency-bound: 5.0 ms per frame)
    const int sub = threadIdx.x & 15;
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    // grid-stride over the V * n_valid rows: the grid is sized from a CAPACITY (the count lives on the device), a fixed number of
    // blocks walking the real rows costs nothing when the capacity is generous
    for (int64_t rowi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; rowi < (int64_t)a.V * n_valid; rowi += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const int v = (int)(rowi / n_valid);
    const int s = (int)(rowi - (int64_t)v * n_valid);
    const float *p = a.loc_w + (size_t)a.vs_item[s] * 3;
    const float x = p[0], y = p[1], z = p[2];
    int px, py;
    const bool inval = hnr_project_pixel(x, y, z, a.w2c + 16 * v, a.Kmat, a.W, a.H, px, py);
    const size_t row = (size_t)v * a.cap + s;
    float *o = a.X6 + row * a.ld6;
    const float *f = a.fm + (((size_t)v * a.H + py) * a.W + px) * 48;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ch = sub + 16 * q;
        if (ch < 45) o[ch] = f[ch];
    }
    const int dcol = a.row_sample ? 45 : 173;          // SPLIT rows are [imgfeat45 | ddir3]
    if (!a.row_sample) {
        // colour feature: 128 floats, 8 per lane
        const float4 *cf = reinterpret_cast<const float4 *>(a.CF + (size_t)s * a.ldcf);
        const float4 c0 = cf[2 * sub], c1 = cf[2 * sub + 1];
        float *oc = o + 45 + 8 * sub;
        oc[0] = c0.x; oc[1] = c0.y; oc[2] = c0.z; oc[3] = c0.w; oc[4] = c1.x; oc[5] = c1.y; oc[6] = c1.z; oc[7] = c1.w;
    } else if (sub == 15) {
        a.row_sample[row] = s;
    }
    if (sub < 3) {
        // delta view direction (:298-305)
        const float cx = x - a.campos[0], cy = y - a.campos[1], cz = z - a.campos[2];
        const float cn = sqrtf(cx * cx + cy * cy + cz * cz) + 1e-6f;
        const float nx = x - a.campos_n[3 * v], ny = y - a.campos_n[3 * v + 1], nz = z - a.campos_n[3 * v + 2];
        const float nn = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f;
        const float cur = hnr_div(sub == 0 ? cx : sub == 1 ? cy : cz, cn);
        const float nea = hnr_div(sub == 0 ? nx : sub == 1 ? ny : nz, nn);
        o[dcol + sub] = nea - cur;
    }
    if (sub == 0) a.vmask[row] = inval ? 0.f : 1.f;
    }
}

// Probe: the pixel every (view, valid sample) row of the merge stage gathers -- the same hnr_project_pixel the merge kernels call.
__global__ __launch_bounds__(256) void proj_pixels_kernel(const float *loc_w, const int32_t *vs_item, const unsigned long long *counts, const float *w2c, const float *Kmat,
                                                          int V, int H, int W, int cap, int32_t *pix)
{
    const int n_valid = (int)counts[HNR_CNT_SAMPLES_VALID] < cap ? (int)counts[HNR_CNT_SAMPLES_VALID] : cap;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t)V * n_valid; t += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)(t / n_valid), s = (int)(t - (int64_t)v * n_valid);
        const float *p = loc_w + (size_t)vs_item[s] * 3;
        int px, py;
        const bool inval = hnr_project_pixel(p[0], p[1], p[2], w2c + 16 * v, Kmat, W, H, px, py);
        int32_t *o = pix + ((size_t)v * cap + s) * 2;
        o[0] = inval ? -1 : px; o[1] = inval ? -1 : py;
    }
}

// Merge (:1199-1217) + mix-up input (:1286-1292): one wave per valid sample.
struct MergeArgs {
    const float *X6; int ld6;                            // rows v*cap+s: [imgfeat45 | ...]
    const float *Hm; int ldh;                            // [V*cap, ldh] last hidden layer of aux_merge_weight_block (64)
    const float *w_last, *b_last;                        // aux_merge_weight_block.6: [64], [1]
    const float *vmask; const float *frame_w;            // [V*cap]; optional [V] (downweight_blurry_feats) or NULL
    const float *CF; int ldcf;
    const unsigned long long *counts;
    int V, cap;
    float *X7; int ld7;                                  // [S_v, ld7]: colfeat[:45] | merged45
    const uint8_t *ray_drop; const int32_t *vs_item; int SR;   // train-time patch drop (:1222-1237): merged = 0 on flagged rays
};

__global__ __launch_bounds__(256) void merge_kernel(MergeArgs a)
{
    const int lane = threadIdx.x & 63;
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const float wl = a.w_last[lane];
    for (int s = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6); s < n_valid; s += (int)((gridDim.x * (unsigned)blockDim.x) >> 6)) {
    float fsum = 0.f, wsum = 0.f;
    constexpr int VB = 4;                                  // views fetched together (independent loads in flight)
    for (int v0 = 0; v0 < a.V; v0 += VB) {
        float hm[VB], f[VB], vm[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int v = v0 + u < a.V ? v0 + u : a.V - 1;
            const size_t row = (size_t)v * a.cap + s;
            hm[u] = a.Hm[row * a.ldh + lane];
            f[u] = lane < 45 ? a.X6[row * a.ld6 + lane] : 0.f;
            vm[u] = a.vmask[row];
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            if (v0 + u >= a.V) break;
            float d = hm[u] * wl;
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            float wv = hnr_div(1.f, 1.f + expf(-(d + a.b_last[0])));
            wv *= vm[u];
            if (a.frame_w) wv *= a.frame_w[v0 + u];
            fsum += f[u] * wv;
            wsum += wv;
        }
    }
    float *o = a.X7 + (size_t)s * a.ld7;
    if (lane < 45) {
        o[lane] = a.CF[(size_t)s * a.ldcf + lane];
        const bool drop = a.ray_drop && a.ray_drop[a.vs_item[s] / a.SR];
        o[45 + lane] = drop ? 0.f : hnr_div(fsum, wsum + 1e-6f);
    }
    }
}

// Final colour (:1293-1295, :1334, :478-482) and scatter into decoded [R,SR,4] (:1337-1338). One wave per sample.
struct FinalArgs {
    const float *Y; int ldy;                             // [S_v, ldy] color_mixup_block output (45)
    const float *CF; int ldcf;
    const float *w_fin, *b_fin;                          // color_final_block.0: [3,128], [3]
    const float *sigma;
    const int32_t *vs_item;
    const unsigned long long *counts;
    float *decoded;                                      // [R*SR, 4], pre-zeroed
};

// 16 lanes per sample (8 columns each, two 16-B loads of the colour feature per lane), four samples per wave; the three dot products are
// reduced over the 16 lanes by DPP (a wave per sample with six ds_bpermute steps per output was latency-bound: 0.90 ms per frame).
__global__ __launch_bounds__(256) void final_color_kernel(FinalArgs a)
{
    const int lane = threadIdx.x & 63, l16 = lane & 15, sub = lane >> 4;
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    float w[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[j][e] = a.w_fin[j * 128 + 8 * l16 + e];
    const float b0 = a.b_fin[0], b1 = a.b_fin[1], b2 = a.b_fin[2];
    const bool yvec = a.ldy >= 48 && (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Y) & 15) == 0;
    const int wave_g = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * (unsigned)blockDim.x) >> 6);
    for (int s0 = 4 * wave_g; s0 < n_valid; s0 += 4 * n_waves) {
        const int s = s0 + sub;
        const bool ok = s < n_valid;
        const size_t sc = ok ? (size_t)s : (size_t)(n_valid - 1);
        const float4 c0 = *reinterpret_cast<const float4 *>(a.CF + sc * a.ldcf + 8 * l16), c1 = *reinterpret_cast<const float4 *>(a.CF + sc * a.ldcf + 8 * l16 + 4);
        float x[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (yvec) {                                                         // mix-up rows padded to 48 columns: two 16-B loads in the six lanes that hold them
            if (l16 < 6) {
                const float4 y0 = *reinterpret_cast<const float4 *>(a.Y + sc * a.ldy + 8 * l16), y1 = *reinterpret_cast<const float4 *>(a.Y + sc * a.ldy + 8 * l16 + 4);
                const float y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) if (8 * l16 + e < 45) x[e] = y[e] + x[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 8 * l16 + e;
                if (c < 45) x[e] = a.Y[sc * a.ldy + c] + x[e];              // learn_residuals (:1294): mix-up output + colour feature, same operand order as before
            }
        }
        float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[j] += x[e] * w[j][e];
            r[j] += __builtin_amdgcn_update_dpp(0.f, r[j], 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
            r[j] += __builtin_amdgcn_update_dpp(0.f, r[j], 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
            r[j] += __builtin_amdgcn_update_dpp(0.f, r[j], 0x141, 0xf, 0xf, false);     // row_half_mirror
            r[j] += __builtin_amdgcn_update_dpp(0.f, r[j], 0x140, 0xf, 0xf, false);     // row_mirror
        }
        if (l16 == 0 && ok) {
            float4 out;
            out.x = a.sigma[s];
            const float bb[3] = {b0, b1, b2};
            float rgb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float sg = hnr_div(1.f, 1.f + expf(-(r[j] + bb[j])));
                rgb[j] = sg * (1.f + 2.f * 0.001f) - 0.001f;
            }
            out.y = rgb[0]; out.z = rgb[1]; out.w = rgb[2];
            reinterpret_cast<float4 *>(a.decoded)[a.vs_item[s]] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Composite: ray_dist (neural_points_volumetric_model.py:331-339) + ray_march (diff_ray_marching.py:508-557)
// + fill_invalid (:87-126).  One lane per ray, serial over SR (front to back).
struct CompositeArgs {
    const float *decoded;                                // [R,SR,4]
    const float *loc_w;                                  // [R,SR,3] (zero padded)
    const int32_t *pidx;                                 // [R,SR,K]
    const int8_t *ray_mask;                              // [R]
    const int32_t *nsamp;                                // [R] or NULL (padded inputs)
    const float *campos, *camrot, *bg;                   // [3], [3,3], [3]
    int R, SR, K;
    float vsize_z;
    int unit_mode;
    float *raycolor;                                     // [R,3]   (background colour where ray_mask == 0)
    float *opacity;                                      // [R,SR]  (0 where ray_mask == 0)
    float *is_bg;                                        // [R]     (1 where ray_mask == 0)
    float *blend_w;                                      // [R,SR] optional
};

__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    float *op = a.opacity + (size_t)r * a.SR;
    if (!a.ray_mask[r]) {
        a.raycolor[3 * (size_t)r] = bg0; a.raycolor[3 * (size_t)r + 1] = bg1; a.raycolor[3 * (size_t)r + 2] = bg2;
        a.is_bg[r] = 1.f;
        for (int s = 0; s < a.SR; ++s) { op[s] = 0.f; if (a.blend_w) a.blend_w[(size_t)r * a.SR + s] = 0.f; }
        return;
    }
    float cr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cr[i] = a.camrot[i];
    const float cp[3] = {a.campos[0], a.campos[1], a.campos[2]};
    const int ns = a.nsamp ? a.nsamp[r] : a.SR;          // slots >= ns hold the padding values (position 0, no neighbour)
    auto zc = [&](int s) {
        const float *p = a.loc_w + ((size_t)r * a.SR + s) * 3;
        const float q0 = s < ns ? p[0] : 0.f, q1 = s < ns ? p[1] : 0.f, q2 = s < ns ? p[2] : 0.f;
        const float s0 = __fsub_rn(q0, cp[0]), s1 = __fsub_rn(q1, cp[1]), s2 = __fsub_rn(q2, cp[2]);
        return __fadd_rn(__fadd_rn(__fmul_rn(cr[2], s0), __fmul_rn(cr[5], s1)), __fmul_rn(cr[8], s2));
    };
    float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    float zmax = zc(0);
    // slots >= ns hold no sample: sigma = 0 there, so opacity and blend weight are exactly 0 and T is multiplied by fl(1 + 1e-10) = 1 -- they are
    // written as zeros without the loads and the exponential (11.9 of 24 slots are kept on the bench frame)
    const int s_end = ns < a.SR ? ns : a.SR;
    for (int s = s_end; s < a.SR; ++s) { op[s] = 0.f; if (a.blend_w) a.blend_w[(size_t)r * a.SR + s] = 0.f; }
    for (int s = 0; s < s_end; ++s) {
        float dist;
        if (s + 1 < a.SR) {
            const float zn = fmaxf(zmax, zc(s + 1));       // cummax
            dist = __fsub_rn(zn, zmax);
            zmax = zn;
        } else {
            dist = a.vsize_z;
        }
        if (dist < 1e-8f || (a.unit_mode && dist > 2.f * a.vsize_z)) dist = a.vsize_z;
        const bool valid = s < ns && a.pidx[((size_t)r * a.SR + s) * a.K] >= 0;   // ray_valid = any(mask over K); ids are a prefix
        const float4 d = reinterpret_cast<const float4 *>(a.decoded)[(size_t)r * a.SR + s];
        const float sigma = valid ? d.x : 0.f;
        const float rd = valid ? dist : 0.f;
        const float o = 1.f - expf(-sigma * rd);
        const float bw = o * T;
        c0 += d.y * bw; c1 += d.z * bw; c2 += d.w * bw;
        op[s] = o;
        if (a.blend_w) a.blend_w[(size_t)r * a.SR + s] = bw;
        T *= (1.f - o + 1e-10f);
    }
    a.raycolor[3 * (size_t)r] = c0 + bg0 * T;
    a.raycolor[3 * (size_t)r + 1] = c1 + bg1 * T;
    a.raycolor[3 * (size_t)r + 2] = c2 + bg2 * T;
    a.is_bg[r] = T;
}

// ------------------------------------------------------------------------------------------------
// ray_march as a stand-alone operator (diff_ray_marching.py:508-557) for callers that already hold ray_dist / ray_valid /
// decoded features (the drop-in `ray_march` function); the fused path uses composite_kernel.  One lane per ray.
__global__ __launch_bounds__(256) void ray_march_kernel(const float *__restrict__ ray_dist, const uint8_t *__restrict__ ray_valid,
                                                        const float *__restrict__ feat /*[R,SR,4]*/, const float *__restrict__ bg /*[3] or NULL*/,
                                                        int R, int SR, float *__restrict__ ray_color, float *__restrict__ opacity,
                                                        float *__restrict__ acc_t, float *__restrict__ blend_w, float *__restrict__ bg_t)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float T = 1.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int s = 0; s < SR; ++s) {
        const size_t i = (size_t)r * SR + s;
        const float4 d = reinterpret_cast<const float4 *>(feat)[i];
        const float sigma = ray_valid[i] ? d.x : 0.f;
        const float o = 1.f - expf(-sigma * ray_dist[i]);
        const float w = o * T;
        opacity[i] = o; acc_t[i] = T; blend_w[i] = w;
        c0 += d.y * w; c1 += d.z * w; c2 += d.w * w;
        T *= (1.f - o + 1e-10f);
    }
    if (bg) { c0 += bg[0] * T; c1 += bg[1] * T; c2 += bg[2] * T; }
    ray_color[3 * (size_t)r] = c0; ray_color[3 * (size_t)r + 1] = c1; ray_color[3 * (size_t)r + 2] = c2;
    bg_t[r] = T;
}

// ------------------------------------------------------------------------------------------------
// Hole-probing outputs (opt.prob == 1; models/neural_points_volumetric_model.py:392-416), read by the shell's point-growing
// step (run/train_ft.py:450-569): per ray the shading sample of maximum opacity, its position, the distance to its nearest
// listed neighbour and the weight * confidence averages of its neighbours' attributes.  8 lanes per ray.
struct ProbeArgs {
    const float *opacity, *loc_w;                        // [R,SR], [R,SR,3]
    const int32_t *pidx;                                 // [R,SR,K]
    const float *weight, *conf_c;                        // [R,SR,K] normalised weights, clamped confidences
    const float *xyz, *emb, *conf, *pdir, *color;        // point buffers
    int F, R, SR, K;
    float *o_opacity, *o_loc, *o_far, *o_color, *o_dir, *o_conf, *o_emb;   // [R], [R,3], [R], [R,3], [R,3], [R], [R,F]
};

__global__ __launch_bounds__(256) void probe_kernel(ProbeArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(t >> 3), sub = (int)(t & 7);
    if (r >= a.R) return;
    // argmax over the ray's samples (first maximum, like torch.max on CPU)
    int best = 0;
    float bo = a.opacity[(size_t)r * a.SR];
    for (int s = 1; s < a.SR; ++s) {
        const float o = a.opacity[(size_t)r * a.SR + s];
        if (o > bo) { bo = o; best = s; }
    }
    const size_t item = (size_t)r * a.SR + best;
    const float lx = a.loc_w[3 * item], ly = a.loc_w[3 * item + 1], lz = a.loc_w[3 * item + 2];
    float far = INFINITY, acc_c[3] = {0.f, 0.f, 0.f}, acc_d[3] = {0.f, 0.f, 0.f}, acc_f = 0.f;
    float acc_e[4] = {0.f, 0.f, 0.f, 0.f};              // this lane's embedding columns sub, sub+8, sub+16, sub+24 (F <= 32)
    for (int k = 0; k < a.K; ++k) {
        const int raw = a.pidx[item * a.K + k];
        const int pid = raw < 0 ? 0 : raw;                // empty slots read point 0 (index clamp, neural_points.py:711); their weight is 0
        const float w = a.weight[item * a.K + k] * a.conf_c[item * a.K + k];
        const float dx = a.xyz[3 * (size_t)pid] - lx, dy = a.xyz[3 * (size_t)pid + 1] - ly, dz = a.xyz[3 * (size_t)pid + 2] - lz;
        far = fminf(far, sqrtf(dx * dx + dy * dy + dz * dz));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc_c[j] += a.color[3 * (size_t)pid + j] * w;
            acc_d[j] += a.pdir[3 * (size_t)pid + j] * w;
        }
        acc_f += a.conf[pid] * w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = sub + 8 * q;
            if (c < a.F) acc_e[q] += a.emb[(size_t)pid * a.F + c] * w;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = sub + 8 * q;
        if (c < a.F) a.o_emb[(size_t)r * a.F + c] = acc_e[q];
    }
    if (sub == 0) {
        a.o_opacity[r] = bo; a.o_far[r] = far; a.o_conf[r] = acc_f;
        a.o_loc[3 * (size_t)r] = lx; a.o_loc[3 * (size_t)r + 1] = ly; a.o_loc[3 * (size_t)r + 2] = lz;
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.o_color[3 * (size_t)r + j] = acc_c[j]; a.o_dir[3 * (size_t)r + j] = acc_d[j]; }
    }
}

// ------------------------------------------------------------------------------------------------
// The materialised gather of NeuralPoints.forward (neural_points.py:709-720) for the drop-in 14-tuple API
// only (the fused path never materialises it).  One 8-lane group per (ray, slot, k) entry; empty entries
// read point 0 like the reference's clamp(min=0).
__global__ __launch_bounds__(256) void gather_points_kernel(const int32_t *__restrict__ pidx, int64_t n, const float *__restrict__ xyz,
                                                            const float *__restrict__ emb, const float *__restrict__ conf,
                                                            const float *__restrict__ pdir, const float *__restrict__ color, int F,
                                                            const float *__restrict__ campos, const float *__restrict__ camrot,
                                                            float *__restrict__ o_color, float *__restrict__ o_dir,
                                                            float *__restrict__ o_conf, float *__restrict__ o_emb,
                                                            float *__restrict__ o_pers, float *__restrict__ o_xyz, uint8_t *__restrict__ o_mask)
{
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    if (e >= n) return;
    const int raw = pidx[e];
    const int pid = raw < 0 ? 0 : raw;
    for (int c = sub; c < F; c += 8) o_emb[e * F + c] = emb[(size_t)pid * F + c];
    if (sub < 3) {
        o_color[e * 3 + sub] = color[3 * (size_t)pid + sub];
        o_dir[e * 3 + sub] = pdir[3 * (size_t)pid + sub];
        o_xyz[e * 3 + sub] = xyz[3 * (size_t)pid + sub];
    }
    if (sub == 3) {
        const float p[3] = {xyz[3 * (size_t)pid], xyz[3 * (size_t)pid + 1], xyz[3 * (size_t)pid + 2]};
        float pp[3];
        w2pers(p, campos, camrot, pp);
        o_pers[e * 3] = pp[0]; o_pers[e * 3 + 1] = pp[1]; o_pers[e * 3 + 2] = pp[2];
    }
    if (sub == 4) { o_conf[e] = conf[pid]; o_mask[e] = raw >= 0 ? 1 : 0; }
}

}  // namespace hnr

using namespace hnr;

// ================================================================================== C ABI
extern "C" int hnr_gather_points(const int32_t *d_sample_pidx, int64_t n_entries, const float *d_xyz, const float *d_emb,
                                 const float *d_conf, const float *d_dir, const float *d_color, int F, const float *d_campos,
                                 const float *d_camrot, float *d_o_color, float *d_o_dir, float *d_o_conf, float *d_o_emb,
                                 float *d_o_xyz_pers, float *d_o_xyz, uint8_t *d_o_mask, void *stream)
{
    if (n_entries < 0 || F <= 0) { set_error("hnr_gather_points: bad sizes"); return HNR_ERR_BADARG; }
    if (n_entries == 0) return HNR_OK;
    if (!d_sample_pidx || !d_xyz || !d_emb || !d_conf || !d_dir || !d_color || !d_campos || !d_camrot || !d_o_color || !d_o_dir ||
        !d_o_conf || !d_o_emb || !d_o_xyz_pers || !d_o_xyz || !d_o_mask) {
        set_error("hnr_gather_points: NULL argument"); return HNR_ERR_BADARG;
    }
    gather_points_kernel<<<cdiv(n_entries * 8, 256), 256, 0, (hipStream_t)stream>>>(d_sample_pidx, n_entries, d_xyz, d_emb, d_conf, d_dir,
                                                                                   d_color, F, d_campos, d_camrot, d_o_color, d_o_dir,
                                                                                   d_o_conf, d_o_emb, d_o_xyz_pers, d_o_xyz, d_o_mask);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_sample_plan(const int32_t *d_work, const int32_t *d_sample_pidx, const int64_t *d_counts, int K,
                               int max_items, int32_t *d_vs_item, int32_t *d_vs_off, int32_t *d_vs_cnt,
                               int cap_samples, int cap_rows, int32_t *d_scratch, int32_t *d_overflow, void *stream)
{
    if (!d_work || !d_sample_pidx || !d_counts || !d_vs_item || !d_vs_off || !d_vs_cnt || !d_scratch || !d_overflow ||
        K <= 0 || max_items < 0) {
        set_error("hnr_sample_plan: bad argument"); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    HNR_HIP_CHECK(hipMemsetAsync(d_overflow, 0, 4, st));
    if (max_items == 0) return HNR_OK;
    const int nb = cdiv(max_items, 1024);
    const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(d_counts);
    plan_block_sum_kernel<<<nb, 1024, 0, st>>>(d_work, d_sample_pidx, cnt, K, d_scratch);
    plan_scan_kernel<<<nb, 1024, 0, st>>>(d_work, d_sample_pidx, cnt, K, d_scratch, d_vs_item, d_vs_off, d_vs_cnt,
                                          cap_samples, cap_rows, d_overflow);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_gather_rows(const float *d_xyz, const float *d_emb, const float *d_conf, const float *d_dir,
                               const float *d_color, int F, const int32_t *d_sample_pidx, const float *d_sample_loc_w,
                               const float *d_raydir, const float *d_campos, const float *d_camrot,
                               const int32_t *d_vs_item, const int32_t *d_vs_off, const int32_t *d_vs_cnt,
                               const int64_t *d_counts, int SR, int K, int cap_samples,
                               float *d_X1, int ld1, float *d_X3, int ld3, float *d_wagg,
                               float *d_weight_out, float *d_conf_out, int32_t *d_row_pid, void *stream)
{
    const bool split = d_row_pid != nullptr;
    if (!d_xyz || !d_emb || !d_conf || !d_dir || !d_color || !d_sample_pidx || !d_sample_loc_w || !d_raydir || !d_campos ||
        !d_camrot || !d_vs_item || !d_vs_off || !d_vs_cnt || !d_counts || !d_X1 || !d_X3 || !d_wagg) {
        set_error("hnr_gather_rows: NULL argument"); return HNR_ERR_BADARG;
    }
    if (F != 32) { set_error("hnr_gather_rows: point_features_dim=%d unsupported (32 in every shipped config)", F); return HNR_ERR_BADARG; }
    if (ld1 < (split ? 60 : 7 * F + 60) || (ld1 & 3) || ld3 < 263 || (ld3 & 3) || K <= 0 || K > HNR_MAX_K || SR <= 0) {
        set_error("hnr_gather_rows: bad leading dimensions / sizes"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    GatherArgs a;
    a.xyz = d_xyz; a.emb = d_emb; a.conf = d_conf; a.pdir = d_dir; a.color = d_color; a.F = F;
    a.pidx = d_sample_pidx; a.loc_w = d_sample_loc_w; a.raydir = d_raydir; a.campos = d_campos; a.camrot = d_camrot;
    a.vs_item = d_vs_item; a.vs_off = d_vs_off; a.vs_cnt = d_vs_cnt;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.SR = SR; a.K = K; a.X1 = d_X1; a.ld1 = ld1; a.X3 = d_X3; a.ld3 = ld3; a.wagg = d_wagg;
    a.weight_out = d_weight_out; a.conf_out = d_weight_out ? d_conf_out : nullptr;
    if (d_weight_out && !d_conf_out) { set_error("hnr_gather_rows: weight_out needs conf_out"); return HNR_ERR_BADARG; }
    a.row_pid = d_row_pid;
    if (split) gather_rows_kernel<32, 1><<<cdiv(cap_samples, G_SAMPLES), 256, 0, (hipStream_t)stream>>>(a);
    else gather_rows_kernel<32, 0><<<cdiv(cap_samples, G_SAMPLES), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_point_rows(const float *d_emb, const int32_t *d_ids, int n_points, int F, float *d_E, int lde, void *stream)
{
    if (n_points < 0 || F != 32 || lde < 7 * F || (lde & 3)) { set_error("hnr_point_rows: bad argument (F must be 32, lde >= 224 and a multiple of 4)"); return HNR_ERR_BADARG; }
    if (n_points == 0) return HNR_OK;
    if (!d_emb || !d_E) { set_error("hnr_point_rows: NULL argument"); return HNR_ERR_BADARG; }
    const int64_t total = (int64_t)n_points * (F / 4 + 3 * F);
    int blocks = cdiv(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    point_rows_kernel<32><<<blocks, 256, 0, (hipStream_t)stream>>>(d_emb, d_ids, n_points, d_E, lde);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_ksum(const float *d_H4, int ldh, const float *d_wagg, const float *d_alpha_w, const float *d_alpha_b,
                        const int32_t *d_vs_item, const int32_t *d_vs_off, const int32_t *d_vs_cnt, const float *d_raydir,
                        const int64_t *d_counts, int SR, int cap_samples, float *d_X5, int ld5, float *d_sigma, void *stream)
{
    if (!d_H4 || !d_wagg || !d_alpha_w || !d_alpha_b || !d_vs_item || !d_vs_off || !d_vs_cnt || !d_raydir || !d_counts ||
        !d_X5 || !d_sigma || ldh < 256 || (ldh & 3) || ld5 < 280 || (ld5 & 3)) {
        set_error("hnr_ksum: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    KsumArgs a;
    a.H4 = d_H4; a.ldh = ldh; a.wagg = d_wagg; a.alpha_w = d_alpha_w; a.alpha_b = d_alpha_b;
    a.vs_item = d_vs_item; a.vs_off = d_vs_off; a.vs_cnt = d_vs_cnt; a.raydir = d_raydir;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.SR = SR; a.X5 = d_X5; a.ld5 = ld5; a.sigma = d_sigma;
    ksum_kernel<<<cdiv((int64_t)cap_samples * 64, 256), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

namespace hnr {
// The same convolution with one thread per output PIXEL and all CT output channels of a group in registers: an input tap is loaded once for
// the CT channels (the kernel above loads it once per channel), the weights of a tap are wave-uniform (scalar loads), and with CIN known the
// nine taps of an input channel are in flight together -- the pyramid's six launches were latency-bound (0.9 GFLOP in 0.37 ms).  Per output
// the products are accumulated in the same order (ci, ky, kx) by the same fmaf: bit-identical results.  Border pixels take the checked loop.
template <int CIN, int CT, int CL>
__global__ __launch_bounds__(256) void conv3x3_lrelu_tile_kernel(const float *__restrict__ in, int Hin, int Win, int in_cstride, const float *__restrict__ w,
                                                                 const float *__restrict__ b, int Cout, int stride, int Hout, int Wout, float slope,
                                                                 float *__restrict__ out, int V)
{
    // the group's weights, transposed to [ci][tap][c] in LDS: the CT weights of a tap are one broadcast read (as scalar loads from the [co][ci][tap]
    // image they were 2592 single-dword loads per wave for 24 -> 24 channels, with SGPR spills: 81 us for 0.1 GFLOP)
    __shared__ __attribute__((aligned(16))) float s_w[CIN * 9 * CT];
    const int co0 = blockIdx.y * CT;
    {   // (the group's weights are one contiguous range of the [co][ci][tap] image: coalesced reads, all in flight, transposed on the LDS side)
        constexpr int NW = CIN * 9 * CT, IT = (NW + 255) / 256;
        float wv[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int i = threadIdx.x + 256 * k; wv[k] = i < NW ? w[co0 * CIN * 9 + i] : 0.f; }
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int i = threadIdx.x + 256 * k; if (i < NW) s_w[(i % (CIN * 9)) * CT + i / (CIN * 9)] = wv[k]; }
    }
    __syncthreads();
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (int64_t)V * Hout * Wout) return;
    const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout), v = (int)(pix / ((int64_t)Wout * Hout));
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = b[co0 + c];
    const int iy0 = oy * stride - 1, ix0 = ox * stride - 1;
    auto at = [&](int ci, int iy, int ix) -> float {
        return CL ? in[(((size_t)v * Hin + iy) * Win + ix) * in_cstride + ci] : in[(((size_t)v * CIN + ci) * Hin + iy) * Win + ix];
    };
    constexpr int CB = CIN % 6 == 0 ? 6 : 3;                                 // input channels whose 9 taps are in flight together
    const bool interior = iy0 >= 0 && iy0 + 2 < Hin && ix0 >= 0 && ix0 + 2 < Win;
    if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) {                    // wave-uniform: no pixel of the wave touches the border
        for (int cb = 0; cb < CIN; cb += CB) {
            float x[CB][9];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t) x[u][t] = at(cb + u, iy0 + t / 3, ix0 + t % 3);
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                float wr[9 * CT];                                           // the channel's 9 x CT weights: all reads issued before the first use
#pragma unroll
                for (int i = 0; i < 9 * CT; i += 2) *reinterpret_cast<float2 *>(wr + i) = *reinterpret_cast<const float2 *>(s_w + (cb + u) * 9 * CT + i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = fmaf(x[u][t], wr[t * CT + c], acc[c]);
            }
        }
    } else {
        // a wave with border pixels (every wave of the 80-pixel-wide level): the same unrolled form with clamped addresses, a tap outside the image is
        // SKIPPED by a select after the fmaf (not multiplied by zero: the skipped sum keeps the bits of the checked loop).  A divergent checked loop
        // here cost 216 dependent load round trips per wave: 55 us of the 24 -> 24 launch
        bool ok[9];
        int iyc[3], ixc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            iyc[k] = iy0 + k < 0 ? 0 : (iy0 + k >= Hin ? Hin - 1 : iy0 + k);
            ixc[k] = ix0 + k < 0 ? 0 : (ix0 + k >= Win ? Win - 1 : ix0 + k);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) ok[t] = iy0 + t / 3 >= 0 && iy0 + t / 3 < Hin && ix0 + t % 3 >= 0 && ix0 + t % 3 < Win;
        for (int cb = 0; cb < CIN; cb += CB) {
            float x[CB][9];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t) x[u][t] = at(cb + u, iyc[t / 3], ixc[t % 3]);
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < CT; ++c) { const float f = fmaf(x[u][t], s_w[((cb + u) * 9 + t) * CT + c], acc[c]); acc[c] = ok[t] ? f : acc[c]; }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) out[(((size_t)v * Cout + co0 + c) * Hout + oy) * Wout + ox] = acc[c] > 0.f ? acc[c] : acc[c] * slope;
}
}  // namespace hnr

extern "C" int hnr_image_features(const float *d_img, int V, int H, int W, const float *const *d_conv_w,
                                  const float *const *d_conv_b, float slope, float *d_scratch, float *d_featmap, void *stream)
{
    if (!d_img || !d_conv_w || !d_conv_b || !d_scratch || !d_featmap || V <= 0 || H <= 1 || W <= 1) {
        set_error("hnr_image_features: bad argument"); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    auto o = [](int n) { return (n + 2 - 3) / 2 + 1; };      // conv 3x3, stride 2, padding 1
    const int H1 = o(H), W1 = o(W), H2 = o(H1), W2 = o(W1), H3 = o(H2), W3 = o(W2);
    // scratch layout: s1a s1 s2a s2 s3a s3 (planar NCHW)
    float *s1a = d_scratch, *s1 = s1a + (size_t)V * 6 * H1 * W1;
    float *s2a = s1 + (size_t)V * 6 * H1 * W1, *s2 = s2a + (size_t)V * 12 * H2 * W2;
    float *s3a = s2 + (size_t)V * 12 * H2 * W2, *s3 = s3a + (size_t)V * 24 * H3 * W3;
    static int old_conv = -1;                                              // HNR_CONV_OLD=1: the one-thread-per-output-element kernel (A/B, tests)
    if (old_conv < 0) { const char *e = getenv("HNR_CONV_OLD"); old_conv = e ? atoi(e) : 0; }
    auto conv = [&](const float *in, int Cin, int Hin, int Win, int cl, int cstride, int li, int Cout, int stride, int Hout, int Wout, float *out) {
        const int64_t total = (int64_t)V * Cout * Hout * Wout, pixels = (int64_t)V * Hout * Wout;
#define HNR_CONV_TILE(CIN_, CT_, CL_) \
        if (!old_conv && Cin == CIN_ && Cout % CT_ == 0 && cl == CL_) { \
            conv3x3_lrelu_tile_kernel<CIN_, CT_, CL_><<<dim3((unsigned)cdiv(pixels, 256), Cout / CT_), 256, 0, st>>>(in, Hin, Win, cstride, d_conv_w[li], d_conv_b[li], \
                                                                                                                 Cout, stride, Hout, Wout, slope, out, V); \
            return; }
        HNR_CONV_TILE(3, 6, 1) HNR_CONV_TILE(6, 6, 0) HNR_CONV_TILE(12, 6, 0) HNR_CONV_TILE(24, 6, 0)       // (groups of 12 channels: 20.5 / 34.7 us instead of 15.0 / 22.2)
#undef HNR_CONV_TILE
        conv3x3_lrelu_kernel<<<cdiv(total, 256), 256, 0, st>>>(in, Cin, Hin, Win, cl, cstride, d_conv_w[li], d_conv_b[li], Cout, stride,
                                                                Hout, Wout, slope, out, V);
    };
    conv(d_img, 3, H, W, 1, 3, 0, 6, 2, H1, W1, s1a);
    conv(s1a, 6, H1, W1, 0, 0, 1, 6, 1, H1, W1, s1);
    conv(s1, 6, H1, W1, 0, 0, 2, 12, 2, H2, W2, s2a);
    conv(s2a, 12, H2, W2, 0, 0, 3, 12, 1, H2, W2, s2);
    conv(s2, 12, H2, W2, 0, 0, 4, 24, 2, H3, W3, s3a);
    conv(s3a, 24, H3, W3, 0, 0, 5, 24, 1, H3, W3, s3);
    // four threads per pixel, one part (12 channels) each; a block's threads share the part
    featmap_kernel<<<dim3((unsigned)cdiv((int64_t)V * H * W, 256), 4), 256, 0, st>>>(d_img, s1, s2, s3, V, H, W, H1, W1, H2, W2, H3, W3, d_featmap);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int64_t hnr_image_features_scratch_elems(int V, int H, int W)
{
    auto o = [](int n) { return (n + 2 - 3) / 2 + 1; };
    const int64_t H1 = o(H), W1 = o(W), H2 = o((int)H1), W2 = o((int)W1), H3 = o((int)H2), W3 = o((int)W2);
    return 2 * (int64_t)V * (6 * H1 * W1 + 12 * H2 * W2 + 24 * H3 * W3);
}

extern "C" int hnr_proj_rows(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                             const float *d_intrinsic, const float *d_campos, const float *d_campos_nearest, const float *d_featmap,
                             int V, int H, int W, const float *d_CF, int ldcf, int cap_samples, float *d_X6, int ld6,
                             float *d_vmask, int32_t *d_row_sample, void *stream)
{
    const bool split = d_row_sample != nullptr;
    if (!d_sample_loc_w || !d_vs_item || !d_counts || !d_w2c || !d_intrinsic || !d_campos || !d_campos_nearest || !d_featmap ||
        (!split && !d_CF) || !d_X6 || !d_vmask || V <= 0 || ld6 < (split ? 48 : 176) || (ld6 & 3) || (!split && (ldcf < 128 || (ldcf & 3)))) {
        set_error("hnr_proj_rows: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    ProjArgs a;
    a.loc_w = d_sample_loc_w; a.vs_item = d_vs_item; a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.w2c = d_w2c; a.Kmat = d_intrinsic; a.campos = d_campos; a.campos_n = d_campos_nearest; a.fm = d_featmap; a.H = H; a.W = W;
    a.CF = d_CF; a.ldcf = ldcf; a.V = V; a.cap = cap_samples; a.X6 = d_X6; a.ld6 = ld6; a.vmask = d_vmask; a.row_sample = d_row_sample;
    { const int need = cdiv((int64_t)V * cap_samples * 16, 256); proj_rows_kernel<<<need < 8192 ? need : 8192, 256, 0, (hipStream_t)stream>>>(a); }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// Probe: hnr_div (csrc/hnr_common.h) beside the compiler's correctly rounded `/` on caller-supplied operand pairs
__global__ void div_probe_kernel(const float *n, const float *d, int count, float *q_hnr, float *q_ieee)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { q_hnr[i] = hnr::hnr_div(n[i], d[i]); q_ieee[i] = n[i] / d[i]; }
}
extern "C" int hnr_div_probe(const float *d_num, const float *d_den, int n, float *d_q_hnr, float *d_q_ieee, void *stream)
{
    if (n < 0 || (n > 0 && (!d_num || !d_den || !d_q_hnr || !d_q_ieee))) { set_error("hnr_div_probe: bad argument"); return HNR_ERR_BADARG; }
    if (n == 0) return HNR_OK;
    div_probe_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_num, d_den, n, d_q_hnr, d_q_ieee);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// ... the other two members of the family (round-4 advice): the cell index floor((p - o) / c) of the query kernels (hnr_div_cell inside cell_coord) beside the
// same expression with the compiler's division, and hnr_div64 (the loss kernels' scalar means) beside the compiler's fp64 division
__global__ void div_probe2_kernel(const float *__restrict__ p, const float *__restrict__ c, float o, int count, int32_t *__restrict__ cell_hnr, int32_t *__restrict__ cell_ieee,
                                  double *__restrict__ q64_hnr, double *__restrict__ q64_ieee)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    cell_hnr[i] = hnr::cell_coord(p[i], o, c[i]);
    const float q = __fsub_rn(p[i], o) / c[i];
    cell_ieee[i] = (q > -2.0e9f && q < 2.0e9f) ? (int)floorf(q) : INT32_MIN;
    const double n = (double)p[i] * 1.0000001192092896, d = (double)c[i] * 3.0000000000000004;      // (operands that are not fp32 values)
    q64_hnr[i] = hnr::hnr_div64(n, d); q64_ieee[i] = n / d;
}
extern "C" int hnr_div_probe2(const float *d_p, const float *d_c, float origin, int n, int32_t *d_cell_hnr, int32_t *d_cell_ieee, double *d_q64_hnr, double *d_q64_ieee,
                              void *stream)
{
    if (n < 0 || (n > 0 && (!d_p || !d_c || !d_cell_hnr || !d_cell_ieee || !d_q64_hnr || !d_q64_ieee))) { set_error("hnr_div_probe2: bad argument"); return HNR_ERR_BADARG; }
    if (n == 0) return HNR_OK;
    div_probe2_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(d_p, d_c, origin, n, d_cell_hnr, d_cell_ieee, d_q64_hnr, d_q64_ieee);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_proj_pixels(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                               const float *d_intrinsic, int V, int H, int W, int cap_samples, int32_t *d_pix, void *stream)
{
    if (!d_sample_loc_w || !d_vs_item || !d_counts || !d_w2c || !d_intrinsic || !d_pix || V <= 0 || H <= 0 || W <= 0) {
        set_error("hnr_proj_pixels: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    const int need = cdiv((int64_t)V * cap_samples, 256);
    proj_pixels_kernel<<<need < 8192 ? need : 8192, 256, 0, (hipStream_t)stream>>>(d_sample_loc_w, d_vs_item, reinterpret_cast<const unsigned long long *>(d_counts), d_w2c,
                                                                                      d_intrinsic, V, H, W, cap_samples, d_pix);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_merge(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last,
                         const float *d_vmask, const float *d_frame_w, const float *d_CF, int ldcf, const int64_t *d_counts,
                         int V, int cap_samples, float *d_X7, int ld7, const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR,
                         void *stream)
{
    if (!d_X6 || !d_Hm || !d_w_last || !d_b_last || !d_vmask || !d_CF || !d_counts || !d_X7 || ldh < 64 || ld7 < 90 ||
        (d_ray_drop && (!d_vs_item || SR <= 0))) {
        set_error("hnr_merge: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    MergeArgs a;
    a.X6 = d_X6; a.ld6 = ld6; a.Hm = d_Hm; a.ldh = ldh; a.w_last = d_w_last; a.b_last = d_b_last; a.vmask = d_vmask;
    a.frame_w = d_frame_w; a.CF = d_CF; a.ldcf = ldcf; a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.V = V; a.cap = cap_samples; a.X7 = d_X7; a.ld7 = ld7; a.ray_drop = d_ray_drop; a.vs_item = d_vs_item; a.SR = SR;
    { const int need = cdiv((int64_t)cap_samples * 64, 256); merge_kernel<<<need < 8192 ? need : 8192, 256, 0, (hipStream_t)stream>>>(a); }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_final_color(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                               const float *d_sigma, const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples,
                               float *d_decoded, void *stream)
{
    if (!d_Y || !d_CF || !d_w_fin || !d_b_fin || !d_sigma || !d_vs_item || !d_counts || !d_decoded || ldy < 45 || ldcf < 128 || (ldcf & 3) ||
        ((uintptr_t)d_CF & 15)) {
        set_error("hnr_final_color: bad argument (NULL pointer, ldy < 45, ldcf < 128 or not a multiple of 4, colour feature not 16-B aligned)"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    FinalArgs a;
    a.Y = d_Y; a.ldy = ldy; a.CF = d_CF; a.ldcf = ldcf; a.w_fin = d_w_fin; a.b_fin = d_b_fin; a.sigma = d_sigma;
    a.vs_item = d_vs_item; a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.decoded = d_decoded;
    { const int need = cdiv((int64_t)cap_samples * 16, 256); final_color_kernel<<<need < 8192 ? need : 8192, 256, 0, (hipStream_t)stream>>>(a); }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_composite(const float *d_decoded, const float *d_sample_loc_w, const int32_t *d_sample_pidx,
                             const int8_t *d_ray_mask, const int32_t *d_ray_nsamp, const float *d_campos, const float *d_camrot, const float *d_bg_color,
                             int R, int SR, int K, float vsize_z, int raydist_mode_unit, float *d_raycolor, float *d_opacity,
                             float *d_is_background, float *d_blend_weight, void *stream)
{
    if (R < 0 || SR <= 0 || K <= 0) { set_error("hnr_composite: bad sizes"); return HNR_ERR_BADARG; }
    if (R == 0) return HNR_OK;
    if (!d_decoded || !d_sample_loc_w || !d_sample_pidx || !d_ray_mask || !d_campos || !d_camrot || !d_bg_color || !d_raycolor ||
        !d_opacity || !d_is_background) {
        set_error("hnr_composite: NULL argument"); return HNR_ERR_BADARG;
    }
    CompositeArgs a;
    a.decoded = d_decoded; a.loc_w = d_sample_loc_w; a.pidx = d_sample_pidx; a.ray_mask = d_ray_mask; a.nsamp = d_ray_nsamp; a.campos = d_campos;
    a.camrot = d_camrot; a.bg = d_bg_color; a.R = R; a.SR = SR; a.K = K; a.vsize_z = vsize_z; a.unit_mode = raydist_mode_unit;
    a.raycolor = d_raycolor; a.opacity = d_opacity; a.is_bg = d_is_background; a.blend_w = d_blend_weight;
    composite_kernel<<<cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_ray_march(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *d_bg_color, int R,
                             int SR, float *d_ray_color, float *d_opacity, float *d_acc_transmission, float *d_blend_weight,
                             float *d_bg_transmission, void *stream)
{
    if (R < 0 || SR <= 0) { set_error("hnr_ray_march: bad sizes"); return HNR_ERR_BADARG; }
    if (R == 0) return HNR_OK;
    if (!d_ray_dist || !d_ray_valid || !d_features || !d_ray_color || !d_opacity || !d_acc_transmission || !d_blend_weight || !d_bg_transmission) {
        set_error("hnr_ray_march: NULL argument"); return HNR_ERR_BADARG;
    }
    ray_march_kernel<<<cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(d_ray_dist, d_ray_valid, d_features, d_bg_color, R, SR, d_ray_color, d_opacity,
                                                                    d_acc_transmission, d_blend_weight, d_bg_transmission);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_probe_outputs(const float *d_opacity, const float *d_sample_loc_w, const int32_t *d_sample_pidx, const float *d_weight,
                                 const float *d_conf_coefficient, const float *d_xyz, const float *d_emb, const float *d_conf, const float *d_dir,
                                 const float *d_color, int F, int R, int SR, int K, float *d_max_opacity, float *d_max_loc_w, float *d_far_dist,
                                 float *d_avg_color, float *d_avg_dir, float *d_avg_conf, float *d_avg_emb, void *stream)
{
    if (R < 0 || SR <= 0 || K <= 0 || F <= 0 || F > 32) { set_error("hnr_probe_outputs: bad sizes (F <= 32)"); return HNR_ERR_BADARG; }
    if (R == 0) return HNR_OK;
    if (!d_opacity || !d_sample_loc_w || !d_sample_pidx || !d_weight || !d_conf_coefficient || !d_xyz || !d_emb || !d_conf || !d_dir || !d_color ||
        !d_max_opacity || !d_max_loc_w || !d_far_dist || !d_avg_color || !d_avg_dir || !d_avg_conf || !d_avg_emb) {
        set_error("hnr_probe_outputs: NULL argument"); return HNR_ERR_BADARG;
    }
    ProbeArgs a;
    a.opacity = d_opacity; a.loc_w = d_sample_loc_w; a.pidx = d_sample_pidx; a.weight = d_weight; a.conf_c = d_conf_coefficient; a.xyz = d_xyz;
    a.emb = d_emb; a.conf = d_conf; a.pdir = d_dir; a.color = d_color; a.F = F; a.R = R; a.SR = SR; a.K = K; a.o_opacity = d_max_opacity;
    a.o_loc = d_max_loc_w; a.o_far = d_far_dist; a.o_color = d_avg_color; a.o_dir = d_avg_dir; a.o_conf = d_avg_conf; a.o_emb = d_avg_emb;
    probe_kernel<<<cdiv((int64_t)R * 8, 256), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

namespace hnr {
int point_rows_dc(const float *d_emb, const int32_t *d_ids, int n_cap, const long long *d_n, float *d_E, int lde, hipStream_t st)
{
    if (n_cap <= 0) return HNR_OK;
    const int64_t total = (int64_t)n_cap * (32 / 4 + 3 * 32);
    int blocks = cdiv(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    point_rows_kernel<32><<<blocks, 256, 0, st>>>(d_emb, d_ids, n_cap, d_E, lde, d_n);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
}  // namespace hnr
