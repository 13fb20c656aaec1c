// Backward of the gather / aggregate / composite path: everything that is not a dense layer (those are in h2gemm.hip: input gradients =
// hnr_h2lin with the transposed weights and the LeakyReLU derivative in its epilogue, weight gradients = hnr_h2wgrad); csrc/render_train.hip
// drives them.
//
// The reference gets these from torch autograd over its eager ops; the formulas below are the analytic derivatives of the
// forward kernels in aggregate.hip, one backward kernel per forward kernel, in the same row order:
//   composite_bwd      <- ray_march / alpha blend         models/rendering/diff_ray_marching.py:508-557
//   final_color_bwd    <- color_final_block + sigmoid      models/aggregators/point_aggregators.py:1294-1295, :1334, :478-482
//   merge_bwd          <- weighted merge + last layer      :1199-1217, :1222-1237 (train-time patch drop), :1286-1292
//   proj_rows_bwd      <- pixel gather + F.interpolate     :1064-1067, :1077-1089, :1193
//   conv3x3_bwd_*      <- aux_block_s1..3                  :1047-1063
//   (the alpha branch + K-weighted sums, :1005-1026, :471-476, are transposed by train_ksum_bwd_kernel in render_train.hip: 8-slot row layout)
//   gather_rows_bwd    <- NeuralPoints gather, block3 extras, conf straight-through clamp
//                                                          models/neural_points/neural_points.py:709-720, :957-971, :1422-1424, :1508-1512
//   point_rows_bwd     <- positional encoding of the embedding (:931-938)
// Gradients flow to points_embeding / points_conf / points_dir / points_color and to every aggregator weight; positions
// (xyz, sample locations, view directions) carry no gradient (xyz_grad = 0 in every shipped script).
#include <utility>

#include "hnr_common.h"
#include "train_internal.h"

namespace hnr {

__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ composite
struct CompositeBwdArgs {
    const float *decoded, *loc_w;
    const int32_t *pidx;
    const int8_t *ray_mask;
    const int32_t *nsamp;
    const float *campos, *camrot, *bg;
    int R, SR, K;
    float vsize_z;
    int unit_mode;
    const float *g_raycolor;                             // [R,3] gradient of the (fill_invalid'ed) ray colour
    float *g_decoded;                                    // [R,SR,4] out: (d sigma, d rgb); doubles as the per-ray scratch
};

// one thread per ray: any SR (the wave-per-ray form below holds a ray's samples in the lanes of one wave: SR <= 64)
__global__ __launch_bounds__(256) void composite_bwd_serial_kernel(CompositeBwdArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    float4 *dd = reinterpret_cast<float4 *>(a.g_decoded) + (size_t)r * a.SR;
    if (!a.ray_mask[r]) {
        for (int s = 0; s < a.SR; ++s) dd[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float cr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cr[i] = a.camrot[i];
    const float cp[3] = {a.campos[0], a.campos[1], a.campos[2]};
    const int ns = a.nsamp ? a.nsamp[r] : a.SR;
    auto zc = [&](int s) {
        const float *p = a.loc_w + ((size_t)r * a.SR + s) * 3;
        const float q0 = s < ns ? p[0] : 0.f, q1 = s < ns ? p[1] : 0.f, q2 = s < ns ? p[2] : 0.f;
        const float s0 = __fsub_rn(q0, cp[0]), s1 = __fsub_rn(q1, cp[1]), s2 = __fsub_rn(q2, cp[2]);
        return __fadd_rn(__fadd_rn(__fmul_rn(cr[2], s0), __fmul_rn(cr[5], s1)), __fmul_rn(cr[8], s2));
    };
    // pass 1 (front to back, identical to composite_kernel): park (opacity, ray_dist * valid, T before the sample)
    float T = 1.f;
    float zmax = zc(0);
    for (int s = 0; s < a.SR; ++s) {
        float dist;
        if (s + 1 < a.SR) {
            const float zn = fmaxf(zmax, zc(s + 1));
            dist = __fsub_rn(zn, zmax);
            zmax = zn;
        } else {
            dist = a.vsize_z;
        }
        if (dist < 1e-8f || (a.unit_mode && dist > 2.f * a.vsize_z)) dist = a.vsize_z;
        const bool valid = s < ns && a.pidx[((size_t)r * a.SR + s) * a.K] >= 0;
        const float4 d = reinterpret_cast<const float4 *>(a.decoded)[(size_t)r * a.SR + s];
        const float sigma = valid ? d.x : 0.f;
        const float rd = valid ? dist : 0.f;
        const float o = 1.f - expf(-sigma * rd);
        dd[s] = make_float4(o, rd, T, 0.f);
        T *= (1.f - o + 1e-10f);
    }
    // pass 2 (back to front).  colour = sum_s w_s rgb_s + bg T_end,  w_s = o_s T_s,  T_s = prod_{j<s} q_j,  q = 1 - o + 1e-10:
    //   d/d o_s = T_s (rgb_s . g) - (sum_{j>s} w_j (rgb_j . g) + (bg . g) T_end) / q_s
    const float g0 = a.g_raycolor[3 * (size_t)r], g1 = a.g_raycolor[3 * (size_t)r + 1], g2 = a.g_raycolor[3 * (size_t)r + 2];
    float S = (a.bg[0] * g0 + a.bg[1] * g1 + a.bg[2] * g2) * T;
    for (int s = a.SR - 1; s >= 0; --s) {
        const float4 t = dd[s];
        const float o = t.x, rd = t.y, Ts = t.z;
        const float4 d = reinterpret_cast<const float4 *>(a.decoded)[(size_t)r * a.SR + s];
        const float gs = d.y * g0 + d.z * g1 + d.w * g2;
        const float q = 1.f - o + 1e-10f;
        const float d_o = Ts * gs - hnr_div(S, q);
        const float w = o * Ts;
        S += gs * w;
        const float sigma = rd > 0.f ? d.x : 0.f;
        const float d_sigma = d_o * rd * expf(-sigma * rd);
        dd[s] = make_float4(d_sigma, w * g0, w * g1, w * g2);
    }
}

// One WAVE per ray, lane = shading sample (SR <= 64): every lane loads its own sample's operands up front, the two serial recurrences (front
// to back: running depth maximum and transmittance; back to front: the suffix sum S) then run over wave-uniform values fetched with
// v_readlane, each lane keeping the step that is its own.  The same operations in the same order as one thread per ray -- which took 56 us
// for 3 136 rays: 2 x 24 dependent steps, each waiting for its own loads (thirteen workgroups on the whole chip).
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompositeBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int r = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    if (r >= a.R) return;
    float4 *dd = reinterpret_cast<float4 *>(a.g_decoded) + (size_t)r * a.SR;
    if (!a.ray_mask[r]) {
        if (lane < a.SR) dd[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int ns = a.nsamp ? a.nsamp[r] : a.SR;
    const bool in = lane < a.SR;
    // this lane's sample: camera depth, validity, decoded (sigma, rgb)
    float z_l = 0.f;
    bool valid_l = false;
    float4 d_l = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        const float *p = a.loc_w + ((size_t)r * a.SR + lane) * 3;
        const float q0 = lane < ns ? p[0] : 0.f, q1 = lane < ns ? p[1] : 0.f, q2 = lane < ns ? p[2] : 0.f;
        const float s0 = __fsub_rn(q0, a.campos[0]), s1 = __fsub_rn(q1, a.campos[1]), s2 = __fsub_rn(q2, a.campos[2]);
        z_l = __fadd_rn(__fadd_rn(__fmul_rn(a.camrot[2], s0), __fmul_rn(a.camrot[5], s1)), __fmul_rn(a.camrot[8], s2));
        valid_l = lane < ns && a.pidx[((size_t)r * a.SR + lane) * a.K] >= 0;
        d_l = reinterpret_cast<const float4 *>(a.decoded)[(size_t)r * a.SR + lane];
    }
    const float sig_l = valid_l ? d_l.x : 0.f;
    auto bc = [&](float v, int s) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s)); };
    // pass 1 (front to back, identical to composite_kernel): (opacity, ray_dist * valid, T before the sample) of every sample
    float T = 1.f, zmax = bc(z_l, 0);
    float o_l = 0.f, rd_l = 0.f, T_l = 1.f;
    for (int s = 0; s < a.SR; ++s) {
        float dist;
        if (s + 1 < a.SR) {
            const float zn = fmaxf(zmax, bc(z_l, s + 1));
            dist = __fsub_rn(zn, zmax);
            zmax = zn;
        } else {
            dist = a.vsize_z;
        }
        if (dist < 1e-8f || (a.unit_mode && dist > 2.f * a.vsize_z)) dist = a.vsize_z;
        const bool valid = __builtin_amdgcn_readlane((int)valid_l, s) != 0;
        const float sigma = bc(sig_l, s);
        const float rd = valid ? dist : 0.f;
        const float o = 1.f - expf(-sigma * rd);
        if (lane == s) { o_l = o; rd_l = rd; T_l = T; }
        T *= (1.f - o + 1e-10f);
    }
    // pass 2 (back to front).  colour = sum_s w_s rgb_s + bg T_end,  w_s = o_s T_s,  T_s = prod_{j<s} q_j,  q = 1 - o + 1e-10:
    //   d/d o_s = T_s (rgb_s . g) - (sum_{j>s} w_j (rgb_j . g) + (bg . g) T_end) / q_s
    const float g0 = a.g_raycolor[3 * (size_t)r], g1 = a.g_raycolor[3 * (size_t)r + 1], g2 = a.g_raycolor[3 * (size_t)r + 2];
    float S = (a.bg[0] * g0 + a.bg[1] * g1 + a.bg[2] * g2) * T;
    const float gs_l = d_l.y * g0 + d_l.z * g1 + d_l.w * g2;
    const float w_l = o_l * T_l;
    float S_l = 0.f;                                     // S as it stands when the loop reaches this lane's sample
    for (int s = a.SR - 1; s >= 0; --s) {
        if (lane == s) S_l = S;
        S += bc(gs_l, s) * bc(w_l, s);
    }
    if (in) {
        const float q = 1.f - o_l + 1e-10f;
        const float d_o = T_l * gs_l - hnr_div(S_l, q);
        const float sigma = rd_l > 0.f ? d_l.x : 0.f;
        const float d_sigma = d_o * rd_l * expf(-sigma * rd_l);
        dd[lane] = make_float4(d_sigma, w_l * g0, w_l * g1, w_l * g2);
    }
}

// ------------------------------------------------------------------------------------------------ final colour
struct FinalBwdArgs {
    const float *Y; int ldy;
    const float *CF; int ldcf;
    const float *w_fin, *b_fin;
    const int32_t *vs_item;
    const unsigned long long *counts;
    const float *g_decoded;                              // [R*SR,4]
    float *gY; int ldgy;                                 // [S, ldgy] d color_mixup_block output (45 columns)
    float *gCF; int ldgcf;                               // [S, ldgcf] d colour feature (128 columns, OVERWRITTEN)
    float *g_sigma;                                      // [S]
    float *g_w_fin, *g_b_fin;                            // [3*128], [3]  accumulated with atomics
    unsigned *gY_max;                                    // optional: max |gY| (bit pattern, atomicMax)
};

__global__ __launch_bounds__(1024) void final_color_bwd_kernel(FinalBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * (unsigned)blockDim.x) >> 6);
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    float aw[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, ab[3] = {0.f, 0.f, 0.f};
    float gy_max = 0.f;
    float wf[2][3], bf[3];                               // this lane's six weights: loaded once, not per sample
#pragma unroll
    for (int j = 0; j < 3; ++j) { wf[0][j] = a.w_fin[j * 128 + lane]; wf[1][j] = a.w_fin[j * 128 + 64 + lane]; bf[j] = a.b_fin[j]; }
    for (int s = wave; s < n_valid; s += n_waves) {
        // every load of the sample is issued before the first use (the upstream gradient is behind an index: its two dependent loads go first)
        const float4 g = reinterpret_cast<const float4 *>(a.g_decoded)[a.vs_item[s]];
        const float cf0 = a.CF[(size_t)s * a.ldcf + lane], cf1 = a.CF[(size_t)s * a.ldcf + 64 + lane];
        const float y0 = lane < 45 ? a.Y[(size_t)s * a.ldy + lane] : 0.f;
        float x[2] = {lane < 45 ? y0 + cf0 : cf0, cf1}, r[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { r[j] = 0.f; r[j] += x[0] * wf[0][j]; r[j] += x[1] * wf[1][j]; }
        const float gc[3] = {g.y, g.z, g.w};
        float dz[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            r[j] = wave_sum(r[j]);
            const float sg = hnr_div(1.f, 1.f + expf(-(r[j] + bf[j])));
            dz[j] = gc[j] * (1.f + 2.f * 0.001f) * sg * (1.f - sg);
            ab[j] += dz[j];
        }
        if (lane == 0) a.g_sigma[s] = g.x;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            float dx = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) { dx += dz[j] * wf[h][j]; aw[h][j] += dz[j] * x[h]; }
            a.gCF[(size_t)s * a.ldgcf + c] = dx;
            if (c < 45) { a.gY[(size_t)s * a.ldgy + c] = dx; gy_max = fmaxf(gy_max, fabsf(dx)); }
        }
    }
    // the 16 waves of the block add up in LDS first: one atomic per block and address (the atomics of one address retire one per ~23 ns: the
    // grid is one 16-wave block per CU, not many small ones)
    __shared__ float s_w[16][6][64];
    __shared__ float s_b[16][3];
    __shared__ float s_m[16];
    const int wid = threadIdx.x >> 6;
    if (a.gY_max) {
        for (int o = 32; o > 0; o >>= 1) gy_max = fmaxf(gy_max, __shfl_xor(gy_max, o));
        if (lane == 0) s_m[wid] = gy_max;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 3; ++j) s_w[wid][h * 3 + j][lane] = aw[h][j];
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < 3; ++j) s_b[wid][j] = ab[j];
    __syncthreads();
    if (wid < 6) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_w[w][wid][lane];
        const int h = wid / 3, j = wid - 3 * h;
        atomicAdd(a.g_w_fin + j * 128 + lane + 64 * h, t);
    } else if (wid == 6 && lane < 3) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_b[w][lane];
        atomicAdd(a.g_b_fin + lane, t);
    } else if (wid == 7 && lane == 0 && a.gY_max) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) m = fmaxf(m, s_m[w]);
        absmax_publish(a.gY_max, m);
    }
}

// ------------------------------------------------------------------------------------------------ merge
struct MergeBwdArgs {
    const float *X6; int ld6;
    const float *Hm; int ldh;
    const float *w_last, *b_last;
    const float *vmask, *frame_w;
    const unsigned long long *counts;
    int V, cap;
    float slope;
    const uint8_t *ray_drop; const int32_t *vs_item; int SR;   // train-time patch drop (ray_drop may be NULL)
    const float *gX7; int ldg7;                          // [S, ldg7] d mix-up input: [d colfeat[:45] | d merged45]
    float *gF; int ldgf;                                 // [V*cap, ldgf] out: d image feature rows (48 columns, 45..47 = 0)
    float *gZ3; int ldgz;                                // [V*cap, ldgz] out: d PRE-activation of the last hidden layer (64)
    float *gCF; int ldgcf;                               // [S, ldgcf] += d colfeat[:45]
    float *g_w_last, *g_b_last;                          // [64], [1] atomics
    unsigned *gZ3_max;                                   // optional: max |gZ3| (bit pattern, atomicMax)
};

constexpr int MAXV = 8;

// VT: compile-time bound of the view loops (4: the shipped configurations; MAXV: any a.V <= MAXV, twice the registers)
template <int VT, int NW>
__global__ __launch_bounds__(64 * NW) void merge_bwd_kernel(MergeBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * (unsigned)blockDim.x) >> 6);
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const float wl = a.w_last[lane], bl = a.b_last[0];
    float acc_w = 0.f, acc_b = 0.f, gz_max = 0.f;
    for (int s = wave; s < n_valid; s += n_waves) {
        float sg[VT], wv[VT], f[VT], hm[VT], scale[VT], vm[VT];
        float fsum = 0.f, wsum = 0.f;
        // all loads of the sample first: with each view's loads behind the previous view's wave reduction an iteration was five dependent
        // memory round trips (8 us per sample and wave)
        const bool drop = a.ray_drop && a.ray_drop[a.vs_item[s] / a.SR];
        const float dm_in = lane < 45 ? a.gX7[(size_t)s * a.ldg7 + 45 + lane] : 0.f;
        const float gcf_in = lane < 45 ? a.gX7[(size_t)s * a.ldg7 + lane] : 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            if (v >= a.V) break;
            const size_t row = (size_t)v * a.cap + s;
            hm[v] = a.Hm[row * a.ldh + lane];
            vm[v] = a.vmask[row];
            f[v] = lane < 45 ? a.X6[row * a.ld6 + lane] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            if (v >= a.V) break;
            const float d = wave_sum(hm[v] * wl);
            sg[v] = hnr_div(1.f, 1.f + expf(-(d + bl)));
            scale[v] = vm[v] * (a.frame_w ? a.frame_w[v] : 1.f);
            wv[v] = sg[v] * vm[v];
            if (a.frame_w) wv[v] *= a.frame_w[v];
            fsum += f[v] * wv[v];
            wsum += wv[v];
        }
        const float den = wsum + 1e-6f;
        const float merged = hnr_div(fsum, den);
        const float dm = drop ? 0.f : dm_in;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            if (v >= a.V) break;
            const size_t row = (size_t)v * a.cap + s;
            if (lane < 48) a.gF[row * a.ldgf + lane] = hnr_div(dm * wv[v], den);
            const float d_wv = wave_sum(hnr_div(dm * (f[v] - merged), den));
            const float d_pre = d_wv * scale[v] * sg[v] * (1.f - sg[v]);
            const float gz = d_pre * wl * (hm[v] > 0.f ? 1.f : a.slope);
            a.gZ3[row * a.ldgz + lane] = gz;
            gz_max = fmaxf(gz_max, fabsf(gz));
            acc_w += d_pre * hm[v];
            acc_b += d_pre;
        }
        if (lane < 45) a.gCF[(size_t)s * a.ldgcf + lane] += gcf_in;
    }
    __shared__ float s_w[NW][64];
    __shared__ float s_b[NW], s_m[NW];
    const int wid = threadIdx.x >> 6;
    s_w[wid][lane] = acc_w;
    for (int o = 32; o > 0; o >>= 1) gz_max = fmaxf(gz_max, __shfl_xor(gz_max, o));
    if (lane == 0) { s_b[wid] = acc_b; s_m[wid] = gz_max; }
    __syncthreads();
    if (wid == 0) {
        float t = 0.f, tb = 0.f, m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t += s_w[w][lane]; tb += s_b[w]; m = fmaxf(m, s_m[w]); }
        atomicAdd(a.g_w_last + lane, t);
        if (lane == 0) atomicAdd(a.g_b_last, tb);
        if (lane == 0 && a.gZ3_max) absmax_publish(a.gZ3_max, m);
    }
}

// ------------------------------------------------------------------------------------------------ pixel gather + upsample
// Three steps, so that the many samples of a training batch that reproject into the same few pixels / pyramid cells do not
// serialise on atomics: (1) every (view, sample) row gets its full-resolution pixel index as a key and widens the per-view
// bounding box of touched pixels; (2) the rows are summed per pixel into g_featmap (sort by key + running sums,
// segment.hip); (3) per pyramid level a gather kernel applies the transpose of F.interpolate(bilinear,
// align_corners=False) over the destination pixels inside the bounding box -- no atomics, fixed order.
struct ProjBwdArgs {
    const float *loc_w;
    const int32_t *vs_item;
    const unsigned long long *counts;
    const float *w2c, *Kmat;
    int V, H, W, cap;
    int32_t *keys;                                       // [V*cap] out: pixel index (v*H + y)*W + x of the row, -1 = no gradient
    int32_t *bbox;                                       // [V,4] = min x, min y, max x, max y; initialised {W, H, -1, -1}
};

__global__ __launch_bounds__(1024) void proj_rows_bwd_kernel(ProjBwdArgs a)
{
    // one lane per (view, sample) row: pixel key; per-wave min/max of the touched pixels -> one atomic per wave and bound
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const int v = n_valid > 0 ? (int)(t / n_valid) : a.V;
    const bool live = v < a.V;
    int px = -1, py = -1;
    bool none = true;
    if (live) {
        const int s = (int)(t - (int64_t)v * n_valid);
        const float *p = a.loc_w + (size_t)a.vs_item[s] * 3;
        const float x = p[0], y = p[1], z = p[2];
        // masked row: reads the zeroed pixel (0,0), no gradient; the feature at (0,0) is the constant 0 (:1089)
        none = hnr_project_pixel(x, y, z, a.w2c + 16 * v, a.Kmat, a.W, a.H, px, py) || (px == 0 && py == 0);
        a.keys[(size_t)v * a.cap + s] = none ? -1 : (v * a.H + py) * a.W + px;
    }
    // bounds of the touched pixels per view: wave reduction (the lanes of a wave almost always belong to one view: rows are view-major; the few
    // lanes of another view go by themselves) -> LDS -> at most 16 global atomics per 1024-thread block.  (One set of global atomics per WAVE
    // was 9600 atomics on one cache line, ~12 ns each: 0.116 ms of the training step for a kernel with 10 us of work.)
    __shared__ int s_bb[8][4];
    if (threadIdx.x < 32) s_bb[threadIdx.x >> 2][threadIdx.x & 3] = (threadIdx.x & 2) ? -1 : 0x7fffffff;
    __syncthreads();
    const int v0 = __shfl(v, 0);
    const bool mine = !none && v == v0;
    int x0 = mine ? px : 0x7fffffff, y0 = mine ? py : 0x7fffffff, x1 = mine ? px : -1, y1 = mine ? py : -1;
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
        x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        int *bb = v0 < 8 ? s_bb[v0] : a.bbox + 4 * v0;                      // (more than 8 reference views: straight to the global words)
        atomicMin(bb, x0); atomicMin(bb + 1, y0); atomicMax(bb + 2, x1); atomicMax(bb + 3, y1);
    }
    if (!none && v != v0) {
        int *bb = v < 8 ? s_bb[v] : a.bbox + 4 * v;
        atomicMin(bb, px); atomicMin(bb + 1, py); atomicMax(bb + 2, px); atomicMax(bb + 3, py);
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int vv = threadIdx.x >> 2, k = threadIdx.x & 3;
        if (vv < a.V && s_bb[vv][2] >= 0) { if (k < 2) atomicMin(a.bbox + 4 * vv + k, s_bb[vv][k]); else atomicMax(a.bbox + 4 * vv + k, s_bb[vv][k]); }
    }
}

// g_featmap[key, c] += gFa[row, c] (+ gFb[row, c]): one lane per (row, channel).  153 k rows of a training batch: sorting them by pixel first
// (21 launches of the library sort + a segment sum: 0.17 ms) cost five times what these 7 M float atomics do, and the convolution weight
// gradients downstream are atomic sums already.
__global__ __launch_bounds__(256) void pixel_scatter_add_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                                const int32_t *__restrict__ keys, int64_t cap, int V, const unsigned long long *__restrict__ counts,
                                                                float *__restrict__ g_fm)
{
    const int64_t n_valid = (int64_t)counts[HNR_CNT_SAMPLES_VALID];
    const int64_t total = (int64_t)V * n_valid * 48;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / 48;
        const int c = (int)(t - r * 48);
        const int64_t v = r / n_valid, row = v * cap + (r - v * n_valid);      // physical row of (view, sample)
        const int key = keys[row];
        if (key < 0) continue;
        float x = A[(size_t)row * lda + c];
        if (B) x += B[(size_t)row * ldb + c];
        atomicAdd(g_fm + (size_t)key * 48 + c, x);
    }
}

// transpose of bilinear_at (aggregate.hip): one lane per (view, source cell, channel of the level), channel fastest
__device__ __forceinline__ void upsample_bwd_body(int64_t block, const float *__restrict__ g_fm, const int32_t *__restrict__ bbox, int V, int H, int W,
                                                  int Hs, int Ws, int C, int c0, float *__restrict__ g_level /*[V,C,Hs,Ws]*/)
{
    const int64_t idx = block * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)V * Hs * Ws * C) return;
    const int cl = (int)(idx % C), xs = (int)((idx / C) % Ws), ys = (int)((idx / ((int64_t)C * Ws)) % Hs), v = (int)(idx / ((int64_t)C * Ws * Hs));
    const int bx0 = bbox[4 * v], by0 = bbox[4 * v + 1], bx1 = bbox[4 * v + 2], by1 = bbox[4 * v + 3];
    if (bx1 < bx0) return;
    const float sy = hnr_div((float)Hs, (float)H), sx = hnr_div((float)Ws, (float)W);
    // destination pixels that can touch this source cell: source coordinate (y + 0.5) sy - 0.5 within (ys - 1, ys + 1), i.e. y within
    // ((ys - 0.5) / sy - 0.5, (ys + 1.5) / sy - 0.5), widened by one pixel; row / column 0 also takes the coordinates clamped up to 0
    // (the window used to be (3 r + 2)^2 pixels, r = H / Hs, for the (2 r)^2 that carry weight: 2.6x the loop trips at level 3)
    int ylo = ys == 0 ? 0 : (int)floorf(hnr_div((float)ys - 0.5f, sy) - 0.5f) - 1, yhi = (int)ceilf(hnr_div((float)ys + 1.5f, sy) - 0.5f) + 1;
    int xlo = xs == 0 ? 0 : (int)floorf(hnr_div((float)xs - 0.5f, sx) - 0.5f) - 1, xhi = (int)ceilf(hnr_div((float)xs + 1.5f, sx) - 0.5f) + 1;
    ylo = ylo < by0 ? by0 : ylo; yhi = yhi > by1 + 1 ? by1 + 1 : yhi;
    xlo = xlo < bx0 ? bx0 : xlo; xhi = xhi > bx1 + 1 ? bx1 + 1 : xhi;
    float acc = 0.f;
    for (int y = ylo; y < yhi; ++y) {
        float qy = ((float)y + 0.5f) * sy - 0.5f;
        if (qy < 0.f) qy = 0.f;
        const int y0 = (int)qy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
        const float ly = qy - (float)y0;
        const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int x = xlo; x < xhi; ++x) {
            float qx = ((float)x + 0.5f) * sx - 0.5f;
            if (qx < 0.f) qx = 0.f;
            const int x0 = (int)qx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
            const float lx = qx - (float)x0;
            const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
            if (wx == 0.f) continue;
            acc += wy * wx * g_fm[(((size_t)v * H + y) * W + x) * 48 + c0 + cl];
        }
    }
    g_level[(((size_t)v * C + cl) * Hs + ys) * Ws + xs] = acc;
}
// the three pyramid levels in one launch (independent of one another): blocks [0, n1) level 1, [n1, n1 + n2) level 2, the rest level 3
struct UpsampleBwdArgs { const float *g_fm; const int32_t *bbox; int V, H, W; int Hs[3], Ws[3], C[3], c0[3]; float *g[3]; int nb[3]; };
__global__ __launch_bounds__(256) void upsample_bwd_kernel(UpsampleBwdArgs a)
{
    int b = (int)blockIdx.x;
    if (b < a.nb[0]) { upsample_bwd_body(b, a.g_fm, a.bbox, a.V, a.H, a.W, a.Hs[0], a.Ws[0], a.C[0], a.c0[0], a.g[0]); return; }
    b -= a.nb[0];
    if (b < a.nb[1]) { upsample_bwd_body(b, a.g_fm, a.bbox, a.V, a.H, a.W, a.Hs[1], a.Ws[1], a.C[1], a.c0[1], a.g[1]); return; }
    b -= a.nb[1];
    upsample_bwd_body(b, a.g_fm, a.bbox, a.V, a.H, a.W, a.Hs[2], a.Ws[2], a.C[2], a.c0[2], a.g[2]);
}

// ------------------------------------------------------------------------------------------------ 3x3 convolutions
// g_out is the gradient w.r.t. the POST-activation output `out`; d pre-activation dz = g_out * (out > 0 ? 1 : slope).
// One block per tile of TO x TO OUTPUT pixels of one view (and the S*TO x S*TO input pixels under it).
// Both gradients of a layer from one LDS copy of the tile: dz = g_out * leaky'(out) with its one-pixel halo, the layer input with its halo, the
// weights transposed to [tap][co][ci].  Data gradient: a thread owns one input pixel (stride 1) or one 2x2 input quad (stride 2: each of the
// nine taps reaches exactly one pixel of the quad, so no lane multiplies by a structural zero) for CIN / NG channels and writes g_in itself (no
// atomics).  Weight gradient: a thread owns 3 output channels x 1 input channel x 9 taps over every `slices`-th pixel of the tile; the slices are
// summed in a fixed order through LDS and the block adds ONE partial per weight with a float atomic (the sum over blocks is the only
// order-dependent part).  A per-element form (one lane per input element / per (co, ci) pair, operands from L2) spent ~10 integer and address
// instructions per multiply-add: 1.35 ms of kernel time per training step against 0.68 ms here (both stretched by the main stream's kernels
// they run beside -- profiles/README.md).
struct ConvTileArgs {
    const float *g_out, *out, *w, *in; int in_cstride, Hin, Win, Hout, Wout; float slope;
    float *g_in, *g_w, *g_b; int V; const int32_t *bbox; int f_out, halo_out; int tiles_x, tiles_y;
};
template <int CIN, int COUT, int S, int TO, bool DGRAD, bool IN_CL>
__global__ __launch_bounds__(256) void conv3x3_bwd_tile_kernel(ConvTileArgs a)
{
    constexpr int NP = TO * TO;                       // output pixels (= input pixels at stride 1, input quads at stride 2) of the tile
    constexpr int DR = (S == 1) ? TO + 2 : TO + 1;    // dz rows/cols held: stride 1 [oy0 - 1, oy0 + TO], stride 2 [oy0, oy0 + TO]
    constexpr int IR = S * TO + ((S == 1) ? 2 : 1);   // input rows/cols held: [S * oy0 - 1, S * oy0 + S * TO - 1 (+1 at stride 1)]
    constexpr int DSTR = (DR * DR) | 1, ISTR = (IR * IR) | 1;        // odd channel strides: lanes that differ in channel hit different banks
    constexpr int CG = COUT / 3, UNITS = CG * CIN, SLICES = 256 / UNITS;
    constexpr int NG = 256 / NP, CPG = CIN / NG;      // data gradient: channel groups per pixel, channels per group
    static_assert(COUT % 3 == 0 && UNITS <= 256 && NP <= 256 && 256 % NP == 0 && CIN % NG == 0, "tile shape");
    extern __shared__ float smem[];
    float *s_dz = smem, *s_in = s_dz + COUT * DSTR, *s_w = s_in + CIN * ISTR, *s_part = s_w + (DGRAD ? 9 * COUT * CIN : 0);
    const int t = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y, v = b / a.tiles_y;
    const int oy0 = ty * TO, ox0 = tx * TO;
    if (a.bbox) {
        // nothing outside the (dilated) image of the pixels the batch touched in this view carries a gradient
        const int bx0 = a.bbox[4 * v], by0 = a.bbox[4 * v + 1], bx1 = a.bbox[4 * v + 2], by1 = a.bbox[4 * v + 3];
        if (bx1 < bx0 || ox0 + TO + 1 < bx0 / a.f_out - a.halo_out || ox0 - 1 > bx1 / a.f_out + a.halo_out ||
            oy0 + TO + 1 < by0 / a.f_out - a.halo_out || oy0 - 1 > by1 / a.f_out + a.halo_out) return;
    }
    // ---- load
    constexpr int D0 = (S == 1) ? 1 : 0;
    // (hnr_opaque: the tile indices are small enough for the compiler to do this arithmetic in PACKED 16-bit instructions -- v_pk_mad_u16 and friends, 20 of
    // them in the <24,24,1,8> instance.  Nothing is known against them, but the failure of the packed fp32 ones beside other waves' MFMAs (DESIGN.md
    // section 2) was never explained either, so the library carries NO v_pk_* instruction at all: tests/test_abi_and_host.py checks the disassembly.)
    for (int i0 = t; i0 < COUT * DR * DR; i0 += 256) {
        const int i = hnr_opaque(i0);
        const int c = i / (DR * DR), r = i % (DR * DR), oy = oy0 - D0 + r / DR, ox = ox0 - D0 + r % DR;
        float dz = 0.f;
        if (oy >= 0 && oy < a.Hout && ox >= 0 && ox < a.Wout) {
            const size_t o = (((size_t)v * COUT + c) * a.Hout + oy) * a.Wout + ox;
            dz = a.g_out[o] * (a.out[o] > 0.f ? 1.f : a.slope);
        }
        s_dz[c * DSTR + r] = dz;
    }
    for (int i0 = t; i0 < CIN * IR * IR; i0 += 256) {
        const int i = hnr_opaque(i0);
        const int c = i / (IR * IR), r = i % (IR * IR), iy = S * oy0 - 1 + r / IR, ix = S * ox0 - 1 + r % IR;
        float x = 0.f;
        if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
            x = IN_CL ? a.in[(((size_t)v * a.Hin + iy) * a.Win + ix) * a.in_cstride + c] : a.in[(((size_t)v * CIN + c) * a.Hin + iy) * a.Win + ix];
        s_in[c * ISTR + r] = x;
    }
    if (DGRAD)
        for (int i0 = t; i0 < 9 * COUT * CIN; i0 += 256) {          // w[co][ci][k] -> s_w[k][co][ci]
            const int i = hnr_opaque(i0);
            const int k = i % 9, ci = (i / 9) % CIN, co = i / (9 * CIN);
            s_w[(k * COUT + co) * CIN + ci] = a.w[i];
        }
    __syncthreads();
    // ---- data gradient
    if (DGRAD) {
        const int p = t % NP, g = __builtin_amdgcn_readfirstlane(t / NP), ly = p / TO, lx = p % TO;
        if (S == 1) {
            float acc[CPG];
#pragma unroll
            for (int c = 0; c < CPG; ++c) acc[c] = 0.f;
#pragma unroll 1
            for (int co = 0; co < COUT; ++co) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float dz = s_dz[co * DSTR + (ly + 2 - ky) * DR + (lx + 2 - kx)];
                        const float *wp = s_w + ((ky * 3 + kx) * COUT + co) * CIN + g * CPG;
#pragma unroll
                        for (int c = 0; c < CPG; ++c) acc[c] = fmaf(dz, wp[c], acc[c]);
                    }
            }
            const int iy = oy0 + ly, ix = ox0 + lx;
            if (iy < a.Hin && ix < a.Win)
#pragma unroll
                for (int c = 0; c < CPG; ++c) a.g_in[(((size_t)v * CIN + g * CPG + c) * a.Hin + iy) * a.Win + ix] += acc[c];
        } else {
            float acc[2][2][CPG];
#pragma unroll
            for (int i = 0; i < 4 * CPG; ++i) (&acc[0][0][0])[i] = 0.f;
#pragma unroll 1
            for (int co = 0; co < COUT; ++co) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        // input row 2q + py meets tap ky at output row (2q + py + 1 - ky) / 2: py = 0 -> ky = 1 (row q); py = 1 -> ky = 0 (row q + 1), ky = 2 (row q)
                        const int py = (ky == 1) ? 0 : 1, px = (kx == 1) ? 0 : 1, dy = (ky == 0) ? 1 : 0, dx = (kx == 0) ? 1 : 0;
                        const float dz = s_dz[co * DSTR + (ly + dy) * DR + (lx + dx)];
                        const float *wp = s_w + ((ky * 3 + kx) * COUT + co) * CIN + g * CPG;
#pragma unroll
                        for (int c = 0; c < CPG; ++c) acc[py][px][c] = fmaf(dz, wp[c], acc[py][px][c]);
                    }
            }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int iy = 2 * (oy0 + ly) + py, ix = 2 * (ox0 + lx) + px;
                    if (iy < a.Hin && ix < a.Win)
#pragma unroll
                        for (int c = 0; c < CPG; ++c) a.g_in[(((size_t)v * CIN + g * CPG + c) * a.Hin + iy) * a.Win + ix] += acc[py][px][c];
                }
        }
    }
    // ---- weight gradient
    const int u = t % UNITS, slice = t / UNITS, ci = u % CIN, cg = u / CIN;
    float wacc[3][9];
#pragma unroll
    for (int i = 0; i < 27; ++i) (&wacc[0][0])[i] = 0.f;
    if (slice < SLICES) {
        for (int p = slice; p < NP; p += SLICES) {
            const int ly = p / TO, lx = p % TO;
            const float *dp = s_dz + (3 * cg) * DSTR + (ly + D0) * DR + (lx + D0);
            const float d0 = dp[0], d1 = dp[DSTR], d2 = dp[2 * DSTR];
            const float *ip = s_in + ci * ISTR + (S * ly) * IR + S * lx;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float x = ip[ky * IR + kx];
                    wacc[0][ky * 3 + kx] = fmaf(d0, x, wacc[0][ky * 3 + kx]);
                    wacc[1][ky * 3 + kx] = fmaf(d1, x, wacc[1][ky * 3 + kx]);
                    wacc[2][ky * 3 + kx] = fmaf(d2, x, wacc[2][ky * 3 + kx]);
                }
        }
    }
    if (SLICES == 1) {
        if (slice < SLICES)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 9; ++k) atomicAdd(a.g_w + ((size_t)(3 * cg + j) * CIN + ci) * 9 + k, wacc[j][k]);
    } else {
        if (slice < SLICES)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 9; ++k) s_part[(size_t)slice * (COUT * CIN * 9) + ((3 * cg + j) * CIN + ci) * 9 + k] = wacc[j][k];
        __syncthreads();
        for (int o = t; o < COUT * CIN * 9; o += 256) {
            float sum = 0.f;
            for (int sl = 0; sl < SLICES; ++sl) sum += s_part[(size_t)sl * (COUT * CIN * 9) + o];
            atomicAdd(a.g_w + o, sum);
        }
    }
    // bias: sum of dz over the tile's own pixels
    if (t < COUT) {
        float sum = 0.f;
        for (int p = 0; p < NP; ++p) sum += s_dz[t * DSTR + (p / TO + D0) * DR + (p % TO + D0)];
        atomicAdd(a.g_b + t, sum);
    }
}
template <int CIN, int COUT, int S, int TO, bool DGRAD, bool IN_CL>
static int conv3x3_bwd_tile_launch(ConvTileArgs a, hipStream_t st)
{
    constexpr int DR = (S == 1) ? TO + 2 : TO + 1, IR = S * TO + ((S == 1) ? 2 : 1), DSTR = (DR * DR) | 1, ISTR = (IR * IR) | 1;
    constexpr int UNITS = (COUT / 3) * CIN, SLICES = 256 / UNITS;
    constexpr size_t lds = sizeof(float) * ((size_t)COUT * DSTR + (size_t)CIN * ISTR + (DGRAD ? 9 * COUT * CIN : 0) + (SLICES > 1 ? (size_t)SLICES * COUT * CIN * 9 : 0));
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv3x3_bwd_tile_kernel<CIN, COUT, S, TO, DGRAD, IN_CL>;
    static PerDeviceOnce attr;
    if (attr.first()) HNR_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    a.tiles_x = (a.Wout + TO - 1) / TO; a.tiles_y = (a.Hout + TO - 1) / TO;
    kern<<<a.tiles_x * a.tiles_y * a.V, 256, lds, st>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// ------------------------------------------------------------------------------------------------ gather
struct GatherBwdArgs {
    const int32_t *pidx;
    const float *raydir;
    const int32_t *vs_item, *vs_off, *vs_cnt;
    const unsigned long long *counts;
    int SR, K;
    const float *gX3; int ldg3;                          // [rows, ldg3] d block3 input; columns 256..262 = d [colour3 | dir - view | dir . view]
    const float *g_wagg;                                 // [rows] d (normalised weight * clamp(conf))
    const float *weight;                                 // [R,SR,K] normalised weights (forward output)
    const float *g_conf_out;                             // [R,SR,K] d conf_coefficient output, may be NULL
    float *G8;                                           // per-row [rows, 8] contributions (summed per point by a segment sum: no atomics)
};

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(GatherBwdArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const int s = (int)(t / a.K), kk = (int)(t - (int64_t)s * a.K);
    if (s >= n_valid || kk >= a.vs_cnt[s]) return;
    const int item = a.vs_item[s];
    const size_t row = (size_t)a.vs_off[s] + kk, e = (size_t)item * a.K + kk;
    const int pid = a.pidx[e];
    const int ray = item / a.SR;
    const float *g = a.gX3 + row * a.ldg3 + 256;
    const float gd = g[6];
    // w_agg = w_norm * clamp_ST(conf): the clamp passes the gradient through unchanged (gradiant_clamp, :1422-1424)
    float gc = a.g_wagg[row] * a.weight[e];
    if (a.g_conf_out) gc += a.g_conf_out[e];
    // the row's contribution [d color 3 | d dir 3 | d conf | 0] is parked; a segment sum over the rows sorted by touched point adds the rows of
    // every point in a fixed order (atomics into the point buffers would sum in a run-dependent order)
    float4 *o = reinterpret_cast<float4 *>(a.G8 + row * 8);
    o[0] = make_float4(g[0], g[1], g[2], g[3] + gd * a.raydir[3 * (size_t)ray]);
    o[1] = make_float4(g[4] + gd * a.raydir[3 * (size_t)ray + 1], g[5] + gd * a.raydir[3 * (size_t)ray + 2], gc, 0.f);
    (void)pid;
}

// [U, 8] per-point sums -> the three point buffers (every touched point once: plain adds)
__global__ void point_small_grads_kernel(const float *__restrict__ P8, const int32_t *__restrict__ ulist, int U, float *__restrict__ g_conf,
                                         float *__restrict__ g_dir, float *__restrict__ g_color, const long long *__restrict__ d_n = nullptr)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n && *d_n < U) U = (int)*d_n;
    if (u >= U) return;
    const float4 a0 = reinterpret_cast<const float4 *>(P8 + (size_t)u * 8)[0], a1 = reinterpret_cast<const float4 *>(P8 + (size_t)u * 8)[1];
    const size_t p = (size_t)ulist[u];
    g_color[3 * p] += a0.x; g_color[3 * p + 1] += a0.y; g_color[3 * p + 2] += a0.z;
    g_dir[3 * p] += a0.w; g_dir[3 * p + 1] += a1.x; g_dir[3 * p + 2] += a1.y;
    g_conf[p] += a1.z;
}

// d emb from d [emb | PE3(emb)] rows of the touched points; E holds the forward sin/cos values
template <int F>
__global__ __launch_bounds__(256) void point_rows_bwd_kernel(const float *__restrict__ gE, int ldg, const float *__restrict__ E, int lde,
                                                             const int32_t *__restrict__ ids, int n, float *__restrict__ g_emb,
                                                             const long long *__restrict__ d_n = nullptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int u = (int)(t / F), d = (int)(t - (int64_t)u * F);
    if (d_n && *d_n < n) n = (int)*d_n;
    if (u >= n) return;
    const float *g = gE + (size_t)u * ldg, *e = E + (size_t)u * lde;
    float acc = g[d];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int col = F + 2 * (3 * d + f);
        // d sin(x 2^f) = cos(x 2^f) 2^f dx,  d cos(x 2^f) = -sin(x 2^f) 2^f dx
        acc += (float)(1 << f) * (e[col + 1] * g[col] - e[col] * g[col + 1]);
    }
    const int p = ids ? ids[u] : u;
    g_emb[(size_t)p * F + d] += acc;
}

// ------------------------------------------------------------------------------------------------ small helpers
// block maximum of |v| -> absmax_publish (hnr_common.h)
__device__ __forceinline__ void block_absmax(float mx, unsigned *absmax)
{
    __shared__ float s_mx[4];
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) absmax_publish(absmax, fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3])));
}

// g[m, n] = (g[m, n] + (n < n_add ? add[m, n] : 0)) * (y[m, n] > 0 ? 1 : slope); optionally max |g| -> absmax.  float4 per lane, a fixed grid
// striding over the rows the device count leaves (one lane per element over the CAPACITY was 125 k workgroups, most of them empty, a 64-bit
// division per element, and two more launches for the addend and the maximum).
__global__ __launch_bounds__(256) void dleaky_add_kernel(float *__restrict__ g, int ldg, const float *__restrict__ add, int lda, int n_add, const float *__restrict__ y, int ldy,
                                                         int64_t M, int N, float slope, const long long *__restrict__ d_n, unsigned *__restrict__ absmax)
{
    if (d_n && *d_n < M) M = *d_n;
    const int n4 = N >> 2;
    float mx = 0.f;
    // (n4 divides 256 in every use: a thread keeps its column and steps over the rows -- no division in the loop)
    const bool fixed_col = (256 % n4) == 0;
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, m0 = t0 / n4, m_step = fixed_col ? (int64_t)gridDim.x * 256 / n4 : 0;
    const int c0 = 4 * (int)(t0 - m0 * n4);
    int64_t it = 0;
    for (int64_t t = t0; t < M * n4; t += (int64_t)gridDim.x * 256, ++it) {
        const int64_t m = fixed_col ? m0 + it * m_step : t / n4;
        const int c = fixed_col ? c0 : 4 * (int)(t - m * n4);
        float4 v = *reinterpret_cast<const float4 *>(g + (size_t)m * ldg + c);
        const float4 yy = *reinterpret_cast<const float4 *>(y + (size_t)m * ldy + c);
        if (add && c < n_add) {
            const float4 ad = *reinterpret_cast<const float4 *>(add + (size_t)m * lda + c);
            v.x += ad.x; v.y += c + 1 < n_add ? ad.y : 0.f; v.z += c + 2 < n_add ? ad.z : 0.f; v.w += c + 3 < n_add ? ad.w : 0.f;
        }
        if (!(yy.x > 0.f)) v.x *= slope;
        if (!(yy.y > 0.f)) v.y *= slope;
        if (!(yy.z > 0.f)) v.z *= slope;
        if (!(yy.w > 0.f)) v.w *= slope;
        *reinterpret_cast<float4 *>(g + (size_t)m * ldg + c) = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (absmax) block_absmax(mx, absmax);
}

// out[s, :] = sum_v in[v * cap + s, :]; optionally max |out| -> absmax
__global__ __launch_bounds__(256) void sum_views_kernel(const float *__restrict__ in, int ldi, int V, int cap, int n_samples, int N, float *__restrict__ out, int ldo,
                                                        const long long *__restrict__ d_n, unsigned *__restrict__ absmax)
{
    if (d_n && *d_n < n_samples) n_samples = (int)*d_n;
    const int n4 = N >> 2;
    float mx = 0.f;
    const bool fixed_col = (256 % n4) == 0;
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, s0 = t0 / n4, s_step = fixed_col ? (int64_t)gridDim.x * 256 / n4 : 0;
    const int c0 = 4 * (int)(t0 - s0 * n4);
    int64_t it = 0;
    for (int64_t t = t0; t < (int64_t)n_samples * n4; t += (int64_t)gridDim.x * 256, ++it) {
        const int64_t s = fixed_col ? s0 + it * s_step : t / n4;
        const int c = fixed_col ? c0 : 4 * (int)(t - s * n4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int v = 0; v < V; ++v) {
            const float4 x = *reinterpret_cast<const float4 *>(in + ((size_t)v * cap + s) * ldi + c);
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        *reinterpret_cast<float4 *>(out + (size_t)s * ldo + c) = acc;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    }
    if (absmax) block_absmax(mx, absmax);
}

// Unique touched points: flags -> exclusive scan (two-level, deterministic) -> compact list + per-row compact index.
__global__ void mark_points_kernel(const int32_t *__restrict__ row_pid, int64_t M, int32_t *__restrict__ flags, const long long *__restrict__ d_n = nullptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n && *d_n < M) M = *d_n;
    if (t < M && (!d_n || row_pid[t] >= 0)) atomicAdd(flags + row_pid[t], 1);          // the point's NUMBER of rows (device-count form: rows of empty neighbour slots carry -1)
}

// flags[i] = number of rows of point i (0: untouched).  Per 1024-point block: the touched points and the rows -- the two prefix sums of
// flag_scan_kernel give a touched point its compact index AND the start of its rows in the point-major row list.
__global__ __launch_bounds__(1024) void flag_block_sum_kernel(const int32_t *__restrict__ flags, int n, int32_t *__restrict__ block_sums)
{
    __shared__ int s_a[16], s_b[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int c = i < n ? flags[i] : 0, v = c > 0 ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); c += __shfl_xor(c, o); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = v; s_b[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; }
        block_sums[2 * blockIdx.x] = a; block_sums[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(1024) void flag_scan_kernel(int32_t *__restrict__ flags /* in: rows per point, out: compact index or -1 */, int n,
                                                         const int32_t *__restrict__ block_sums, int32_t *__restrict__ ulist, int cap,
                                                         int32_t *__restrict__ count, int32_t *__restrict__ seg_start /* [cap + 1]: first row-list entry of compact point u */)
{
    __shared__ int s_a[16], s_b[16];
    __shared__ int s_base, s_base_b;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    int pa = 0, pb = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) { pa += block_sums[2 * k]; pb += block_sums[2 * k + 1]; }
    for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o); pb += __shfl_xor(pb, o); }
    if (lane == 0) { s_a[wid] = pa; s_b[wid] = pb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 16; ++k) { a += s_a[k]; b += s_b[k]; }
        s_base = a; s_base_b = b;
    }
    __syncthreads();
    const int c = i < n ? flags[i] : 0, v = c > 0 ? 1 : 0;
    int ia = v, ib = c;
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(ia, o), tb = __shfl_up(ib, o);
        if (lane >= o) { ia += ta; ib += tb; }
    }
    __syncthreads();
    if (lane == 63) { s_a[wid] = ia; s_b[wid] = ib; }
    __syncthreads();
    int oa = s_base + ia - v, ob = s_base_b + ib - c;
    for (int k = 0; k < wid; ++k) { oa += s_a[k]; ob += s_b[k]; }
    if (i < n) {
        flags[i] = v ? oa : -1;
        if (v && oa < cap) { ulist[oa] = i; if (seg_start) seg_start[oa] = ob; }
        if (i == n - 1) { *count = oa + v; if (seg_start && oa + v <= cap) seg_start[oa + v] = ob + c; }
    }
}

__global__ void map_rows_kernel(const int32_t *__restrict__ row_pid, int64_t M, const int32_t *__restrict__ uidx, int32_t *__restrict__ row_u,
                                const long long *__restrict__ d_n = nullptr, int skip_key = -1, const int32_t *__restrict__ seg_start = nullptr,
                                int32_t *__restrict__ cursor = nullptr, int32_t *__restrict__ row_list = nullptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!d_n) { if (t < M) row_u[t] = uidx[row_pid[t]]; return; }
    // device-count form: every row of the capacity gets a key; rows past *d_n and empty slots (-1) get the sentinel `skip_key` (> every compact
    // index: sorted last, outside every segment)
    if (t < M) {
        const int u = (t < *d_n && row_pid[t] >= 0) ? uidx[row_pid[t]] : skip_key;
        row_u[t] = u;
        // the point-major row list: entry order inside a point's segment is whatever the atomics give -- the consumers sort a segment's rows
        // before they add them (segment_sum_rows_csr_kernel), so the sums stay bit-identical run to run
        if (row_list && u != skip_key && u < skip_key) row_list[seg_start[u] + atomicAdd(cursor + u, 1)] = (int32_t)t;
    }
}

}  // namespace hnr

using namespace hnr;

// 16-wave blocks, two per CU at most: the loops are latency-bound (one sample per wave and iteration) and every block ends with one atomic per weight
static int persistent_blocks(int64_t n_waves_wanted)
{
    int64_t b = (n_waves_wanted + 15) / 16;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

// ================================================================================== C ABI
extern "C" int hnr_composite_bwd(const float *d_decoded, const float *d_sample_loc_w, const int32_t *d_sample_pidx, const int8_t *d_ray_mask,
                                 const int32_t *d_ray_nsamp, const float *d_campos, const float *d_camrot, const float *d_bg_color,
                                 int R, int SR, int K, float vsize_z, int raydist_mode_unit, const float *d_g_raycolor,
                                 float *d_g_decoded, void *stream)
{
    if (R < 0 || SR <= 0 || K <= 0) { set_error("hnr_composite_bwd: bad sizes"); return HNR_ERR_BADARG; }
    if (R == 0) return HNR_OK;
    if (!d_decoded || !d_sample_loc_w || !d_sample_pidx || !d_ray_mask || !d_campos || !d_camrot || !d_bg_color || !d_g_raycolor || !d_g_decoded) {
        set_error("hnr_composite_bwd: NULL argument"); return HNR_ERR_BADARG;
    }
    CompositeBwdArgs a;
    a.decoded = d_decoded; a.loc_w = d_sample_loc_w; a.pidx = d_sample_pidx; a.ray_mask = d_ray_mask; a.nsamp = d_ray_nsamp;
    a.campos = d_campos; a.camrot = d_camrot; a.bg = d_bg_color; a.R = R; a.SR = SR; a.K = K; a.vsize_z = vsize_z;
    a.unit_mode = raydist_mode_unit; a.g_raycolor = d_g_raycolor; a.g_decoded = d_g_decoded;
    if (SR <= 64) composite_bwd_kernel<<<cdiv((int64_t)R * 64, 256), 256, 0, (hipStream_t)stream>>>(a);
    else composite_bwd_serial_kernel<<<cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_final_color_bwd(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                                   const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples, const float *d_g_decoded,
                                   float *d_gY, int ldgy, float *d_gCF, int ldgcf, float *d_g_sigma, float *d_g_w_fin, float *d_g_b_fin,
                                   void *stream)
{
    return hnr::final_color_bwd_max(d_Y, ldy, d_CF, ldcf, d_w_fin, d_b_fin, d_vs_item, d_counts, cap_samples, d_g_decoded, d_gY, ldgy, d_gCF, ldgcf, d_g_sigma, d_g_w_fin,
                                    d_g_b_fin, nullptr, stream);
}

// csrc/render_train.hip: the same, and max |d_gY| into *d_gY_max (atomicMax on the bit pattern; hnr_absmax's convention) -- one launch less
int hnr::final_color_bwd_max(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                             const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples, const float *d_g_decoded,
                             float *d_gY, int ldgy, float *d_gCF, int ldgcf, float *d_g_sigma, float *d_g_w_fin, float *d_g_b_fin,
                             uint32_t *d_gY_max, void *stream)
{
    if (!d_Y || !d_CF || !d_w_fin || !d_b_fin || !d_vs_item || !d_counts || !d_g_decoded || !d_gY || !d_gCF || !d_g_sigma || !d_g_w_fin ||
        !d_g_b_fin || ldy < 45 || ldcf < 128 || ldgy < 45 || ldgcf < 128) {
        set_error("hnr_final_color_bwd: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    FinalBwdArgs a;
    a.Y = d_Y; a.ldy = ldy; a.CF = d_CF; a.ldcf = ldcf; a.w_fin = d_w_fin; a.b_fin = d_b_fin; a.vs_item = d_vs_item;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.g_decoded = d_g_decoded; a.gY = d_gY; a.ldgy = ldgy;
    a.gCF = d_gCF; a.ldgcf = ldgcf; a.g_sigma = d_g_sigma; a.g_w_fin = d_g_w_fin; a.g_b_fin = d_g_b_fin; a.gY_max = d_gY_max;
    final_color_bwd_kernel<<<persistent_blocks(cap_samples), 1024, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_merge_bwd(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last,
                             const float *d_vmask, const float *d_frame_w, const int64_t *d_counts, int V, int cap_samples, float slope,
                             const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR, const float *d_gX7, int ldg7,
                             float *d_gF, int ldgf, float *d_gZ3, int ldgz, float *d_gCF, int ldgcf, float *d_g_w_last, float *d_g_b_last,
                             void *stream)
{
    return hnr::merge_bwd_max(d_X6, ld6, d_Hm, ldh, d_w_last, d_b_last, d_vmask, d_frame_w, d_counts, V, cap_samples, slope, d_ray_drop, d_vs_item, SR, d_gX7, ldg7, d_gF, ldgf,
                              d_gZ3, ldgz, d_gCF, ldgcf, d_g_w_last, d_g_b_last, nullptr, stream);
}

// csrc/render_train.hip: the same, and max |d_gZ3| into *d_gZ3_max
int hnr::merge_bwd_max(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last,
                       const float *d_vmask, const float *d_frame_w, const int64_t *d_counts, int V, int cap_samples, float slope,
                       const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR, const float *d_gX7, int ldg7,
                       float *d_gF, int ldgf, float *d_gZ3, int ldgz, float *d_gCF, int ldgcf, float *d_g_w_last, float *d_g_b_last,
                       uint32_t *d_gZ3_max, void *stream)
{
    if (!d_X6 || !d_Hm || !d_w_last || !d_b_last || !d_vmask || !d_counts || !d_vs_item || !d_gX7 || !d_gF || !d_gZ3 || !d_gCF ||
        !d_g_w_last || !d_g_b_last || V <= 0 || V > MAXV || ldh < 64 || ldg7 < 90 || ldgf < 48 || ldgz < 64 || ldgcf < 45 || SR <= 0) {
        set_error("hnr_merge_bwd: bad argument (V <= %d)", MAXV); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    MergeBwdArgs a;
    a.X6 = d_X6; a.ld6 = ld6; a.Hm = d_Hm; a.ldh = ldh; a.w_last = d_w_last; a.b_last = d_b_last; a.vmask = d_vmask; a.frame_w = d_frame_w;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.V = V; a.cap = cap_samples; a.slope = slope;
    a.ray_drop = d_ray_drop; a.vs_item = d_vs_item; a.SR = SR; a.gX7 = d_gX7; a.ldg7 = ldg7; a.gF = d_gF; a.ldgf = ldgf;
    a.gZ3 = d_gZ3; a.ldgz = ldgz; a.gCF = d_gCF; a.ldgcf = ldgcf; a.g_w_last = d_g_w_last; a.g_b_last = d_g_b_last; a.gZ3_max = d_gZ3_max;
    if (V <= 4) merge_bwd_kernel<4, 16><<<persistent_blocks(cap_samples), 1024, 0, (hipStream_t)stream>>>(a);
    else merge_bwd_kernel<MAXV, 4><<<persistent_blocks(cap_samples), 256, 0, (hipStream_t)stream>>>(a);      // (more than 4 views: twice the registers per lane, 4-wave blocks)
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

static inline int conv_out(int n) { return (n + 2 - 3) / 2 + 1; }

extern "C" int64_t hnr_sort_rows_scratch_bytes(int64_t M);
extern "C" int hnr_sort_rows_by_key(const int32_t *d_keys, int64_t M, int32_t *d_keys_sorted, int32_t *d_perm, void *d_scratch,
                                    int64_t scratch_bytes, void *stream);
extern "C" int hnr_segment_sum_rows(const float *d_A, int lda, const float *d_B, int ldb, const int32_t *d_keys_sorted,
                                    const int32_t *d_perm, int64_t M, int n_cols, float *d_dst, int64_t dst_stride, void *stream);

extern "C" int hnr_proj_rows_bwd(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                                 const float *d_intrinsic, int V, int H, int W, int cap_samples, const float *d_gFa, int lda,
                                 const float *d_gFb, int ldb, float *d_g_featmap, int32_t *d_bbox, float *d_g_pyramid,
                                 int32_t *d_key_scratch, void *d_sort_scratch, int64_t sort_scratch_bytes, void *stream)
{
    if (!d_sample_loc_w || !d_vs_item || !d_counts || !d_w2c || !d_intrinsic || !d_gFa || !d_g_featmap || !d_bbox || !d_g_pyramid ||
        !d_key_scratch || !d_sort_scratch || V <= 0 || H <= 1 || W <= 1 || lda < 48 || (lda & 3) || (d_gFb && (ldb < 48 || (ldb & 3))) ||
        (int64_t)V * H * W >= (1ll << 31)) {
        set_error("hnr_proj_rows_bwd: bad argument (row strides >= 48, multiples of 4)"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)V * cap_samples;
    int32_t *keys = d_key_scratch, *keys_sorted = keys + rows, *perm = keys_sorted + rows;
    ProjBwdArgs a;
    a.loc_w = d_sample_loc_w; a.vs_item = d_vs_item; a.counts = reinterpret_cast<const unsigned long long *>(d_counts);
    a.w2c = d_w2c; a.Kmat = d_intrinsic; a.V = V; a.H = H; a.W = W; a.cap = cap_samples; a.keys = keys; a.bbox = d_bbox;
    // rows past counts[SAMPLES_VALID] (none when cap_samples is exact) must not carry stale keys
    HNR_HIP_CHECK(hipMemsetAsync(keys, 0xff, (size_t)rows * 4, st));
    proj_rows_bwd_kernel<<<cdiv(rows, 1024), 1024, 0, st>>>(a);
    HNR_LAUNCH_CHECK();
    (void)keys_sorted; (void)perm; (void)d_sort_scratch; (void)sort_scratch_bytes;
    {
        const int64_t nb = cdiv(rows * 48, 256);
        pixel_scatter_add_kernel<<<(int)(nb < 4096 ? nb : 4096), 256, 0, st>>>(d_gFa, lda, d_gFb, ldb, keys, cap_samples, V, a.counts, d_g_featmap);
        HNR_LAUNCH_CHECK();
    }
    const int H1 = conv_out(H), W1 = conv_out(W), H2 = conv_out(H1), W2 = conv_out(W1), H3 = conv_out(H2), W3 = conv_out(W2);
    // same layout as the forward scratch of hnr_image_features: s1a s1 s2a s2 s3a s3
    const size_t n1 = (size_t)V * 6 * H1 * W1, n2 = (size_t)V * 12 * H2 * W2, n3 = (size_t)V * 24 * H3 * W3;
    float *g1 = d_g_pyramid + n1, *g2 = d_g_pyramid + 2 * n1 + n2, *g3 = d_g_pyramid + 2 * n1 + 2 * n2 + n3;
    UpsampleBwdArgs ua;
    ua.g_fm = d_g_featmap; ua.bbox = d_bbox; ua.V = V; ua.H = H; ua.W = W;
    ua.Hs[0] = H1; ua.Ws[0] = W1; ua.C[0] = 6; ua.c0[0] = 3; ua.g[0] = g1; ua.nb[0] = cdiv((int64_t)n1, 256);
    ua.Hs[1] = H2; ua.Ws[1] = W2; ua.C[1] = 12; ua.c0[1] = 9; ua.g[1] = g2; ua.nb[1] = cdiv((int64_t)n2, 256);
    ua.Hs[2] = H3; ua.Ws[2] = W3; ua.C[2] = 24; ua.c0[2] = 21; ua.g[2] = g3; ua.nb[2] = cdiv((int64_t)n3, 256);
    // level 3 first in block order? its cells loop over the most pixels: the slowest blocks should start first -- levels are dispatched 1, 2, 3 by
    // block index, so level 3 is given the LOWEST indices by swapping the roles
    { std::swap(ua.Hs[0], ua.Hs[2]); std::swap(ua.Ws[0], ua.Ws[2]); std::swap(ua.C[0], ua.C[2]); std::swap(ua.c0[0], ua.c0[2]); std::swap(ua.g[0], ua.g[2]); std::swap(ua.nb[0], ua.nb[2]); }
    upsample_bwd_kernel<<<ua.nb[0] + ua.nb[1] + ua.nb[2], 256, 0, st>>>(ua);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

static int image_features_bwd_impl(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope,
                                   const float *d_scratch, float *d_g_pyramid, float *const *g_conv_w, float *const *g_conv_b,
                                   const int32_t *d_bbox, void *stream);

extern "C" int hnr_image_features_bwd(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope,
                                      const float *d_scratch, float *d_g_pyramid, float *const *g_conv_w, float *const *g_conv_b,
                                      void *stream)
{
    return image_features_bwd_impl(d_img, V, H, W, conv_w, slope, d_scratch, d_g_pyramid, g_conv_w, g_conv_b, nullptr, stream);
}

// csrc/render_train.hip: the same with the per-view rectangle of touched pixels (hnr_proj_rows_bwd's d_bbox): every kernel skips what lies outside
// its image at the level's resolution, dilated by more than the 3x3 / stride-2 chain can spread a gradient
namespace hnr {
int image_features_bwd_bbox(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope, const float *d_scratch, float *d_g_pyramid,
                            float *const *g_conv_w, float *const *g_conv_b, const int32_t *d_bbox, void *stream)
{
    return image_features_bwd_impl(d_img, V, H, W, conv_w, slope, d_scratch, d_g_pyramid, g_conv_w, g_conv_b, d_bbox, stream);
}
}

static int image_features_bwd_impl(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope,
                                   const float *d_scratch, float *d_g_pyramid, float *const *g_conv_w, float *const *g_conv_b,
                                   const int32_t *d_bbox, void *stream)
{
    if (!d_img || !conv_w || !d_scratch || !d_g_pyramid || !g_conv_w || !g_conv_b || V <= 0 || H <= 1 || W <= 1) {
        set_error("hnr_image_features_bwd: bad argument"); return HNR_ERR_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int H1 = conv_out(H), W1 = conv_out(W), H2 = conv_out(H1), W2 = conv_out(W1), H3 = conv_out(H2), W3 = conv_out(W2);
    const size_t n1 = (size_t)V * 6 * H1 * W1, n2 = (size_t)V * 12 * H2 * W2, n3 = (size_t)V * 24 * H3 * W3;
    const float *s1a = d_scratch, *s1 = s1a + n1, *s2a = s1 + n1, *s2 = s2a + n2, *s3a = s2 + n2, *s3 = s3a + n3;
    float *g1a = d_g_pyramid, *g1 = g1a + n1, *g2a = g1 + n1, *g2 = g2a + n2, *g3a = g2 + n2, *g3 = g3a + n3;
    // resolution divisor and dilation of a tensor at pyramid level 1 / 2 / 3 (upsample +-1, every 3x3 conv +-1, every stride-2 step x2 + 1)
    auto lvl_f = [&](int Hl) { return Hl == H1 ? 2 : (Hl == H2 ? 4 : 8); };
    auto lvl_halo = [&](int Hl) { return Hl == H1 ? 30 : (Hl == H2 ? 14 : 6); };
    // conv5: s3a -> s3 ; conv4: s2 -> s3a (stride 2) ; conv3: s2a -> s2 ; conv2: s1 -> s2a (stride 2) ; conv1: s1a -> s1 ; conv0: img -> s1a
    auto tile = [&](const float *g_out, const float *out, const float *in, int cstride, int Hin, int Win, int Hout, int Wout, int li, float *g_in) {
        ConvTileArgs a;
        a.g_out = g_out; a.out = out; a.w = conv_w[li]; a.in = in; a.in_cstride = cstride; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.slope = slope;
        a.g_in = g_in; a.g_w = g_conv_w[li]; a.g_b = g_conv_b[li]; a.V = V; a.bbox = d_bbox; a.f_out = lvl_f(Hout); a.halo_out = lvl_halo(Hout);
        a.tiles_x = a.tiles_y = 0;
        return a;
    };
    { const int rc = conv3x3_bwd_tile_launch<24, 24, 1, 8, true, false>(tile(g3, s3, s3a, 0, H3, W3, H3, W3, 5, g3a), st); if (rc != HNR_OK) return rc; }
    { const int rc = conv3x3_bwd_tile_launch<12, 24, 2, 8, true, false>(tile(g3a, s3a, s2, 0, H2, W2, H3, W3, 4, g2), st); if (rc != HNR_OK) return rc; }
    { const int rc = conv3x3_bwd_tile_launch<12, 12, 1, 16, true, false>(tile(g2, s2, s2a, 0, H2, W2, H2, W2, 3, g2a), st); if (rc != HNR_OK) return rc; }
    { const int rc = conv3x3_bwd_tile_launch<6, 12, 2, 16, true, false>(tile(g2a, s2a, s1, 0, H1, W1, H2, W2, 2, g1), st); if (rc != HNR_OK) return rc; }
    { const int rc = conv3x3_bwd_tile_launch<6, 6, 1, 16, true, false>(tile(g1, s1, s1a, 0, H1, W1, H1, W1, 1, g1a), st); if (rc != HNR_OK) return rc; }
    { const int rc = conv3x3_bwd_tile_launch<3, 6, 2, 16, false, true>(tile(g1a, s1a, d_img, 3, H, W, H1, W1, 0, nullptr), st); if (rc != HNR_OK) return rc; }
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

extern "C" int hnr_gather_rows_bwd_rows(const int32_t *d_sample_pidx, const float *d_raydir, const int32_t *d_vs_item, const int32_t *d_vs_off,
                                        const int32_t *d_vs_cnt, const int64_t *d_counts, int SR, int K, int cap_samples, const float *d_gX3,
                                        int ldg3, const float *d_g_wagg, const float *d_weight, const float *d_g_conf_out, float *d_G8, void *stream)
{
    if (!d_sample_pidx || !d_raydir || !d_vs_item || !d_vs_off || !d_vs_cnt || !d_counts || !d_gX3 || !d_g_wagg || !d_weight || !d_G8 ||
        ((uintptr_t)d_G8 & 15) || ldg3 < 263 || SR <= 0 || K <= 0) {
        set_error("hnr_gather_rows_bwd_rows: bad argument"); return HNR_ERR_BADARG;
    }
    if (cap_samples <= 0) return HNR_OK;
    GatherBwdArgs a;
    a.pidx = d_sample_pidx; a.raydir = d_raydir; a.vs_item = d_vs_item; a.vs_off = d_vs_off; a.vs_cnt = d_vs_cnt;
    a.counts = reinterpret_cast<const unsigned long long *>(d_counts); a.SR = SR; a.K = K; a.gX3 = d_gX3; a.ldg3 = ldg3; a.g_wagg = d_g_wagg;
    a.weight = d_weight; a.g_conf_out = d_g_conf_out; a.G8 = d_G8;
    gather_rows_bwd_kernel<<<cdiv((int64_t)cap_samples * K, 256), 256, 0, (hipStream_t)stream>>>(a);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}

// ------------------------------------------------------------------------------------------------ device-count forms (csrc/render_train.hip)
// The same kernels with their work sizes read on the device (*d_n, clamped to the capacity the grid is sized for): the single-call training
// step never learns a count on the host.
namespace hnr {
int unique_points_dc(const int32_t *d_row_pid, int64_t M_cap, const long long *d_m, int n_points, int32_t *d_uidx, int32_t *d_ulist, int cap,
                     int32_t *d_row_u, int32_t *d_count, int32_t *d_scratch /* 2 x ceil(n_points / 1024) */, int32_t *d_seg_start /* [cap + 1] */,
                     int32_t *d_seg_count /* [cap] */, int32_t *d_row_list /* [M_cap] */, hipStream_t st)
{
    HNR_HIP_CHECK(hipMemsetAsync(d_uidx, 0, (size_t)n_points * 4, st));
    if (d_seg_count) HNR_HIP_CHECK(hipMemsetAsync(d_seg_count, 0, (size_t)cap * 4, st));
    if (M_cap > 0) mark_points_kernel<<<cdiv(M_cap, 256), 256, 0, st>>>(d_row_pid, M_cap, d_uidx, d_m);
    const int nb = cdiv(n_points, 1024);
    flag_block_sum_kernel<<<nb, 1024, 0, st>>>(d_uidx, n_points, d_scratch);
    flag_scan_kernel<<<nb, 1024, 0, st>>>(d_uidx, n_points, d_scratch, d_ulist, cap, d_count, d_seg_start);
    if (M_cap > 0) map_rows_kernel<<<cdiv(M_cap, 256), 256, 0, st>>>(d_row_pid, M_cap, d_uidx, d_row_u, d_m, cap, d_seg_start, d_seg_count, d_row_list);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
int point_small_grads_dc(const float *d_P8, const int32_t *d_ulist, int U_cap, const long long *d_u, float *d_g_conf, float *d_g_dir, float *d_g_color, hipStream_t st)
{
    if (U_cap <= 0) return HNR_OK;
    point_small_grads_kernel<<<cdiv(U_cap, 256), 256, 0, st>>>(d_P8, d_ulist, U_cap, d_g_conf, d_g_dir, d_g_color, d_u);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
int point_rows_bwd_dc(const float *d_gE, int ldg, const float *d_E, int lde, const int32_t *d_ids, int n_cap, const long long *d_n, float *d_g_emb, hipStream_t st)
{
    if (n_cap <= 0) return HNR_OK;
    point_rows_bwd_kernel<32><<<cdiv((int64_t)n_cap * 32, 256), 256, 0, st>>>(d_gE, ldg, d_E, lde, d_ids, n_cap, d_g_emb, d_n);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
int dleaky_add_dc(float *d_g, int ldg, const float *d_add, int lda, int n_add, const float *d_y, int ldy, int64_t M_cap, const long long *d_m, int N, float slope,
                  uint32_t *d_absmax, hipStream_t st)
{
    if (M_cap <= 0) return HNR_OK;
    if ((N & 3) || (ldg & 3) || (ldy & 3) || (d_add && (lda & 3)) || ((uintptr_t)d_g & 15) || ((uintptr_t)d_y & 15) || ((uintptr_t)d_add & 15)) {
        set_error("dleaky_add: rows must be 16-byte aligned"); return HNR_ERR_BADARG;
    }
    int64_t nb = cdiv(M_cap * (N / 4), 256);
    dleaky_add_kernel<<<(int)(nb < 1024 ? nb : 1024), 256, 0, st>>>(d_g, ldg, d_add, lda, n_add, d_y, ldy, M_cap, N, slope, d_m, d_absmax);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
int sum_views_dc(const float *d_in, int ldi, int V, int cap, const long long *d_n, int N, float *d_out, int ldo, uint32_t *d_absmax, hipStream_t st)
{
    if (cap <= 0) return HNR_OK;
    if ((N & 3) || (ldi & 3) || (ldo & 3) || ((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) { set_error("sum_views: rows must be 16-byte aligned"); return HNR_ERR_BADARG; }
    int64_t nb = cdiv((int64_t)cap * (N / 4), 256);
    sum_views_kernel<<<(int)(nb < 1024 ? nb : 1024), 256, 0, st>>>(d_in, ldi, V, cap, cap, N, d_out, ldo, d_n, d_absmax);
    HNR_LAUNCH_CHECK();
    return HNR_OK;
}
}  // namespace hnr
