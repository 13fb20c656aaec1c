// The training step of the hybrid render path as TWO library calls: hnr_render_train_forward (one pass of NeuralPointsRayMarching.forward in
// train mode + fill_invalid, /root/reference/models/neural_points_volumetric_model.py:257-427, :87-126, with every activation the backward
// pass needs kept in the caller's workspace) and hnr_render_train_backward (what loss.backward() makes torch autograd compute for it,
// models/mvs_points_volumetric_model.py:111-131: gradients of the four trainable point buffers and of every aggregator parameter on the
// order-2 hybrid path).  Every launch is issued here, back to back on the caller's stream; all work sizes (valid samples, neighbour rows,
// touched points) are read from device counters, the buffers are carved from ONE workspace sized for `cap_samples` valid shading samples,
// and nothing is read back: the reference's three host synchronisations per chunk (query_point_indices_worldcoords.py:645-646, :705;
// point_aggregators.py:1092) and the per-step host reads of the round-2 training path have no counterpart.
//
// Row layout of the per-neighbour tensors: 8 row slots per valid shading sample (row = 8 s + k, the chain kernel's layout); slots without a
// neighbour carry weight 0, so their gradient rows are exact zeros and add nothing to any sum.
//
// Dense layers: forward = the fused f16x2 kernels of the render path (csrc/chain.hip in its activation-keeping form, csrc/mlp.hip);
// input gradients = hnr_h2lin with the transposed weights, weight / bias gradients = hnr_h2wgrad (csrc/h2gemm.hip); the weights change every
// step, so their kernel images are re-packed at the top of each call.
#include <mutex>
#include <limits.h>

#include "chain_defs.h"
#include <stdlib.h>

#include "train_internal.h"

using namespace hnr;

namespace {

// ---------------------------------------------------------------------------------------------------------------- workspace
struct Carver {
    char *base; size_t off, cap; bool ok;
    template <class T> T *take(size_t n)
    {
        off = (off + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(base + off);
        off += n * sizeof(T);
        if (base && off > cap) ok = false;
        return base ? p : nullptr;
    }
};

enum {  // slots of the abs-max table (uint32 bit patterns): scales of the weight-gradient GEMMs
    AM_H1 = 0, AM_X3, AM_H3, AM_H4, AM_ONE, AM_T1, AM_T2, AM_CF, AM_M1, AM_M2, AM_M3, AM_Y1, AM_Y2, AM_Y3, AM_X5, AM_X6, AM_X7, AM_E,
    AM_gY3, AM_dY2, AM_dY1, AM_gZ3m, AM_dM2, AM_dM1, AM_gpre, AM_gCF, AM_dT2, AM_dT1, AM_gZ4, AM_dZ3, AM_dZ2, AM_dZ1, AM_gTu, AM_N
};
enum { TC_S = 0, TC_M8, TC_U, TC_N = 8 };      // int64 device counters of the step: valid samples, neighbour row slots (8 S), touched points
enum {  // h2lin images (input-gradient layers + the per-point table layer)
    IM_TAB = 0, IM_TABT, IM_B32T, IM_B30T, IM_B12T, IM_CF2T, IM_CF1T, IM_CF0T, IM_MW2T, IM_MW1T, IM_MW0FDT, IM_MW0CFT, IM_MX2T, IM_MX1T, IM_MX0T, IM_N
};
const int IM_K[IM_N] = {224, 256, 256, 256, 256, 128, 128, 128, 64, 64, 64, 64, 45, 45, 45};

struct Layout {
    // saved by the forward pass
    int32_t *work, *vs_item, *vs_off, *vs_cnt, *scratch, *row_pid, *row_u, *uidx, *ulist, *ucount, *uscratch, *row_s;
    uint8_t *ray_drop;
    long long *tc;
    uint32_t *amax;
    char *chain_ws, *img_chain, *img_cf, *img_mw, *img_mx, *img[IM_N];
    uint32_t *hbits;
    float *W0fd, *Xd, *H1, *X3, *H3, *H4, *E, *Tu, *X5, *sigma, *T1, *T2, *CF, *pre, *X6, *vmask, *M1, *M2, *M3, *X7, *Y1, *Y2, *Y3, *fm, *fm_scratch;
    // backward temporaries
    float *g_dec, *gY3, *gCF, *g_sigma, *dY2, *dY1, *gX7, *gF, *gZ3m, *dM2, *dM1, *gX6, *gpre, *tmpCF, *tmpWfd, *g_pyr, *g_fm, *dT2, *dT1, *gX5, *gZ4, *g_wagg,
          *dZ3, *gX3, *dZ1, *G8, *P8, *gTu, *gE;
    int32_t *bbox, *key_scratch, *row_list, *seg_cnt, *seg_start;
    char *sort_scratch, *wg_scratch, *wg_scratch2;                      // (wg_scratch2: the second weight-gradient queue of HNR_TRAIN_SIDE bit 5)
    float *conf0;
    size_t sort_bytes, wg_bytes;
    size_t rows_cap, ucap, VS, fm_elems, bytes;
};

Layout carve(void *ws, size_t ws_bytes, const hnr_train_params *p, bool *ok)
{
    Carver c{(char *)ws, 0, ws_bytes, true};
    Layout L;
    const size_t cap = (size_t)p->cap_samples, V = (size_t)p->V, N = (size_t)p->n_points, R = (size_t)p->R;
    const size_t tiles = (cap + 15) / 16 + 2;
    L.rows_cap = tiles * 128;
    L.ucap = L.rows_cap < N ? L.rows_cap : N;
    L.VS = V * cap;
    const size_t rows = L.rows_cap, ucap = L.ucap, VS = L.VS > 0 ? L.VS : 1;
    L.work = c.take<int32_t>((size_t)hnr_query_work_elems(p->R, p->SR));
    L.vs_item = c.take<int32_t>(cap + 1); L.vs_off = c.take<int32_t>(cap + 1); L.vs_cnt = c.take<int32_t>(cap + 1);
    L.scratch = c.take<int32_t>(3 * ((R * p->SR + 1023) / 1024) + 3);
    L.row_pid = c.take<int32_t>(rows); L.row_u = c.take<int32_t>(rows);
    L.uidx = c.take<int32_t>(N); L.ulist = c.take<int32_t>(ucap + 1); L.ucount = c.take<int32_t>(4); L.uscratch = c.take<int32_t>(2 * ((N + 1023) / 1024) + 2);
    L.row_s = c.take<int32_t>(VS + 1);
    L.ray_drop = c.take<uint8_t>(R);
    L.tc = c.take<long long>(TC_N);
    L.amax = c.take<uint32_t>(AM_N);
    L.chain_ws = c.take<char>((size_t)hnr_chain_workspace_bytes(p->cap_samples));
    L.img_chain = c.take<char>((size_t)hnr_chain_packed_bytes());
    const int cfK[4] = {280, 128, 128, 128}, mwK[3] = {48, 64, 64}, mxK[3] = {90, 45, 45};
    L.img_cf = c.take<char>((size_t)hnr_mlp3_packed_bytes(4, cfK));
    L.img_mw = c.take<char>((size_t)hnr_mlp3_packed_bytes(3, mwK));
    L.img_mx = c.take<char>((size_t)hnr_mlp3_packed_bytes(3, mxK));
    for (int i = 0; i < IM_N; ++i) L.img[i] = c.take<char>((size_t)hnr_h2lin_packed_bytes(IM_K[i]));
    L.W0fd = c.take<float>(64 * 48);
    L.Xd = c.take<float>(rows * 64);
    L.H1 = c.take<float>(rows * 256); L.X3 = c.take<float>(rows * 264); L.H3 = c.take<float>(rows * 256); L.H4 = c.take<float>(rows * 256);
    L.hbits = c.take<uint32_t>(3 * rows * 8);                           // signs of H1 / X3[:, :256] / H3: 256 bits per row (ChainArgs::hbits)
    L.E = c.take<float>(ucap * 224); L.Tu = c.take<float>(ucap * 256);
    L.X5 = c.take<float>(cap * 280); L.sigma = c.take<float>(cap + 1);
    L.T1 = c.take<float>(cap * 128); L.T2 = c.take<float>(cap * 128); L.CF = c.take<float>(cap * 128); L.pre = c.take<float>(cap * 64);
    L.X6 = c.take<float>(VS * 48); L.vmask = c.take<float>(VS + 1);
    L.M1 = c.take<float>(VS * 64); L.M2 = c.take<float>(VS * 64); L.M3 = c.take<float>(VS * 64);
    L.X7 = c.take<float>(cap * 92); L.Y1 = c.take<float>(cap * 48); L.Y2 = c.take<float>(cap * 48); L.Y3 = c.take<float>(cap * 48);
    L.fm_elems = V > 0 ? (size_t)hnr_image_features_scratch_elems(p->V, p->H, p->W) : 1;
    L.fm = c.take<float>(V > 0 ? V * p->H * p->W * 48 : 4); L.fm_scratch = c.take<float>(L.fm_elems);
    // ---- backward
    L.g_dec = c.take<float>(R * p->SR * 4);
    L.gY3 = c.take<float>(cap * 48); L.gCF = c.take<float>(cap * 128); L.g_sigma = c.take<float>(cap + 1);
    L.dY2 = c.take<float>(cap * 48); L.dY1 = c.take<float>(cap * 48); L.gX7 = c.take<float>(cap * 92);
    L.gF = c.take<float>(VS * 48); L.gZ3m = c.take<float>(VS * 64); L.dM2 = c.take<float>(VS * 64); L.dM1 = c.take<float>(VS * 64); L.gX6 = c.take<float>(VS * 48);
    L.gpre = c.take<float>(cap * 64); L.tmpCF = c.take<float>(cap * 128); L.tmpWfd = c.take<float>(64 * 48);
    L.g_pyr = c.take<float>(L.fm_elems); L.g_fm = c.take<float>(V > 0 ? V * p->H * p->W * 48 : 4);
    L.bbox = c.take<int32_t>(V > 0 ? 4 * V : 4); L.key_scratch = c.take<int32_t>(3 * VS);
    L.sort_bytes = 256; L.sort_scratch = c.take<char>(L.sort_bytes);                        // (hnr_proj_rows_bwd no longer sorts: a token buffer for its argument check)
    L.dT2 = c.take<float>(cap * 128); L.dT1 = c.take<float>(cap * 128); L.gX5 = c.take<float>(cap * 256);
    L.gZ4 = c.take<float>(rows * 256); L.g_wagg = c.take<float>(rows);
    L.dZ3 = c.take<float>(rows * 256); L.gX3 = c.take<float>(rows * 264); L.dZ1 = c.take<float>(rows * 256);
    L.row_list = c.take<int32_t>(rows); L.seg_cnt = c.take<int32_t>(ucap + 1); L.seg_start = c.take<int32_t>(ucap + 1);
    L.G8 = c.take<float>(rows * 8); L.P8 = c.take<float>(ucap * 8); L.gTu = c.take<float>(ucap * 256); L.gE = c.take<float>(ucap * 224);
    L.wg_bytes = (size_t)hnr_h2wgrad_scratch_bytes(256, 280); L.wg_scratch = c.take<char>(L.wg_bytes); L.wg_scratch2 = c.take<char>(L.wg_bytes);
    L.conf0 = c.take<float>(512);
    L.bytes = (c.off + 255) & ~(size_t)255;
    if (ok) *ok = c.ok;
    return L;
}

// ---------------------------------------------------------------------------------------------------------------- small kernels
// status word + the step's device counters.  tc[TC_S] = valid samples (clamped to the capacity), tc[TC_M8] = 8 x that.
__global__ void train_counts_kernel(unsigned long long *counts, int cap_samples, int32_t *status, long long *tc, uint32_t *amax)
{
    const unsigned long long nv = counts[HNR_CNT_SAMPLES_VALID];
    status[1] = (int32_t)nv;
    status[0] = nv > (unsigned long long)cap_samples ? 1 : 0;
    const unsigned long long s = nv > (unsigned long long)cap_samples ? (unsigned long long)cap_samples : nv;
    counts[HNR_CNT_SAMPLES_VALID] = s;
    tc[TC_S] = (long long)s; tc[TC_M8] = 8 * (long long)s; tc[TC_U] = 0;
    for (int i = 0; i < AM_N; ++i) amax[i] = 0u;
    amax[AM_ONE] = __float_as_uint(1.0f);               // |sin|, |cos| <= 1: block1.0's distance inputs
    amax[AM_X5] = __float_as_uint(1.0f);                // ... and X5's 24 direction-encoding columns; the chain kernel adds the maximum of its 256 feature sums
}

__global__ void train_ucount_kernel(const int32_t *ucount, long long ucap, long long *tc) { const long long u = ucount[0]; tc[TC_U] = u < ucap ? u : ucap; }

// vs_off / vs_cnt of the 8-slot row layout (consumed by the per-row backward kernels): off = 8 s, cnt = valid ids (a prefix of the K slots)
__global__ void train_vs_fill_kernel(const int32_t *__restrict__ vs_item, const int32_t *__restrict__ pidx, const unsigned long long *__restrict__ counts,
                                     int32_t *__restrict__ vs_off, int32_t *__restrict__ vs_cnt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (int)counts[HNR_CNT_SAMPLES_VALID]) return;
    const int32_t *p = pidx + (size_t)vs_item[s] * 8;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c += p[k] >= 0 ? 1 : 0;
    vs_off[s] = 8 * s; vs_cnt[s] = c;
}

// rays whose merged image feature is dropped (point_aggregators.py:1222-1237): the pattern is indexed by VALID-ray row (`drop_ray_flag[...]` over
// the R' compacted rays), so flags[r] = mask[r] && lut[number of valid rays before r]; or explicit per-ray flags ANDed with the mask.  One block.
__global__ __launch_bounds__(1024) void train_ray_drop_kernel(const int8_t *__restrict__ mask, int R, const uint8_t *__restrict__ lut, const uint8_t *__restrict__ explicit_flags,
                                                              uint8_t *__restrict__ out)
{
    __shared__ int s_w[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        const int m = (r < R && mask[r] > 0) ? 1 : 0;
        int inc = m;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wid] = inc;
        __syncthreads();
        int pre = s_base;
        for (int k = 0; k < wid; ++k) pre += s_w[k];
        if (r < R) out[r] = explicit_flags ? (uint8_t)(m && explicit_flags[r]) : (uint8_t)(m && lut && lut[pre + inc - m]);
        __syncthreads();
        if (threadIdx.x == 1023) s_base = pre + inc;
        __syncthreads();
    }
}

__global__ void train_conf_fill_kernel(const float *__restrict__ conf, float *__restrict__ out, long long n)
{
    // empty slots read point 0 through the reference's index clamp (neural_points.py:711)
    const float c = fminf(fmaxf(conf[0], 0.0001f), 1.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = c;
}

// aux_merge_weight_block.0's image-feature and direction columns [64, 45 | 3] <-> the contiguous [64, 48] matrix of the split first layer
__global__ void train_w0fd_kernel(const float *__restrict__ w0 /*[64,176]*/, float *__restrict__ out /*[64,48]*/)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 48) return;
    const int n = i / 48, k = i - 48 * n;
    out[i] = w0[n * 176 + (k < 45 ? k : k + 128)];
}
__global__ void train_w0fd_grad_kernel(const float *__restrict__ g /*[64,48]*/, float *__restrict__ gw0 /*[64,176]*/)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 48) return;
    const int n = i / 48, k = i - 48 * n;
    gw0[n * 176 + (k < 45 ? k : k + 128)] = g[i];
}

// use_nearest = 0: mix-up row [colfeat[:45] | 0] (point_aggregators.py:1257-1258)
__global__ void train_x7_noviews_kernel(const float *__restrict__ CF, const long long *__restrict__ d_s, float *__restrict__ X7)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = t / 92;
    if (s >= *d_s) return;
    const int c = (int)(t - s * 92);
    X7[t] = c < 45 ? CF[s * 128 + c] : 0.f;
}

// g_conf[0] += sum of g_conf_out over the EMPTY neighbour slots (they read point 0 through the index clamp).  Two stages, fixed order.
__global__ __launch_bounds__(256) void train_conf0_partial_kernel(const float *__restrict__ g_conf_out, const int32_t *__restrict__ pidx, long long n, float *__restrict__ part)
{
    __shared__ float s_w[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += pidx[i] < 0 ? g_conf_out[i] : 0.f;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(256) void train_conf0_final_kernel(const float *__restrict__ part, int n_part, float *__restrict__ g_conf)
{
    __shared__ float s_w[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) g_conf[0] += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// alpha branch + softplus(x - 1) + K-weighted sums transposed (point_aggregators.py:1005-1026, :471-476), 8-slot row layout: one wave per valid sample
// writes all 8 rows of gZ4 (zeros for the empty slots) -- d pre-activation of block3's last layer -- and g_wagg; alpha weights by atomics.
struct KsumPadArgs {
    const float *H4; const char *aux;                    // [rows,256]; the chain workspace's per-row scalars (pid, w_agg)
    const float *alpha_w, *alpha_b;
    const unsigned long long *counts;
    const float *gX5; int ldg5; const float *g_sigma;
    float slope;
    float *gZ4, *g_wagg, *g_alpha_w, *g_alpha_b;
    unsigned *absmax;
};
__device__ __forceinline__ float train_wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
// 16-wave workgroups (the loop is latency-bound -- 16 wave reductions in series per sample -- and every workgroup ends with 257 float atomics on the same
// addresses: many waves, few workgroups)
__global__ __launch_bounds__(1024) void train_ksum_bwd_kernel(KsumPadArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * (unsigned)blockDim.x) >> 6);
    const int n_valid = (int)a.counts[HNR_CNT_SAMPLES_VALID];
    const float4 aw = reinterpret_cast<const float4 *>(a.alpha_w)[lane];
    const float ab = a.alpha_b[0];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc_b = 0.f, gmax = 0.f;
    for (int s = wave; s < n_valid; s += n_waves) {
        const float4 gf = reinterpret_cast<const float4 *>(a.gX5 + (size_t)s * a.ldg5)[lane];
        const float gs = a.g_sigma[s];
        // the sample's 8 rows sit in one 32-row group of the chain workspace; all their loads go out together
        const size_t row0 = (size_t)8 * s;
        const char *ax = a.aux + (row0 >> 5) * CH_AUX_GROUP;
        const int j0 = (int)(row0 & 31);
        int pid[8]; float wv[8]; float4 hv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pid[k] = reinterpret_cast<const int32_t *>(ax)[j0 + k];
            wv[k] = reinterpret_cast<const float *>(ax + 128)[j0 + k];
            hv[k] = reinterpret_cast<const float4 *>(a.H4 + (row0 + k) * 256)[lane];
        }
        // The sample's sixteen wave sums (per row: h . alpha_w and h . gX5) in ONE butterfly: every step halves the number of values a lane carries (the
        // lanes of one half keep the first half of the values and hand the others over), 8 + 4 + 2 + 1 + 1 + 1 exchanges instead of 16 x 6 in series --
        // the kernel was bound by that latency.  Each sum is formed by the same pairwise additions as a plain xor butterfly: same bits.
        float r8[8], r4[4], r2[2], r1;
        {
            float v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 h = hv[k];
                v[2 * k] = h.x * aw.x + h.y * aw.y + h.z * aw.z + h.w * aw.w;
                v[2 * k + 1] = h.x * gf.x + h.y * gf.y + h.z * gf.z + h.w * gf.w;
            }
            const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) r8[i] = (b5 ? v[i + 8] : v[i]) + __shfl_xor(b5 ? v[i] : v[i + 8], 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) r4[i] = (b4 ? r8[i + 4] : r8[i]) + __shfl_xor(b4 ? r8[i] : r8[i + 4], 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) r2[i] = (b3 ? r4[i + 2] : r4[i]) + __shfl_xor(b3 ? r4[i] : r4[i + 2], 8);
            r1 = (b2 ? r2[1] : r2[0]) + __shfl_xor(b2 ? r2[0] : r2[1], 4);
            r1 += __shfl_xor(r1, 2);
            r1 += __shfl_xor(r1, 1);
        }
        // value index i = 2 k (+ 1) sits in the lanes 4 i .. 4 i + 3
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t row = row0 + k;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            float gw = 0.f;
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), 8 * k));
            const float hf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), 8 * k + 4));
            if (pid[k] >= 0) {
                const float w = wv[k];
                const float4 h = hv[k];
                const float yv = __fsub_rn(d + ab, 1.0f);
                const float ez = expf(yv);
                const float sp = yv > 20.f ? yv : log1pf(ez);
                const float spd = yv > 20.f ? 1.f : hnr_div(ez, ez + 1.f);                 // (not `/`: see hnr_div)
                const float da = w * gs * spd;
                o.x = (w * gf.x + da * aw.x) * (h.x > 0.f ? 1.f : a.slope);
                o.y = (w * gf.y + da * aw.y) * (h.y > 0.f ? 1.f : a.slope);
                o.z = (w * gf.z + da * aw.z) * (h.z > 0.f ? 1.f : a.slope);
                o.w = (w * gf.w + da * aw.w) * (h.w > 0.f ? 1.f : a.slope);
                gw = sp * gs + hf;
                acc.x += da * h.x; acc.y += da * h.y; acc.z += da * h.z; acc.w += da * h.w;
                acc_b += da;
                gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
            reinterpret_cast<float4 *>(a.gZ4 + row * 256)[lane] = o;
            if (lane == 0) a.g_wagg[row] = gw;
        }
    }
    __shared__ float4 s_w[16][64];
    __shared__ float s_b[16], s_m[16];
    const int wid = threadIdx.x >> 6;
    s_w[wid][lane] = acc;
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o));
    if (lane == 0) { s_b[wid] = acc_b; s_m[wid] = gmax; }
    __syncthreads();
    if (wid == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        float tb = 0.f, m = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const float4 p = s_w[w][lane]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; tb += s_b[w]; m = fmaxf(m, s_m[w]); }
        atomicAdd(a.g_alpha_w + 4 * lane, t.x); atomicAdd(a.g_alpha_w + 4 * lane + 1, t.y);
        atomicAdd(a.g_alpha_w + 4 * lane + 2, t.z); atomicAdd(a.g_alpha_w + 4 * lane + 3, t.w);
        if (lane == 0) { atomicAdd(a.g_alpha_b, tb); if (m > 0.f) atomicMax(a.absmax, __float_as_uint(m)); }
    }
}

// gX3[row, 256 + e] = sum_n dZ3[row, n] W30[n, 256 + e], e < 7 (column 263: 0): the gradient of block3's 7 extra inputs (point colour, direction
// terms; point_aggregators.py:957-971).  16 lanes per row, fp32.  A lane multiplies the same sixteen rows n of W30 for every row of dZ3: their 112 weights
// stay in registers (from LDS, [256][8] floats read 32 B per lane, the sixteen lanes of a row hit the same banks eight at a time: 0.16 ms for a 0.04-ms stream).
__global__ __launch_bounds__(256) void train_extras_dgrad_kernel(const float *__restrict__ dZ3, const float *__restrict__ w30 /*[256,263]*/, const long long *__restrict__ d_m,
                                                                 float *__restrict__ gX3 /*[rows,264]*/)
{
    const long long M = *d_m;
    const int sub = threadIdx.x & 15;
    float w[16][7];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 7; ++e) w[4 * it + q][e] = w30[(4 * (sub + 16 * it) + q) * 263 + 256 + e];
    for (long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; row < M; row += ((long long)gridDim.x * blockDim.x) >> 4) {
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float4 z[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) z[it] = *reinterpret_cast<const float4 *>(dZ3 + (size_t)row * 256 + 4 * (sub + 16 * it));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float zz[4] = {z[it].x, z[it].y, z[it].z, z[it].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 7; ++e) acc[e] = fmaf(zz[q], w[4 * it + q][e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 7; ++e) {
            float v = acc[e];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            acc[e] = v;
        }
        if (sub == 0) {
            float *o = gX3 + (size_t)row * 264 + 256;
            *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(acc[4], acc[5], acc[6], 0.f);
        }
    }
}

// the small accumulated-into gradient buffers of one step, zeroed by one launch (a memset per buffer is a launch per buffer)
struct ZeroJobs { float *p[24]; int n[24]; };
__global__ void train_zero_kernel(ZeroJobs z)
{
    float *p = z.p[blockIdx.x];
    if (!p) return;
    for (int i = threadIdx.x; i < z.n[blockIdx.x]; i += blockDim.x) p[i] = 0.f;
}

__global__ void train_bbox_init_kernel(int32_t *bbox, int V, int H, int W)
{
    const int v = threadIdx.x;
    if (v < V) { bbox[4 * v] = W; bbox[4 * v + 1] = H; bbox[4 * v + 2] = -1; bbox[4 * v + 3] = -1; }
}

int check_params(const hnr_train_params *p, const char *who)
{
    if (!p) { set_error("%s: NULL params", who); return HNR_ERR_BADARG; }
    if (p->K != 8) { set_error("%s: built for K = 8 (got %d)", who, p->K); return HNR_ERR_BADARG; }
    if (p->R <= 0 || p->SR <= 0 || p->cap_samples <= 0 || p->V < 0 || p->V > 8 || p->n_points <= 0 || (p->V > 0 && (p->H <= 1 || p->W <= 1)) ||
        !(p->slope > 0.f && p->slope < 1.f) || (int64_t)p->cap_samples * 8 + 512 >= INT_MAX / 2) {
        set_error("%s: bad sizes (R=%d SR=%d cap_samples=%d V=%d n_points=%d H=%d W=%d slope=%g)", who, p->R, p->SR, p->cap_samples, p->V, p->n_points, p->H, p->W, (double)p->slope);
        return HNR_ERR_BADARG;
    }
    return HNR_OK;
}

#define TR(call) do { int rc_ = (call); if (rc_ != HNR_OK) return rc_; } while (0)

// Side streams of the library, per device (created at the first training call on that device).  HNR_TRAIN_SIDE is a bit mask, default 63 (all of them;
// 15 until round 4 found what had made three busy queues differ run to run -- packed fp32 instructions, DESIGN.md section 2 -- and removed it):
//   bit 0 (1): the reference-view CNN on `stream` -- its forward beside the query and the per-neighbour chain, its backward (pixel scatter, upsample,
//              conv pyramid) beside the backward stages 7 - 11: strings of small latency-bound kernels whose results are needed late / not at all downstream;
//   bit 1 (2): the backward call's buffer clears on `stream`;
//   bit 2 (4): ALL fifteen weight-gradient GEMMs (hnr_h2wgrad) on `stream`, each behind an event recorded after the kernel that wrote its dZ (the stream is
//              in order, so they share one partial-sum scratch) -- shipped: 4.9 -> 4.65 ms per step, 6 000 + 16 000 repeated steps bit-identical;
//   bit 3 (8): the step's weight-image packs (forward images, the backward's transposed images, the per-point table image) on `stream_w`, the pack stream;
//   bit 4 (16): the image branch's backward on the pack stream instead -- three busy queues (in round 3 such an arrangement made ~1 step in 10 differ:
//              train_ksum_bwd_kernel, lanes 48..63 -- the packed-fp32 failure);
//   bit 5 (32): the weight gradients of the per-neighbour layers (the four 256-wide GEMMs over all row slots + block1.0's two) alternate between
//              `stream` and the pack stream, each queue with its own partial-sum scratch: the last of them end the step (they trail the input-gradient
//              chain by one kernel each), two queues shorten that tail.
// Forked from / joined to the caller's stream with events inside each call.  HNR_TRAIN_SIDE=0: everything in line on the caller's stream.
struct TrainSide {
    hipStream_t stream = nullptr, stream_w = nullptr;                     // image branch / clears; weight packs
    hipEvent_t fork_f = nullptr, join_f = nullptr, fork_b = nullptr, join_b = nullptr, fork_z = nullptr, join_z = nullptr, fork_g = nullptr, ev_w[3] = {};
    int on = -1;
};
// An error return between a fork and its join must not leave side-stream work running on a workspace the caller may free: the guard drains the
// side streams unless the call reached its joins.
struct TrainSideGuard {
    TrainSide *t = nullptr; bool armed = false;
    ~TrainSideGuard() { if (armed && t) { if (t->stream) (void)hipStreamSynchronize(t->stream); if (t->stream_w) (void)hipStreamSynchronize(t->stream_w); } }
};
TrainSide &train_side()
{
    static TrainSide per_dev[64];                                          // one process per GPU is the deployment; a second device in the process gets its own streams
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    TrainSide &t = per_dev[dev];
    if (t.on < 0) {
        const char *e = getenv("HNR_TRAIN_SIDE");
        t.on = e ? atoi(e) : 63;                                         // see the bit list above
        if (t.on && (hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&t.fork_f, hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&t.join_f, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t.fork_b, hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&t.join_b, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t.fork_z, hipEventDisableTiming) != hipSuccess ||
                     hipEventCreateWithFlags(&t.join_z, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t.fork_g, hipEventDisableTiming) != hipSuccess ||
                     hipStreamCreateWithFlags(&t.stream_w, hipStreamNonBlocking) != hipSuccess)) t.on = 0;
        for (int i = 0; i < 3 && t.on; ++i) if (hipEventCreateWithFlags(&t.ev_w[i], hipEventDisableTiming) != hipSuccess) t.on = 0;
    }
    return t;
}

// Images of the transposed weights for the backward call's input-gradient GEMMs (hnr_h2lin); L.W0fd (the merge-weight MLP's first layer without
// its colour-feature columns) must have been written on `stream` before.
static int pack_transposed_images(const Layout &L, const hnr_train_weights *w, int V, void *stream)
{
    const float *W[IM_N - 1] = {w->block1_0_w, w->block3_2_w, w->block3_0_w, w->block1_2_w, w->cf_w[2], w->cf_w[1], w->cf_w[0], w->mw_w[2], w->mw_w[1], L.W0fd, w->mw_w[0] + 45,
                                w->mx_w[2], w->mx_w[1], w->mx_w[0]};
    //                         IM_TABT [224 <- 256]  B32T  B30T [256 <- 256: the H2 columns]  B12T  CF2T  CF1T  CF0T [256 <- 128]  MW2T  MW1T  MW0FDT [48 <- 64]  MW0CFT [128 <- 64]  MX2T  MX1T  MX0T [90 <- 45]
    const int64_t rs[IM_N - 1] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    const int64_t cs[IM_N - 1] = {284, 256, 263, 256, 128, 128, 280, 64, 64, 48, 176, 45, 45, 90};
    const int Nn[IM_N - 1] = {224, 256, 256, 256, 128, 128, 256, 64, 64, 48, 128, 45, 45, 90};
    const int Kk[IM_N - 1] = {256, 256, 256, 256, 128, 128, 128, 64, 64, 64, 64, 45, 45, 45};
    void *out[IM_N - 1];
    for (int i = 1; i < IM_N; ++i) out[i - 1] = L.img[i];
    if (V > 0) TR(hnr_h2lin_pack(IM_N - 1, W, rs, cs, Nn, Kk, nullptr, out, stream));
    else {
        // image branch off: no merge-weight layers
        const float *W2[10] = {W[0], W[1], W[2], W[3], W[4], W[5], W[6], W[11], W[12], W[13]};
        const int64_t rs2[10] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, cs2[10] = {cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], cs[6], cs[11], cs[12], cs[13]};
        const int N2[10] = {Nn[0], Nn[1], Nn[2], Nn[3], Nn[4], Nn[5], Nn[6], Nn[11], Nn[12], Nn[13]}, K2[10] = {Kk[0], Kk[1], Kk[2], Kk[3], Kk[4], Kk[5], Kk[6], Kk[11], Kk[12], Kk[13]};
        void *out2[10] = {out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[11], out[12], out[13]};
        TR(hnr_h2lin_pack(10, W2, rs2, cs2, N2, K2, nullptr, out2, stream));
    }
    return HNR_OK;
}

}  // namespace

// Debug / tools: where the forward pass keeps its intermediate tensors inside the caller's workspace (tools/bisect_train_forward.py compares them between
// repeated steps).  names: "Xd H1 X3 H3 H4 E Tu X5 sigma T1 T2 CF X6 vmask M1 M2 M3 X7 Y1 Y2 Y3 fm row_pid vs_item fm_scratch" in this order; out[2 i] = byte offset,
// out[2 i + 1] = bytes.  Returns the number of entries.
extern "C" int hnr_render_train_debug_layout(const hnr_train_params *p, int64_t *out, int max_entries)
{
    if (!p || !out || check_params(p, "hnr_render_train_debug_layout")) return -1;
    bool ok = true;
    char *base = reinterpret_cast<char *>(256);                              // a non-null base: only differences are used
    const Layout L = carve(base, ~(size_t)0 >> 1, p, &ok);
    const size_t rows = L.rows_cap, cap = (size_t)p->cap_samples, VS = L.VS > 0 ? L.VS : 1, uc = L.ucap;
    const struct { const void *ptr; size_t bytes; } e[] = {
        {L.Xd, rows * 64 * 4}, {L.H1, rows * 256 * 4}, {L.X3, rows * 264 * 4}, {L.H3, rows * 256 * 4}, {L.H4, rows * 256 * 4}, {L.E, uc * 224 * 4}, {L.Tu, uc * 256 * 4},
        {L.X5, cap * 280 * 4}, {L.sigma, cap * 4}, {L.T1, cap * 128 * 4}, {L.T2, cap * 128 * 4}, {L.CF, cap * 128 * 4}, {L.X6, VS * 48 * 4}, {L.vmask, VS * 4},
        {L.M1, VS * 64 * 4}, {L.M2, VS * 64 * 4}, {L.M3, VS * 64 * 4}, {L.X7, cap * 92 * 4}, {L.Y1, cap * 48 * 4}, {L.Y2, cap * 48 * 4}, {L.Y3, cap * 48 * 4},
        {L.fm, (p->V > 0 ? (size_t)p->V * p->H * p->W * 48 : 4) * 4}, {L.row_pid, rows * 4}, {L.vs_item, cap * 4}, {L.fm_scratch, L.fm_elems * 4}};
    const int n = (int)(sizeof(e) / sizeof(e[0]));
    for (int i = 0; i < n && i < max_entries; ++i) { out[2 * i] = (int64_t)(reinterpret_cast<const char *>(e[i].ptr) - base); out[2 * i + 1] = (int64_t)e[i].bytes; }
    return n < max_entries ? n : max_entries;
}

// The batch's touched points (ascending point ids) and their number, as the forward call left them in the workspace: what a sparse optimiser step or
// a sparse gradient exchange between ranks needs (parallel.PointGradExchange) -- without a torch.unique over sample_pidx (a sort + a host read).
extern "C" int hnr_render_train_touched(const hnr_train_params *p, void *d_workspace, int64_t workspace_bytes, const int32_t **d_ids, const int64_t **d_count,
                                        int64_t *capacity)
{
    TR(check_params(p, "hnr_render_train_touched"));
    if (!d_workspace || ((uintptr_t)d_workspace & 255) || !d_ids || !d_count || !capacity) { set_error("hnr_render_train_touched: NULL / unaligned argument"); return HNR_ERR_BADARG; }
    bool ok = true;
    const Layout L = carve(d_workspace, (size_t)workspace_bytes, p, &ok);
    if (!ok) { set_error("hnr_render_train_touched: workspace too small"); return HNR_ERR_BADARG; }
    *d_ids = L.ulist; *d_count = reinterpret_cast<const int64_t *>(L.tc + TC_U); *capacity = (int64_t)L.ucap;
    return HNR_OK;
}

extern "C" int64_t hnr_render_train_workspace_bytes(const hnr_train_params *p)
{
    if (check_params(p, "hnr_render_train_workspace_bytes") != HNR_OK) return -1;
    return (int64_t)carve(nullptr, 0, p, nullptr).bytes + 256;
}

extern "C" int hnr_render_train_forward(const hnr_grid *grid, const hnr_train_params *p, const hnr_train_cloud *cl, const hnr_train_weights *w,
                                        const hnr_render_camera *cam, const hnr_train_views *vw, const uint8_t *d_drop_lut, const uint8_t *d_ray_drop,
                                        void *d_workspace, int64_t workspace_bytes, const hnr_render_outputs *o, void *stream)
{
    TR(check_params(p, "hnr_render_train_forward"));
    if (!grid || !cl || !w || !cam || !o || (p->V > 0 && !vw)) { set_error("hnr_render_train_forward: NULL argument block"); return HNR_ERR_BADARG; }
    if (!d_workspace || ((uintptr_t)d_workspace & 255)) { set_error("hnr_render_train_forward: workspace must be 256-byte aligned"); return HNR_ERR_BADARG; }
    if (!o->d_raycolor || !o->d_opacity || !o->d_is_background || !o->d_blend_weight || !o->d_ray_mask || !o->d_decoded || !o->d_sample_pidx || !o->d_sample_loc_w ||
        !o->d_ray_nsamp || !o->d_counts || !o->d_status || !o->d_weight || !o->d_conf_coefficient) { set_error("hnr_render_train_forward: NULL output pointer"); return HNR_ERR_BADARG; }
    bool ok = true;
    const Layout L = carve(d_workspace, (size_t)workspace_bytes, p, &ok);
    if (!ok) { set_error("hnr_render_train_forward: workspace too small (%lld bytes, need %lld)", (long long)workspace_bytes, (long long)hnr_render_train_workspace_bytes(p)); return HNR_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    const int R = p->R, SR = p->SR, K = 8, cap = p->cap_samples, V = p->V;
    const float sl = p->slope;
    const int64_t *dS = reinterpret_cast<const int64_t *>(L.tc + TC_S), *dU = reinterpret_cast<const int64_t *>(L.tc + TC_U);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(o->d_counts);
    int stage = 0;
    auto mark = [&]() -> int {      // optional HIP events at the stage boundaries (profiling hook, as in hnr_render_forward)
        if (o->stage_events && o->stage_events[stage]) { if (hipEventRecord((hipEvent_t)o->stage_events[stage], st) != hipSuccess) { set_error("hipEventRecord failed"); return HNR_ERR_HIP; } }
        ++stage;
        return HNR_OK;
    };
    TR(mark());
    // the reference-view feature pyramid needs the images and the conv weights only and is first read by the merge stage: side stream, from here
    TrainSide &side = train_side();
    TrainSideGuard guard; guard.t = &side; guard.armed = side.on != 0;
    bool fwd_forked = false;
    if (V > 0 && (side.on & 1)) {
        HNR_HIP_CHECK(hipEventRecord(side.fork_f, st));
        HNR_HIP_CHECK(hipStreamWaitEvent(side.stream, side.fork_f, 0));
        TR(hnr_image_features(vw->d_images, V, p->H, p->W, w->conv_w, w->conv_b, sl, L.fm_scratch, L.fm, (void *)side.stream));
        HNR_HIP_CHECK(hipEventRecord(side.join_f, side.stream));
        fwd_forked = true;
    }

    // ---- kernel images of this step's weights: first used by the chain kernel -- packed on the second side stream meanwhile (the per-point table's
    //      image, needed at once, on the caller's)
    void *sp = stream;
    if (side.on & 8) {
        HNR_HIP_CHECK(hipEventRecord(side.ev_w[0], st));
        HNR_HIP_CHECK(hipStreamWaitEvent(side.stream_w, side.ev_w[0], 0));
        sp = (void *)side.stream_w;
    }
    {
        // per-point table layer: [emb | PE(emb)] W0[:, :224]^T (no bias: block1.0's bias is added per row by the chain kernel); first: it is the first image used
        const float *W[1] = {w->block1_0_w}; const int64_t rs[1] = {284}, cs[1] = {1}; const int N1[1] = {256}, K1[1] = {224}; void *out[1] = {L.img[IM_TAB]};
        TR(hnr_h2lin_pack(1, W, rs, cs, N1, K1, nullptr, out, sp));
    }
    TR(hnr_chain_pack(w->block1_0_w + 224, 284, w->block1_0_b, w->block1_2_w, w->block1_2_b, w->block3_0_w, w->block3_0_b, w->block3_2_w, w->block3_2_b,
                      w->alpha_w, w->alpha_b, L.img_chain, sp));
    const int cfN[4] = {128, 128, 128, 64}, cfK[4] = {280, 128, 128, 128}, cfld[4] = {280, 128, 128, 176};
    {
        const float *W[4] = {w->cf_w[0], w->cf_w[1], w->cf_w[2], V > 0 ? w->mw_w[0] + 45 : nullptr}, *B[4] = {w->cf_b[0], w->cf_b[1], w->cf_b[2], V > 0 ? w->mw_b[0] : nullptr};
        TR(hnr_mlp3_pack(V > 0 ? 4 : 3, W, cfld, cfN, cfK, B, L.img_cf, sp));
    }
    const int mwN[3] = {64, 64, 64}, mwK[3] = {48, 64, 64}, mwld[3] = {48, 64, 64};
    if (V > 0) {
        train_w0fd_kernel<<<(64 * 48 + 255) / 256, 256, 0, (hipStream_t)sp>>>(w->mw_w[0], L.W0fd);
        const float *W[3] = {L.W0fd, w->mw_w[1], w->mw_w[2]}, *B[3] = {nullptr, w->mw_b[1], w->mw_b[2]};
        TR(hnr_mlp3_pack(3, W, mwld, mwN, mwK, B, L.img_mw, sp));
    }
    const int mxN[3] = {45, 45, 45}, mxK[3] = {90, 45, 45}, mxld[3] = {90, 45, 45};
    {
        const float *W[3] = {w->mx_w[0], w->mx_w[1], w->mx_w[2]}, *B[3] = {w->mx_b[0], w->mx_b[1], w->mx_b[2]};
        TR(hnr_mlp3_pack(3, W, mxld, mxN, mxK, B, L.img_mx, sp));
    }
    if (side.on & 8) HNR_HIP_CHECK(hipEventRecord(side.ev_w[1], side.stream_w));
    // ---- images of the TRANSPOSED weights (the backward call's input-gradient GEMMs): the same weights, so they are packed here, behind the
    //      forward's own images on the pack stream, instead of at the head of the backward call on the caller's stream (0.07 ms there)
    TR(pack_transposed_images(L, w, V, sp));
    if (side.on & 8) HNR_HIP_CHECK(hipEventRecord(side.ev_w[2], side.stream_w));
    TR(mark());
    // ---- query (jittered depths: cam->d_tmid with tmid_stride = D), padded outputs
    hnr_query_params q;
    q.R = R; q.D = p->D; q.SR = SR; q.K = K; q.radius2 = p->radius2; q.tmid_stride = p->tmid_stride; q.pad_outputs = 1; q.knn_order = p->knn_order;
    for (int i = 0; i < 3; ++i) q.kernel_size[i] = p->kernel_size[i];
    TR(hnr_march_query(grid, cam->d_campos, cam->d_raydir, cam->d_tmid, &q, o->d_sample_pidx, o->d_sample_loc_w, o->d_ray_nsamp, o->d_ray_mask, L.work, o->d_counts, stream));
    TR(hnr_chain_plan(L.work, o->d_sample_pidx, o->d_counts, K, R * SR, 0, L.vs_item, cap, L.scratch, stream));
    train_counts_kernel<<<1, 1, 0, st>>>(cnt, cap, o->d_status, L.tc, L.amax);
    train_vs_fill_kernel<<<cdiv(cap, 256), 256, 0, st>>>(L.vs_item, o->d_sample_pidx, cnt, L.vs_off, L.vs_cnt);
    train_ray_drop_kernel<<<1, 1024, 0, st>>>(o->d_ray_mask, R, d_drop_lut, d_ray_drop, L.ray_drop);
    HNR_LAUNCH_CHECK();
    HNR_HIP_CHECK(hipMemsetAsync(o->d_decoded, 0, (size_t)R * SR * 4 * sizeof(float), st));
    HNR_HIP_CHECK(hipMemsetAsync(o->d_weight, 0, (size_t)R * SR * K * sizeof(float), st));
    train_conf_fill_kernel<<<256, 256, 0, st>>>(cl->d_conf, o->d_conf_coefficient, (long long)R * SR * K);
    TR(mark());
    // ---- reference-view feature pyramid (activations kept for the conv backward): on the side stream, forked at the start of the call
    if (V > 0 && !fwd_forked) TR(hnr_image_features(vw->d_images, V, p->H, p->W, w->conv_w, w->conv_b, sl, L.fm_scratch, L.fm, stream));
    TR(mark());
    // ---- per-neighbour chain
    TR(chain_gather_train(cl->d_xyz, cl->d_conf, cl->d_dir, cl->d_color, o->d_sample_pidx, o->d_sample_loc_w, cam->d_raydir, cam->d_campos, cam->d_camrot, L.vs_item,
                          o->d_counts, SR, K, cap, L.chain_ws, L.X5, 280, o->d_weight, o->d_conf_coefficient, L.Xd, L.row_pid, stream));
    TR(unique_points_dc(L.row_pid, (int64_t)L.rows_cap, L.tc + TC_M8, p->n_points, L.uidx, L.ulist, (int)L.ucap, L.row_u, L.ucount, L.uscratch, L.seg_start, L.seg_cnt,
                        L.row_list, st));
    train_ucount_kernel<<<1, 1, 0, st>>>(L.ucount, (long long)L.ucap, L.tc);
    TR(point_rows_dc(cl->d_emb, L.ulist, (int)L.ucap, L.tc + TC_U, L.E, 224, st));
    if (side.on & 8) HNR_HIP_CHECK(hipStreamWaitEvent(st, side.ev_w[1], 0));    // the forward's weight images are packed
    TR(hnr_h2lin(L.E, 224, (int64_t)L.ucap, dU, 1, 0, L.img[IM_TAB], 256, 224, 0, 0, sl, nullptr, 0, L.Tu, 256, nullptr, stream));
    TR(mark());
    {
        float *H[4] = {L.H1, L.X3, L.H3, L.H4}; const int ldh[4] = {256, 264, 256, 256};
        TR(chain_forward_train(L.chain_ws, L.Tu, 256, L.uidx, L.img_chain, o->d_counts, cap, sl, L.X5, 280, L.sigma, H, ldh, L.amax + AM_H1, L.amax + AM_X5, stream, L.row_u, (int)L.ucap, L.hbits, (long long)L.rows_cap * 8));     // (AM_X5 starts at 1: the direction encoding's columns)
    }
    TR(mark());
    // maxima that only the backward call's weight gradients read (per-tensor scales of X6, X7): on the side stream once the image branch is done with it
    bool fwd_side2 = false;
    auto absmax_bwd = [&](const float *A, int lda, int nseg, int64_t segs, int Nn, int slot) -> int {
        if (!(side.on & 1)) return hnr_absmax(A, lda, cap, dS, nseg, segs, Nn, L.amax + slot, stream);
        HNR_HIP_CHECK(hipEventRecord(side.fork_g, st));
        HNR_HIP_CHECK(hipStreamWaitEvent(side.stream, side.fork_g, 0));
        fwd_side2 = true;
        return hnr_absmax(A, lda, cap, dS, nseg, segs, Nn, L.amax + slot, (void *)side.stream);
    };
    // ---- per-sample MLPs
    const int act1110[4] = {1, 1, 1, 0}, act111[3] = {1, 1, 1}, act110[3] = {1, 1, 0};
    TR(mlp3_forward_train(L.X5, 280, cap, o->d_counts, HNR_CNT_SAMPLES_VALID, 1, 0, L.img_cf, V > 0 ? 4 : 3, cfN, cfK, act1110, sl, nullptr, nullptr, 0, L.CF, 128,
                          V > 0 ? L.pre : nullptr, 64, L.T1, 128, L.T2, 128, L.amax + AM_T1, stream));
    if (V > 0) {
        if (fwd_forked) HNR_HIP_CHECK(hipStreamWaitEvent(st, side.join_f, 0));      // the feature map is ready
        TR(hnr_proj_rows(o->d_sample_loc_w, L.vs_item, o->d_counts, vw->d_w2c, vw->d_intrinsic, cam->d_campos, vw->d_campos_nearest, L.fm, V, p->H, p->W, L.CF, 128, cap,
                         L.X6, 48, L.vmask, L.row_s, stream));
        TR(absmax_bwd(L.X6, 48, V, cap, 48, AM_X6));
        TR(mlp3_forward_train(L.X6, 48, (int64_t)V * cap, o->d_counts, HNR_CNT_SAMPLES_VALID, V, cap, L.img_mw, 3, mwN, mwK, act111, sl, L.pre, L.row_s, 64, L.M3, 64,
                              nullptr, 0, L.M1, 64, L.M2, 64, L.amax + AM_M1, stream));
        TR(hnr_merge(L.X6, 48, L.M3, 64, w->mw_w[3], w->mw_b[3], L.vmask, vw->d_frame_w, L.CF, 128, o->d_counts, V, cap, L.X7, 92, L.ray_drop, L.vs_item, SR, stream));
    } else {
        train_x7_noviews_kernel<<<cdiv((int64_t)cap * 92, 256), 256, 0, st>>>(L.CF, L.tc + TC_S, L.X7);
    }
    TR(absmax_bwd(L.X7, 92, 1, 0, 90, AM_X7));
    TR(mlp3_forward_train(L.X7, 92, cap, o->d_counts, HNR_CNT_SAMPLES_VALID, 1, 0, L.img_mx, 3, mxN, mxK, act110, sl, nullptr, nullptr, 0, L.Y3, 48, nullptr, 0,
                          L.Y1, 48, L.Y2, 48, L.amax + AM_Y1, stream));
    TR(mark());
    TR(hnr_final_color(L.Y3, 48, L.CF, 128, w->fin_w, w->fin_b, L.sigma, L.vs_item, o->d_counts, cap, o->d_decoded, stream));
    TR(hnr_composite(o->d_decoded, o->d_sample_loc_w, o->d_sample_pidx, o->d_ray_mask, nullptr, cam->d_campos, cam->d_camrot, cam->d_bg_color, R, SR, K, p->vsize_z,
                     p->raydist_mode_unit, o->d_raycolor, o->d_opacity, o->d_is_background, o->d_blend_weight, stream));
    TR(mark());
    if (side.on & 8) HNR_HIP_CHECK(hipStreamWaitEvent(st, side.ev_w[2], 0));    // ... and the backward's (long done)
    if (fwd_side2) { HNR_HIP_CHECK(hipEventRecord(side.join_f, side.stream)); HNR_HIP_CHECK(hipStreamWaitEvent(st, side.join_f, 0)); }
    HNR_LAUNCH_CHECK();
    guard.armed = false;                                             // every fork of this call has been joined
    return HNR_OK;
}

extern "C" int hnr_render_train_backward(const hnr_train_params *p, const hnr_train_cloud *cl, const hnr_train_weights *w, const hnr_render_camera *cam,
                                         const hnr_train_views *vw, void *d_workspace, int64_t workspace_bytes, const hnr_render_outputs *o,
                                         const float *d_g_raycolor, const float *d_g_conf_coefficient, const hnr_train_cloud_grads *gc,
                                         const hnr_train_weights *gw_, void *stream)
{
    TR(check_params(p, "hnr_render_train_backward"));
    if (!cl || !w || !cam || !o || !gc || !gw_ || !d_g_raycolor || (p->V > 0 && !vw)) { set_error("hnr_render_train_backward: NULL argument"); return HNR_ERR_BADARG; }
    if (!d_workspace || ((uintptr_t)d_workspace & 255)) { set_error("hnr_render_train_backward: workspace must be 256-byte aligned"); return HNR_ERR_BADARG; }
    bool ok = true;
    const Layout L = carve(d_workspace, (size_t)workspace_bytes, p, &ok);
    if (!ok) { set_error("hnr_render_train_backward: workspace too small"); return HNR_ERR_BADARG; }
    // the gradient block has the parameter block's layout; its pointers are written through
    struct G { float *block1_0_w, *block1_0_b, *block1_2_w, *block1_2_b, *block3_0_w, *block3_0_b, *block3_2_w, *block3_2_b, *alpha_w, *alpha_b;
               float *cf_w[3], *cf_b[3], *mw_w[4], *mw_b[4], *mx_w[3], *mx_b[3], *fin_w, *fin_b, *conv_w[6], *conv_b[6]; };
    static_assert(sizeof(G) == sizeof(hnr_train_weights), "gradient block layout");
    const G &g = *reinterpret_cast<const G *>(gw_);
    hipStream_t st = (hipStream_t)stream;
    const int R = p->R, SR = p->SR, K = 8, cap = p->cap_samples, V = p->V, N = p->n_points;
    const float sl = p->slope;
    const int64_t *dS = reinterpret_cast<const int64_t *>(L.tc + TC_S), *dM = reinterpret_cast<const int64_t *>(L.tc + TC_M8), *dU = reinterpret_cast<const int64_t *>(L.tc + TC_U);
    const int64_t rows = (int64_t)L.rows_cap, ucap = (int64_t)L.ucap;
    uint32_t *am = L.amax;
    int stage = 0;
    auto mark = [&]() -> int {
        if (o->stage_events && o->stage_events[stage]) { if (hipEventRecord((hipEvent_t)o->stage_events[stage], st) != hipSuccess) { set_error("hipEventRecord failed"); return HNR_ERR_HIP; } }
        ++stage;
        return HNR_OK;
    };
    TR(mark());
    TrainSide &side = train_side();
    TrainSideGuard guard; guard.t = &side; guard.armed = side.on != 0;
    const int side_on = side.on & 1, side_z = side.on & 2;
    hipStream_t side_stream = side.stream;
    hipEvent_t side_fork = side.fork_b, side_join = side.join_b;
    bool forked = false, forked6w = false;

    // ---- zero what is accumulated into.  The dense point-gradient buffers (312 MB at 2 M points) are first written by the call's last kernels:
    //      cleared on the side stream, waited for before stage 10
    {
        hipStream_t sz = st;
        if (side_z) { HNR_HIP_CHECK(hipEventRecord(side.fork_z, st)); HNR_HIP_CHECK(hipStreamWaitEvent(side_stream, side.fork_z, 0)); sz = side_stream; }
        HNR_HIP_CHECK(hipMemsetAsync(gc->d_emb, 0, (size_t)N * 32 * 4, sz));
        HNR_HIP_CHECK(hipMemsetAsync(gc->d_conf, 0, (size_t)N * 4, sz));
        HNR_HIP_CHECK(hipMemsetAsync(gc->d_dir, 0, (size_t)N * 12, sz));
        HNR_HIP_CHECK(hipMemsetAsync(gc->d_color, 0, (size_t)N * 12, sz));
        if (side_z) HNR_HIP_CHECK(hipEventRecord(side.join_z, side_stream));
    }
    {
        ZeroJobs z;
        int nz = 0;
        auto add = [&](float *ptr, int n) { z.p[nz] = ptr; z.n[nz] = n; ++nz; };
        add(g.fin_w, 3 * 128); add(g.fin_b, 3); add(g.alpha_w, 256); add(g.alpha_b, 1);
        if (V > 0) {
            add(g.mw_w[3], 64); add(g.mw_b[3], 1);
            const int cw[6] = {6 * 3 * 9, 6 * 6 * 9, 12 * 6 * 9, 12 * 12 * 9, 24 * 12 * 9, 24 * 24 * 9}, cb[6] = {6, 6, 12, 12, 24, 24};
            for (int i = 0; i < 6; ++i) { add(g.conv_w[i], cw[i]); add(g.conv_b[i], cb[i]); }
        }
        for (int i = nz; i < 24; ++i) { z.p[i] = nullptr; z.n[i] = 0; }
        train_zero_kernel<<<nz, 256, 0, st>>>(z);
    }
    // (the images of the transposed weights were packed by the forward call: pack_transposed_images)
    // weight gradient of one layer: dW = dZ^T X, db = column sums of dZ
    auto wgrad = [&](const float *dZ, int ldz, const float *X, int ldx, int64_t Mcap, const int64_t *dm, int nseg, int64_t segs, int Nn, int Kk, int amz, int amx,
                     float *dW, int lddw, float *db) -> int {
        return hnr_h2wgrad(dZ, ldz, X, ldx, Mcap, dm, nseg, segs, Nn, Kk, am + amz, am + amx, dW, lddw, db, 0, L.wg_scratch, stream);
    };
    // The weight gradients (fifteen GEMMs + reductions, 1.5 ms of kernels) are results nothing in this call reads: they go to the side stream (the one the
    // image branch's backward runs on: two busy queues), each behind an event recorded after the kernel that wrote its dZ; the stream runs them in order,
    // so they share one partial-sum scratch
    const bool side_g = (side.on & 4) != 0;
    int n_big = 0;
    auto wgrad_n = [&](const float *dZ, int ldz, const float *X, int ldx, int64_t Mcap, const int64_t *dm, int nseg, int64_t segs, int Nn, int Kk, int amz, int amx,
                       float *dW, int lddw, float *db) -> int {
        if (!side_g) return wgrad(dZ, ldz, X, ldx, Mcap, dm, nseg, segs, Nn, Kk, amz, amx, dW, lddw, db);
        HNR_HIP_CHECK(hipEventRecord(side.fork_g, st));
        const bool alt = (side.on & 32) && Mcap == rows && ((n_big++) & 1);     // every second per-neighbour weight gradient: the other queue + its own scratch
        hipStream_t sg = alt ? side.stream_w : side_stream;
        HNR_HIP_CHECK(hipStreamWaitEvent(sg, side.fork_g, 0));
        forked = true;
        if (alt) forked6w = true;                                               // (joined below like the image branch of bit 4)
        return hnr_h2wgrad(dZ, ldz, X, ldx, Mcap, dm, nseg, segs, Nn, Kk, am + amz, am + amx, dW, lddw, db, 0, alt ? L.wg_scratch2 : L.wg_scratch, (void *)sg);
    };
    // input gradient through a LeakyReLU: out = (dZ W) * LeakyReLU'(side); side == NULL: out = dZ W
    auto dgrad = [&](const float *dZ, int ldz, int64_t Mcap, const int64_t *dm, int nseg, int64_t segs, int im, int Nn, int Kk, const float *side, int lds_, float *out, int ldo,
                     int amo) -> int {
        return hnr_h2lin(dZ, ldz, Mcap, dm, nseg, segs, L.img[im], Nn, Kk, side ? 1 : 0, 0, sl, side, lds_, out, ldo, amo >= 0 ? am + amo : nullptr, stream);
    };
    TR(mark());
    // ---- 1. composite, 2. final colour
    TR(hnr_composite_bwd(o->d_decoded, o->d_sample_loc_w, o->d_sample_pidx, o->d_ray_mask, nullptr, cam->d_campos, cam->d_camrot, cam->d_bg_color, R, SR, K, p->vsize_z,
                         p->raydist_mode_unit, d_g_raycolor, L.g_dec, stream));
    TR(final_color_bwd_max(L.Y3, 48, L.CF, 128, w->fin_w, w->fin_b, L.vs_item, o->d_counts, cap, L.g_dec, L.gY3, 48, L.gCF, 128, L.g_sigma, g.fin_w, g.fin_b, am + AM_gY3,
                           stream));
    // ---- 3. mix-up block (its last layer has no activation: gY3 is the gradient of its pre-activation)
    TR(wgrad_n(L.gY3, 48, L.Y2, 48, cap, dS, 1, 0, 45, 45, AM_gY3, AM_Y2, g.mx_w[2], 45, g.mx_b[2]));
    TR(dgrad(L.gY3, 48, cap, dS, 1, 0, IM_MX2T, 45, 45, L.Y2, 48, L.dY2, 48, AM_dY2));
    TR(wgrad_n(L.dY2, 48, L.Y1, 48, cap, dS, 1, 0, 45, 45, AM_dY2, AM_Y1, g.mx_w[1], 45, g.mx_b[1]));
    TR(dgrad(L.dY2, 48, cap, dS, 1, 0, IM_MX1T, 45, 45, L.Y1, 48, L.dY1, 48, AM_dY1));
    TR(wgrad_n(L.dY1, 48, L.X7, 92, cap, dS, 1, 0, 45, 90, AM_dY1, AM_X7, g.mx_w[0], 90, g.mx_b[0]));
    TR(dgrad(L.dY1, 48, cap, dS, 1, 0, IM_MX0T, 90, 45, nullptr, 0, L.gX7, 92, -1));
    TR(mark());
    if (V > 0) {
        // ---- 4. merge; 5. merge-weight MLP (first layer split: [imgfeat45 | ddir3] per (view, sample) row, colour feature once per sample)
        TR(merge_bwd_max(L.X6, 48, L.M3, 64, w->mw_w[3], w->mw_b[3], L.vmask, vw->d_frame_w, o->d_counts, V, cap, sl, L.ray_drop, L.vs_item, SR, L.gX7, 92, L.gF, 48, L.gZ3m, 64,
                         L.gCF, 128, g.mw_w[3], g.mw_b[3], am + AM_gZ3m, stream));
        TR(wgrad_n(L.gZ3m, 64, L.M2, 64, cap, dS, V, cap, 64, 64, AM_gZ3m, AM_M2, g.mw_w[2], 64, g.mw_b[2]));
        TR(dgrad(L.gZ3m, 64, cap, dS, V, cap, IM_MW2T, 64, 64, L.M2, 64, L.dM2, 64, AM_dM2));
        TR(wgrad_n(L.dM2, 64, L.M1, 64, cap, dS, V, cap, 64, 64, AM_dM2, AM_M1, g.mw_w[1], 64, g.mw_b[1]));
        TR(dgrad(L.dM2, 64, cap, dS, V, cap, IM_MW1T, 64, 64, L.M1, 64, L.dM1, 64, AM_dM1));
        TR(wgrad_n(L.dM1, 64, L.X6, 48, cap, dS, V, cap, 64, 48, AM_dM1, AM_X6, L.tmpWfd, 48, nullptr));
        train_w0fd_grad_kernel<<<(64 * 48 + 255) / 256, 256, 0, side_g ? side_stream : st>>>(L.tmpWfd, g.mw_w[0]);      // (behind the weight gradient that wrote tmpWfd)
        TR(sum_views_dc(L.dM1, 64, V, cap, L.tc + TC_S, 64, L.gpre, 64, am + AM_gpre, st));
        TR(wgrad_n(L.gpre, 64, L.CF, 128, cap, dS, 1, 0, 64, 128, AM_gpre, AM_CF, g.mw_w[0] + 45, 176, g.mw_b[0]));
        TR(dgrad(L.dM1, 64, cap, dS, V, cap, IM_MW0FDT, 48, 64, nullptr, 0, L.gX6, 48, -1));
        TR(dgrad(L.gpre, 64, cap, dS, 1, 0, IM_MW0CFT, 128, 64, nullptr, 0, L.tmpCF, 128, -1));
        TR(mark());                                                      // (tmpCF is added to gCF by stage 7's kernel)
        // ---- 6. pixel gather + upsample + conv pyramid.  Nothing downstream reads this stage's results (the reference-view CNN's weight gradients),
        //      and it is 0.6 ms of small latency-bound kernels: it runs on a side stream of the library beside stages 7 - 11 (forked here, joined
        //      before the call returns; HNR_TRAIN_SIDE=0: in line on the caller's stream)
        hipStream_t s6 = st;
        if (side_on) {
            HNR_HIP_CHECK(hipEventRecord(side_fork, st));
            // bit 4: the image branch on the pack stream beside the weight gradients' stream -- three busy queues
            s6 = (side.on & 16) ? side.stream_w : side_stream;
            HNR_HIP_CHECK(hipStreamWaitEvent(s6, side_fork, 0));
            forked = true;
            forked6w = (side.on & 16) != 0;
        }
        HNR_HIP_CHECK(hipMemsetAsync(L.g_pyr, 0, L.fm_elems * 4, s6));
        HNR_HIP_CHECK(hipMemsetAsync(L.g_fm, 0, (size_t)V * p->H * p->W * 48 * 4, s6));
        train_bbox_init_kernel<<<1, 64, 0, s6>>>(L.bbox, V, p->H, p->W);
        TR(hnr_proj_rows_bwd(o->d_sample_loc_w, L.vs_item, o->d_counts, vw->d_w2c, vw->d_intrinsic, V, p->H, p->W, cap, L.gF, 48, L.gX6, 48, L.g_fm, L.bbox, L.g_pyr, L.key_scratch,
                             L.sort_scratch, (int64_t)L.sort_bytes, (void *)s6));
        TR(image_features_bwd_bbox(vw->d_images, V, p->H, p->W, w->conv_w, sl, L.fm_scratch, L.g_pyr, g.conv_w, g.conv_b, L.bbox, (void *)s6));
    } else {
        TR(mark());                                                      // (X7 = [colfeat[:45] | 0]: gX7[:, :45] is added to gCF by stage 7's kernel)
    }
    TR(mark());
    // ---- 7. colour-feature branch
    //      gCF = (gCF + d colfeat from the merge-weight MLP's first layer [or, without views, from the mix-up input]) * LeakyReLU'(CF), and its maximum
    TR(dleaky_add_dc(L.gCF, 128, V > 0 ? L.tmpCF : L.gX7, V > 0 ? 128 : 92, V > 0 ? 128 : 45, L.CF, 128, cap, L.tc + TC_S, 128, sl, am + AM_gCF, st));
    TR(wgrad_n(L.gCF, 128, L.T2, 128, cap, dS, 1, 0, 128, 128, AM_gCF, AM_T2, g.cf_w[2], 128, g.cf_b[2]));
    TR(dgrad(L.gCF, 128, cap, dS, 1, 0, IM_CF2T, 128, 128, L.T2, 128, L.dT2, 128, AM_dT2));
    TR(wgrad_n(L.dT2, 128, L.T1, 128, cap, dS, 1, 0, 128, 128, AM_dT2, AM_T1, g.cf_w[1], 128, g.cf_b[1]));
    TR(dgrad(L.dT2, 128, cap, dS, 1, 0, IM_CF1T, 128, 128, L.T1, 128, L.dT1, 128, AM_dT1));
    TR(wgrad_n(L.dT1, 128, L.X5, 280, cap, dS, 1, 0, 128, 280, AM_dT1, AM_X5, g.cf_w[0], 280, g.cf_b[0]));
    TR(dgrad(L.dT1, 128, cap, dS, 1, 0, IM_CF0T, 256, 128, nullptr, 0, L.gX5, 256, -1));
    TR(mark());
    // ---- 8. K-sums + alpha branch
    {
        KsumPadArgs a;
        const int blocks_g = cdiv(cap, 16) + 2;
        a.H4 = L.H4; a.aux = L.chain_ws + (size_t)blocks_g * 4 * CH_XP_GROUP; a.alpha_w = w->alpha_w; a.alpha_b = w->alpha_b;
        a.counts = reinterpret_cast<const unsigned long long *>(o->d_counts); a.gX5 = L.gX5; a.ldg5 = 256; a.g_sigma = L.g_sigma; a.slope = sl;
        a.gZ4 = L.gZ4; a.g_wagg = L.g_wagg; a.g_alpha_w = g.alpha_w; a.g_alpha_b = g.alpha_b; a.absmax = am + AM_gZ4;
        int nb = cdiv(cap, 16); if (nb > 512) nb = 512; if (nb < 1) nb = 1;
        train_ksum_bwd_kernel<<<nb, 1024, 0, st>>>(a);
        HNR_LAUNCH_CHECK();
    }
    TR(mark());
    // ---- 9. block3
    TR(wgrad_n(L.gZ4, 256, L.H3, 256, rows, dM, 1, 0, 256, 256, AM_gZ4, AM_H3, g.block3_2_w, 256, g.block3_2_b));
    // (the three input gradients of the per-neighbour chain read their LeakyReLU' from the forward's sign words: 32 B per row instead of 1 KiB)
    const long long hb = (long long)L.rows_cap * 8;
    TR(h2lin_dgrad_bits(L.gZ4, 256, rows, dM, L.img[IM_B32T], 256, 256, sl, L.hbits + 2 * hb, L.dZ3, 256, am + AM_dZ3, stream));
    TR(wgrad_n(L.dZ3, 256, L.X3, 264, rows, dM, 1, 0, 256, 263, AM_dZ3, AM_X3, g.block3_0_w, 263, g.block3_0_b));
    TR(h2lin_dgrad_bits(L.dZ3, 256, rows, dM, L.img[IM_B30T], 256, 256, sl, L.hbits + hb, L.gX3, 264, am + AM_dZ2, stream));
    {
        int nb = cdiv(rows, 16); if (nb > 2048) nb = 2048;
        train_extras_dgrad_kernel<<<nb, 256, 0, st>>>(L.dZ3, w->block3_0_w, L.tc + TC_M8, L.gX3);
        HNR_LAUNCH_CHECK();
    }
    TR(mark());
    if (side_z) HNR_HIP_CHECK(hipStreamWaitEvent(st, side.join_z, 0));        // the point-gradient buffers are clear
    if (d_g_conf_coefficient) {
        // the empty slots' share of d conf_coefficient lands on point 0 (the reference's index clamp)
        train_conf0_partial_kernel<<<512, 256, 0, st>>>(d_g_conf_coefficient, o->d_sample_pidx, (long long)R * SR * K, L.conf0);
        train_conf0_final_kernel<<<1, 256, 0, st>>>(L.conf0, 512, gc->d_conf);
    }
    // ---- 10. rows -> touched point: the point-major row list was built in the forward pass (unique_points_dc); both per-point reductions add a
    //          point's rows in ascending row order (deterministic)
    TR(hnr_gather_rows_bwd_rows(o->d_sample_pidx, cam->d_raydir, L.vs_item, L.vs_off, L.vs_cnt, o->d_counts, SR, K, cap, L.gX3, 264, L.g_wagg, o->d_weight, d_g_conf_coefficient,
                                L.G8, stream));
    TR(mark());                                                      // (the rows' small gradients G8 are summed per point together with block1's rows below)
    // ---- 11. block1 (first layer split: 60 distance columns per row + the per-point table)
    TR(wgrad_n(L.gX3, 264, L.H1, 256, rows, dM, 1, 0, 256, 256, AM_dZ2, AM_H1, g.block1_2_w, 256, g.block1_2_b));
    TR(h2lin_dgrad_bits(L.gX3, 264, rows, dM, L.img[IM_B12T], 256, 256, sl, L.hbits, L.dZ1, 256, am + AM_dZ1, stream));
    TR(wgrad_n(L.dZ1, 256, L.Xd, 64, rows, dM, 1, 0, 256, 60, AM_dZ1, AM_ONE, g.block1_0_w + 224, 284, g.block1_0_b));
    TR(segment_sum_rows_csr_dc(L.dZ1, 256, L.row_list, L.seg_start, L.seg_cnt, 256, (int)ucap, L.tc + TC_U, L.gTu, 256, L.G8, 8, 8, L.P8, 8, am + AM_gTu, st));
    TR(point_small_grads_dc(L.P8, L.ulist, (int)ucap, L.tc + TC_U, gc->d_conf, gc->d_dir, gc->d_color, st));
    if (side_g) {                                                          // (max |E| is read by the weight gradient below only: same stream)
        HNR_HIP_CHECK(hipEventRecord(side.fork_g, st));
        HNR_HIP_CHECK(hipStreamWaitEvent(side_stream, side.fork_g, 0));
        forked = true;
        TR(hnr_absmax(L.E, 224, ucap, dU, 1, 0, 224, am + AM_E, (void *)side_stream));
    } else {
        TR(hnr_absmax(L.E, 224, ucap, dU, 1, 0, 224, am + AM_E, stream));
    }
    TR(wgrad_n(L.gTu, 256, L.E, 224, ucap, dU, 1, 0, 256, 224, AM_gTu, AM_E, g.block1_0_w, 284, nullptr));
    TR(dgrad(L.gTu, 256, ucap, dU, 1, 0, IM_TABT, 224, 256, nullptr, 0, L.gE, 224, -1));
    TR(point_rows_bwd_dc(L.gE, 224, L.E, 224, L.ulist, (int)ucap, L.tc + TC_U, gc->d_emb, st));
    if (forked) {                                                          // the side stream's work (image branch, weight gradients) is part of this call
        HNR_HIP_CHECK(hipEventRecord(side_join, side_stream));
        HNR_HIP_CHECK(hipStreamWaitEvent(st, side_join, 0));
        if (forked6w) { HNR_HIP_CHECK(hipEventRecord(side.ev_w[1], side.stream_w)); HNR_HIP_CHECK(hipStreamWaitEvent(st, side.ev_w[1], 0)); }
    }
    TR(mark());
    HNR_LAUNCH_CHECK();
    guard.armed = false;
    return HNR_OK;
}
