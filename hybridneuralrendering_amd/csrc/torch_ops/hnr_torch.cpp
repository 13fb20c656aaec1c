// TORCH_LIBRARY(hnr): dispatcher-visible PyTorch ops over the C ABI of libhnr_hip.so (include/hnr.h) -- the "op layer" of SURVEY 8b.
//
// The reference reaches its kernels through pycuda launches inside nn.Module.forward (models/neural_points/query_point_indices_worldcoords.py:80-93,
// :540-711) and plain torch ops (models/neural_points/neural_points.py:709-733, models/aggregators/point_aggregators.py:1427-1522,
// models/rendering/diff_ray_marching.py:508-557); torch autograd differentiates the latter (models/mvs_points_volumetric_model.py:111-131).
// Here the same path is four ops with schemas, so it is visible to the dispatcher (torch.ops.hnr.*, torch.library, export / compile tooling)
// without going through Python ctypes marshalling:
//   hnr::grid_build / hnr::grid_free     hnr_grid_build / hnr_grid_free          (build_occ_vox, :540-602)
//   hnr::march_query                     hnr_march_query                         (query_grid_point_index, :605-711)
//   hnr::render_forward                  hnr_render_forward                      (NeuralPointsRayMarching.forward + fill_invalid, eval)
//   hnr::render_train                    hnr_render_train_forward + _backward    (the same in train mode; autograd formula registered)
// Nothing is computed here: every op validates its tensors, fills the C structs, takes the current HIP stream and calls the library.  Errors
// of the library surface as c10::Error with hnr_last_error() as the message.  Host code only (no kernels): built by g++ against libtorch.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <torch/csrc/autograd/custom_function.h>

#include <cstring>
#include <vector>

#include "hnr.h"

namespace {

using at::Tensor;
using OptT = std::optional<Tensor>;

void hnr_check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", hnr_last_error());
}

void *cur_stream(const Tensor &t) { return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

const Tensor &need(const Tensor &t, const char *name, at::ScalarType dt)
{
    TORCH_CHECK(t.defined() && t.is_cuda(), "hnr: `", name, "` must be a GPU tensor");
    TORCH_CHECK(t.scalar_type() == dt, "hnr: `", name, "` must be ", c10::toString(dt), ", got ", c10::toString(t.scalar_type()));
    TORCH_CHECK(t.is_contiguous(), "hnr: `", name, "` must be contiguous");
    return t;
}
template <class T> const T *cptr(const Tensor &t, const char *name, at::ScalarType dt) { return reinterpret_cast<const T *>(need(t, name, dt).data_ptr()); }
const float *fptr(const Tensor &t, const char *name) { return cptr<float>(t, name, at::kFloat); }
const float *fptr_opt(const OptT &t, const char *name) { return (t.has_value() && t->defined()) ? fptr(*t, name) : nullptr; }

Tensor new_f32(at::IntArrayRef shape, const Tensor &like) { return at::empty(shape, like.options().dtype(at::kFloat)); }

// ---- hnr::grid_build(Tensor xyz, float[] origin, float[] cell, int[] dims, int[] query_size, int P, int max_o) -> int   (opaque handle)
int64_t grid_build(const Tensor &xyz, c10::ArrayRef<double> origin, c10::ArrayRef<double> cell, at::IntArrayRef dims, at::IntArrayRef query_size, int64_t P,
                   int64_t max_o)
{
    TORCH_CHECK(origin.size() == 3 && cell.size() == 3 && dims.size() == 3 && query_size.size() == 3, "hnr::grid_build: origin / cell / dims / query_size hold 3 values");
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "hnr::grid_build: xyz must be [N,3]");
    hnr_grid_params p;
    for (int i = 0; i < 3; ++i) { p.origin[i] = (float)origin[i]; p.cell[i] = (float)cell[i]; p.dims[i] = (int)dims[i]; p.query_size[i] = (int)query_size[i]; }
    p.P = (int)P; p.max_o = (int)max_o;
    hnr_grid *g = nullptr;
    hnr_check(hnr_grid_build(fptr(xyz, "xyz"), (int)xyz.size(0), &p, cur_stream(xyz), &g), "hnr_grid_build");
    return (int64_t)reinterpret_cast<intptr_t>(g);
}
void grid_free(int64_t handle) { hnr_check(hnr_grid_free(reinterpret_cast<hnr_grid *>((intptr_t)handle)), "hnr_grid_free"); }

// ---- hnr::march_query -> (sample_pidx [R,SR,K] i32, sample_loc_w [R,SR,3], ray_nsamp [R] i32, ray_mask [R] i8, counts [HNR_NCOUNTS] i64)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> march_query(int64_t grid, const Tensor &campos, const Tensor &raydir, const Tensor &tmid, int64_t SR, int64_t K,
                                                               double radius2, at::IntArrayRef kernel_size, bool pad, int64_t knn_order)
{
    TORCH_CHECK(raydir.dim() == 2 && raydir.size(1) == 3, "hnr::march_query: raydir must be [R,3]");
    TORCH_CHECK(kernel_size.size() == 3, "hnr::march_query: kernel_size holds 3 values");
    const int64_t R = raydir.size(0);
    hnr_query_params q;
    std::memset(&q, 0, sizeof(q));
    q.R = (int)R; q.D = (int)tmid.size(-1); q.SR = (int)SR; q.K = (int)K;
    for (int i = 0; i < 3; ++i) q.kernel_size[i] = (int)kernel_size[i];
    q.radius2 = (float)radius2;
    q.tmid_stride = tmid.dim() == 1 ? 0 : (int)tmid.size(1);
    q.pad_outputs = pad ? 1 : 0;
    q.knn_order = (int)knn_order;
    auto oi = raydir.options().dtype(at::kInt);
    Tensor pidx = at::empty({R, SR, K}, oi), loc = new_f32({R, SR, 3}, raydir), nsamp = at::empty({R}, oi);
    Tensor mask = at::empty({R}, raydir.options().dtype(at::kChar)), counts = at::empty({HNR_NCOUNTS}, raydir.options().dtype(at::kLong));
    Tensor work = at::empty({hnr_query_work_elems((int)R, (int)SR)}, oi);
    hnr_check(hnr_march_query(reinterpret_cast<const hnr_grid *>((intptr_t)grid), fptr(campos, "campos"), fptr(raydir, "raydir"), fptr(tmid, "tmid"), &q,
                              pidx.data_ptr<int32_t>(), loc.data_ptr<float>(), nsamp.data_ptr<int32_t>(), mask.data_ptr<int8_t>(), work.data_ptr<int32_t>(),
                              counts.data_ptr<int64_t>(), cur_stream(raydir)),
              "hnr_march_query");
    return {pidx, loc, nsamp, mask, counts};
}

// ---- hnr::render_forward: the frame as one library call.  packed = [chain, mlp_cf, mlp_mw, mlp_mx] images (uint8) + [mw_last_w, mw_last_b, fin_w, fin_b] (f32).
//      Returns [raycolor [R,3], opacity [R,SR], is_background [R], ray_mask [R] i8, decoded [R,SR,4], sample_pidx, sample_loc_w, ray_nsamp, counts, status [2] i32]
std::vector<Tensor> render_forward(int64_t grid, const Tensor &xyz, const Tensor &conf, const Tensor &dir, const Tensor &color, const Tensor &point_table,
                                   const OptT &point_records, c10::ArrayRef<Tensor> packed, const Tensor &campos, const Tensor &camrot, const Tensor &raydir,
                                   const Tensor &tmid, const Tensor &bg_color, const OptT &w2c, const OptT &intrinsic, const OptT &campos_nearest, const OptT &featmap,
                                   const OptT &frame_w, int64_t SR, int64_t K, at::IntArrayRef kernel_size, double radius2, double vsize_z, int64_t raydist_mode_unit,
                                   int64_t knn_order, double slope, int64_t cap_samples)
{
    TORCH_CHECK(packed.size() == 8, "hnr::render_forward: packed = [chain, mlp_cf, mlp_mw, mlp_mx, mw_last_w, mw_last_b, fin_w, fin_b]");
    TORCH_CHECK(raydir.dim() == 2 && raydir.size(1) == 3 && kernel_size.size() == 3, "hnr::render_forward: raydir [R,3], kernel_size [3]");
    const int64_t R = raydir.size(0);
    const bool views = featmap.has_value() && featmap->defined();
    hnr_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.R = (int)R; p.SR = (int)SR; p.K = (int)K; p.D = (int)tmid.size(-1);
    p.tmid_stride = tmid.dim() == 1 ? 0 : (int)tmid.size(1);
    for (int i = 0; i < 3; ++i) p.kernel_size[i] = (int)kernel_size[i];
    p.radius2 = (float)radius2; p.vsize_z = (float)vsize_z; p.raydist_mode_unit = (int)raydist_mode_unit;
    p.V = views ? (int)featmap->size(0) : 0;
    p.cap_samples = (int)(cap_samples > 0 ? cap_samples : R * SR);
    p.knn_order = (int)knn_order;
    const int64_t nbytes = hnr_render_workspace_bytes(&p);
    TORCH_CHECK(nbytes >= 0, "hnr_render_workspace_bytes: ", hnr_last_error());
    Tensor ws = at::empty({nbytes + 256}, raydir.options().dtype(at::kByte));
    char *wsp = reinterpret_cast<char *>(ws.data_ptr());
    wsp += (256 - (reinterpret_cast<uintptr_t>(wsp) & 255)) & 255;
    hnr_render_cloud cl{fptr(xyz, "xyz"), fptr(conf, "conf"), fptr(dir, "dir"), fptr(color, "color"), fptr(point_table, "point_table"), (int)point_table.stride(0),
                        fptr_opt(point_records, "point_records")};
    hnr_render_weights wt{need(packed[0], "packed chain", at::kByte).data_ptr(), need(packed[1], "packed mlp_cf", at::kByte).data_ptr(),
                          need(packed[2], "packed mlp_mw", at::kByte).data_ptr(), need(packed[3], "packed mlp_mx", at::kByte).data_ptr(),
                          fptr(packed[4], "mw_last_w"), fptr(packed[5], "mw_last_b"), fptr(packed[6], "fin_w"), fptr(packed[7], "fin_b"), (float)slope};
    hnr_render_camera cam{fptr(campos, "campos"), fptr(camrot, "camrot"), fptr(raydir, "raydir"), fptr(tmid, "tmid"), fptr(bg_color, "bg_color")};
    hnr_render_views vw;
    std::memset(&vw, 0, sizeof(vw));
    if (views) {
        TORCH_CHECK(w2c.has_value() && intrinsic.has_value() && campos_nearest.has_value(), "hnr::render_forward: featmap needs w2c, intrinsic, campos_nearest");
        vw.d_w2c = fptr(*w2c, "w2c"); vw.d_intrinsic = fptr(*intrinsic, "intrinsic"); vw.d_campos_nearest = fptr(*campos_nearest, "campos_nearest");
        vw.d_featmap = fptr(*featmap, "featmap"); vw.H = (int)featmap->size(1); vw.W = (int)featmap->size(2);
        vw.d_frame_w = fptr_opt(frame_w, "frame_w");
    }
    auto oi = raydir.options().dtype(at::kInt);
    Tensor col = new_f32({R, 3}, raydir), opa = new_f32({R, SR}, raydir), isbg = new_f32({R}, raydir), decoded = new_f32({R, SR, 4}, raydir);
    Tensor mask = at::empty({R}, raydir.options().dtype(at::kChar)), pidx = at::empty({R, SR, K}, oi), loc = new_f32({R, SR, 3}, raydir), nsamp = at::empty({R}, oi);
    Tensor counts = at::empty({HNR_NCOUNTS}, raydir.options().dtype(at::kLong)), status = at::empty({2}, oi);
    hnr_render_outputs out;
    std::memset(&out, 0, sizeof(out));
    out.d_raycolor = col.data_ptr<float>(); out.d_opacity = opa.data_ptr<float>(); out.d_is_background = isbg.data_ptr<float>();
    out.d_ray_mask = mask.data_ptr<int8_t>(); out.d_decoded = decoded.data_ptr<float>(); out.d_sample_pidx = pidx.data_ptr<int32_t>();
    out.d_sample_loc_w = loc.data_ptr<float>(); out.d_ray_nsamp = nsamp.data_ptr<int32_t>(); out.d_counts = counts.data_ptr<int64_t>();
    out.d_status = status.data_ptr<int32_t>();
    auto st = c10::hip::getCurrentHIPStream(raydir.device().index());
    hnr_check(hnr_render_forward(reinterpret_cast<const hnr_grid *>((intptr_t)grid), &p, &cl, &wt, &cam, views ? &vw : nullptr, wsp, nbytes, &out, (void *)st.stream()),
              "hnr_render_forward");
    // (the workspace is freed into the caching allocator's pool of THIS stream when the op returns: a later allocation that reuses the block is
    //  stream-ordered behind the launches above; the library joins its side streams back into the caller's stream before it returns)
    return {col, opa, isbg, mask, decoded, pidx, loc, nsamp, counts, status};
}

// ---- hnr::render_train: forward in train mode; backward through the registered autograd formula.
// Three ops: hnr::render_train_fwd / hnr::render_train_bwd are the two library calls (CUDA kernels + shape functions registered from Python), and
// hnr::render_train is the differentiable op whose Autograd kernel is a C++ autograd::Function that REDISPATCHES to those two -- so tracing
// (FakeTensorMode, torch.compile, export) sees ops with shape functions on both sides of the tape and never runs a kernel.
// weights: the 44 tensors of hnr_train_weights in struct order (block1.0 w, b, block1.2 w, b, block3.0 w, b, block3.2 w, b, alpha w, b, cf w x3, cf b x3,
// mw w x4, mw b x4, mx w x3, mx b x3, fin w, b, conv w x6, conv b x6).
constexpr int N_TW = 44, N_OUT = 13;
// inputs: [xyz, emb, conf, dir, color, campos, camrot, raydir, tmid, bg_color, w2c, intrinsic, campos_nearest, images, frame_w]; the last five are optional
enum { I_XYZ, I_EMB, I_CONF, I_DIR, I_COLOR, I_CAMPOS, I_CAMROT, I_RAYDIR, I_TMID, I_BG, I_W2C, I_INTR, I_CAMN, I_IMG, I_FW, N_IN };
using OptList = c10::List<std::optional<Tensor>>;

void fill_train_weights(hnr_train_weights &w, c10::ArrayRef<Tensor> t)
{
    const float **slot = reinterpret_cast<const float **>(&w);
    static_assert(sizeof(hnr_train_weights) == N_TW * sizeof(void *), "hnr_train_weights is 44 pointers");
    for (int i = 0; i < N_TW; ++i) slot[i] = t[i].defined() ? fptr(t[i], "train weight") : nullptr;
}
hnr_render_outputs outputs_of(c10::ArrayRef<Tensor> o)
{
    hnr_render_outputs out;
    std::memset(&out, 0, sizeof(out));
    out.d_raycolor = o[0].data_ptr<float>(); out.d_opacity = o[1].data_ptr<float>(); out.d_is_background = o[2].data_ptr<float>(); out.d_blend_weight = o[3].data_ptr<float>();
    out.d_ray_mask = o[4].data_ptr<int8_t>(); out.d_decoded = o[5].data_ptr<float>(); out.d_sample_pidx = o[6].data_ptr<int32_t>(); out.d_sample_loc_w = o[7].data_ptr<float>();
    out.d_ray_nsamp = o[8].data_ptr<int32_t>(); out.d_counts = o[9].data_ptr<int64_t>(); out.d_status = o[10].data_ptr<int32_t>(); out.d_weight = o[11].data_ptr<float>();
    out.d_conf_coefficient = o[12].data_ptr<float>();
    return out;
}
// None -> undefined; the point buffers in the library's flat shapes (the reference's are [1,N,32], [1,N,1], ...)
std::vector<Tensor> unpack(const OptList &l)
{
    TORCH_CHECK(l.size() == N_IN, "hnr::render_train: 15 inputs expected");
    std::vector<Tensor> v;
    for (size_t i = 0; i < l.size(); ++i) { std::optional<Tensor> t = l.get(i); v.push_back(t.has_value() ? *t : Tensor()); }
    TORCH_CHECK(v[I_XYZ].defined() && v[I_RAYDIR].defined(), "hnr::render_train: xyz and raydir are required");
    const int64_t N = v[I_XYZ].size(-2);
    v[I_XYZ] = v[I_XYZ].reshape({N, 3}); v[I_EMB] = v[I_EMB].reshape({N, 32}); v[I_CONF] = v[I_CONF].reshape({N});
    v[I_DIR] = v[I_DIR].reshape({N, 3}); v[I_COLOR] = v[I_COLOR].reshape({N, 3});
    return v;
}
struct TrainStructs {
    hnr_train_params p;
    hnr_train_cloud cl;
    hnr_train_weights w;
    hnr_render_camera cam;
    hnr_train_views vw;
    bool views;
};
TrainStructs train_structs(const std::vector<Tensor> &in, c10::ArrayRef<Tensor> weights, int64_t SR, at::IntArrayRef kernel_size, double radius2, double vsize_z,
                           int64_t raydist_mode_unit, int64_t knn_order, double slope, int64_t cap_samples)
{
    TORCH_CHECK(weights.size() == N_TW && kernel_size.size() == 3, "hnr::render_train: 44 weights, kernel_size [3]");
    const Tensor &raydir = in[I_RAYDIR], &tmid = in[I_TMID], &img = in[I_IMG];
    const int64_t R = raydir.size(0);
    TrainStructs t;
    std::memset(&t.p, 0, sizeof(t.p));
    t.p.R = (int)R; t.p.SR = (int)SR; t.p.K = 8; t.p.D = (int)tmid.size(-1);
    t.p.tmid_stride = tmid.dim() == 1 ? 0 : (int)tmid.size(1);
    for (int i = 0; i < 3; ++i) t.p.kernel_size[i] = (int)kernel_size[i];
    t.p.radius2 = (float)radius2; t.p.vsize_z = (float)vsize_z; t.p.raydist_mode_unit = (int)raydist_mode_unit;
    t.views = img.defined() && img.numel() > 0;
    t.p.V = t.views ? (int)img.size(0) : 0; t.p.H = t.views ? (int)img.size(1) : 0; t.p.W = t.views ? (int)img.size(2) : 0;
    t.p.n_points = (int)in[I_XYZ].size(0);
    t.p.cap_samples = (int)(cap_samples > 0 ? cap_samples : R * SR);
    t.p.knn_order = (int)knn_order; t.p.slope = (float)slope;
    t.cl = hnr_train_cloud{fptr(in[I_XYZ], "xyz"), fptr(in[I_EMB], "emb"), fptr(in[I_CONF], "conf"), fptr(in[I_DIR], "dir"), fptr(in[I_COLOR], "color")};
    fill_train_weights(t.w, weights);
    t.cam = hnr_render_camera{fptr(in[I_CAMPOS], "campos"), fptr(in[I_CAMROT], "camrot"), fptr(raydir, "raydir"), fptr(tmid, "tmid"), fptr(in[I_BG], "bg_color")};
    t.vw = hnr_train_views{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (t.views) t.vw = hnr_train_views{fptr(in[I_W2C], "w2c"), fptr(in[I_INTR], "intrinsic"), fptr(in[I_CAMN], "campos_nearest"), fptr(img, "images"),
                                        in[I_FW].defined() && in[I_FW].numel() > 0 ? fptr(in[I_FW], "frame_w") : nullptr};
    return t;
}
char *aligned(const Tensor &ws)
{
    char *q = reinterpret_cast<char *>(ws.data_ptr());
    return q + ((256 - (reinterpret_cast<uintptr_t>(q) & 255)) & 255);
}

// hnr::render_train_fwd -> the 13 outputs (raycolor, opacity, is_background, blend_weight, ray_mask, decoded, sample_pidx, sample_loc_w, ray_nsamp, counts,
// status, weight, conf_coefficient) + the step's workspace (uint8), which hnr::render_train_bwd needs
std::vector<Tensor> render_train_fwd(int64_t grid, const OptList &in_l, c10::ArrayRef<Tensor> weights, const OptT &drop_lut, const OptT &ray_drop, int64_t SR,
                                     at::IntArrayRef kernel_size, double radius2, double vsize_z, int64_t raydist_mode_unit, int64_t knn_order, double slope, int64_t cap_samples)
{
    const std::vector<Tensor> in = unpack(in_l);
    TrainStructs t = train_structs(in, weights, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples);
    const Tensor &raydir = in[I_RAYDIR];
    const int64_t R = raydir.size(0), K = 8;
    const int64_t nbytes = hnr_render_train_workspace_bytes(&t.p);
    TORCH_CHECK(nbytes >= 0, "hnr_render_train_workspace_bytes: ", hnr_last_error());
    Tensor ws = at::empty({nbytes + 256}, raydir.options().dtype(at::kByte));
    auto oi = raydir.options().dtype(at::kInt);
    std::vector<Tensor> outs = {new_f32({R, 3}, raydir), new_f32({R, SR}, raydir), new_f32({R}, raydir), new_f32({R, SR}, raydir),
                                at::empty({R}, raydir.options().dtype(at::kChar)), new_f32({R, SR, 4}, raydir), at::empty({R, SR, K}, oi), new_f32({R, SR, 3}, raydir),
                                at::empty({R}, oi), at::empty({HNR_NCOUNTS}, raydir.options().dtype(at::kLong)), at::empty({2}, oi), new_f32({R, SR, K}, raydir),
                                new_f32({R, SR, K}, raydir)};
    hnr_render_outputs out = outputs_of(outs);
    const uint8_t *lut = (drop_lut.has_value() && drop_lut->defined()) ? cptr<uint8_t>(*drop_lut, "drop_lut", at::kByte) : nullptr;
    const uint8_t *rd = (ray_drop.has_value() && ray_drop->defined()) ? cptr<uint8_t>(*ray_drop, "ray_drop", at::kByte) : nullptr;
    hnr_check(hnr_render_train_forward(reinterpret_cast<const hnr_grid *>((intptr_t)grid), &t.p, &t.cl, &t.w, &t.cam, t.views ? &t.vw : nullptr, lut, rd, aligned(ws), nbytes,
                                       &out, cur_stream(raydir)),
              "hnr_render_train_forward");
    outs.push_back(ws);
    return outs;
}

// hnr::render_train_bwd -> [d emb [N,32], d conf [N], d dir [N,3], d color [N,3]] + 44 weight gradients (zero-size where the forward had no views)
std::vector<Tensor> render_train_bwd(const OptList &in_l, c10::ArrayRef<Tensor> weights, c10::ArrayRef<Tensor> fwd, const Tensor &g_col, const OptT &g_cc, int64_t SR,
                                     at::IntArrayRef kernel_size, double radius2, double vsize_z, int64_t raydist_mode_unit, int64_t knn_order, double slope, int64_t cap_samples)
{
    TORCH_CHECK(fwd.size() == N_OUT + 1, "hnr::render_train_bwd: the 14 tensors hnr::render_train_fwd returned");
    const std::vector<Tensor> in = unpack(in_l);
    TrainStructs t = train_structs(in, weights, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples);
    const int64_t N = in[I_XYZ].size(0);
    const Tensor &like = in[I_RAYDIR], &ws = fwd[N_OUT];
    std::vector<Tensor> g = {new_f32({N, 32}, like), new_f32({N}, like), new_f32({N, 3}, like), new_f32({N, 3}, like)};
    std::vector<Tensor> gw(N_TW), gw_ptr(N_TW);
    for (int i = 0; i < N_TW; ++i) {
        const bool image_branch = (i >= 16 && i < 24) || i >= 32;             // mw_w/b (aux_merge_weight_block), conv_w/b (aux_block_s*)
        const bool have = t.views || !image_branch;
        gw[i] = have ? at::empty_like(weights[i]) : at::empty({0}, weights[i].options());
        if (have) gw_ptr[i] = gw[i];
    }
    hnr_train_weights wg;
    fill_train_weights(wg, gw_ptr);
    hnr_render_outputs out = outputs_of(fwd.slice(0, N_OUT));
    hnr_train_cloud_grads cg{g[0].data_ptr<float>(), g[1].data_ptr<float>(), g[2].data_ptr<float>(), g[3].data_ptr<float>()};
    const Tensor gc = g_col.contiguous();
    Tensor gcc;
    if (g_cc.has_value() && g_cc->defined()) gcc = g_cc->contiguous();
    const int64_t nbytes = hnr_render_train_workspace_bytes(&t.p);
    hnr_check(hnr_render_train_backward(&t.p, &t.cl, &t.w, &t.cam, t.views ? &t.vw : nullptr, aligned(ws), nbytes, &out, fptr(gc, "grad coarse_raycolor"),
                                        gcc.defined() ? fptr(gcc, "grad conf_coefficient") : nullptr, &cg, &wg, cur_stream(like)),
              "hnr_render_train_backward");
    g.insert(g.end(), gw.begin(), gw.end());
    return g;
}

std::vector<Tensor> call_fwd(int64_t grid, const OptList &in, at::TensorList weights, const OptT &drop_lut, const OptT &ray_drop, int64_t SR, at::IntArrayRef ks, double r2,
                             double vz, int64_t rmu, int64_t order, double slope, int64_t cap)
{
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("hnr::render_train_fwd", "")
                         .typed<std::vector<Tensor>(int64_t, const OptList &, at::TensorList, const OptT &, const OptT &, int64_t, at::IntArrayRef, double, double, int64_t,
                                                    int64_t, double, int64_t)>();
    return op.call(grid, in, weights, drop_lut, ray_drop, SR, ks, r2, vz, rmu, order, slope, cap);
}
std::vector<Tensor> call_bwd(const OptList &in, at::TensorList weights, at::TensorList fwd, const Tensor &g_col, const OptT &g_cc, int64_t SR, at::IntArrayRef ks, double r2,
                             double vz, int64_t rmu, int64_t order, double slope, int64_t cap)
{
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("hnr::render_train_bwd", "")
                         .typed<std::vector<Tensor>(const OptList &, at::TensorList, at::TensorList, const Tensor &, const OptT &, int64_t, at::IntArrayRef, double, double,
                                                    int64_t, int64_t, double, int64_t)>();
    return op.call(in, weights, fwd, g_col, g_cc, SR, ks, r2, vz, rmu, order, slope, cap);
}

struct RenderTrainFn : public torch::autograd::Function<RenderTrainFn> {
    // tensor arguments first (15 inputs, 44 weights: the flat order backward() returns gradients in), then the rest
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, at::TensorList in, at::TensorList weights, int64_t grid, OptT drop_lut, OptT ray_drop,
                                                  int64_t SR, std::vector<int64_t> ks, double r2, double vz, int64_t rmu, int64_t order, double slope, int64_t cap)
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        OptList in_l;
        for (const Tensor &t : in) in_l.push_back((t.defined() && t.numel() > 0) ? std::optional<Tensor>(t.detach()) : std::optional<Tensor>());
        std::vector<Tensor> wd;
        for (const Tensor &t : weights) wd.push_back(t.detach());
        std::vector<Tensor> outs = call_fwd(grid, in_l, wd, drop_lut, ray_drop, SR, ks, r2, vz, rmu, order, slope, cap);
        std::vector<Tensor> save;
        for (const Tensor &t : in) save.push_back(t);
        for (const Tensor &t : weights) save.push_back(t);
        for (const Tensor &t : outs) save.push_back(t);
        ctx->save_for_backward(save);
        ctx->saved_data["ks"] = ks; ctx->saved_data["SR"] = SR; ctx->saved_data["r2"] = r2; ctx->saved_data["vz"] = vz; ctx->saved_data["rmu"] = rmu;
        ctx->saved_data["order"] = order; ctx->saved_data["slope"] = slope; ctx->saved_data["cap"] = cap;
        std::vector<int64_t> shp;
        for (int i : {I_EMB, I_CONF, I_DIR, I_COLOR}) { auto sz = in[i].sizes(); shp.push_back((int64_t)sz.size()); shp.insert(shp.end(), sz.begin(), sz.end()); }
        ctx->saved_data["leaf_shapes"] = shp;
        torch::autograd::variable_list out(outs.begin(), outs.begin() + N_OUT);
        std::vector<Tensor> nd(out.begin() + 1, out.begin() + 12);            // everything but raycolor (0) and conf_coefficient (12) is returned detached
        ctx->mark_non_differentiable(nd);
        return out;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list go)
    {
        const auto saved = ctx->get_saved_variables();
        TORCH_CHECK(saved.size() == N_IN + N_TW + N_OUT + 1, "hnr::render_train: saved state missing");
        OptList in_l;
        for (int i = 0; i < N_IN; ++i) in_l.push_back((saved[i].defined() && saved[i].numel() > 0) ? std::optional<Tensor>(saved[i].detach()) : std::optional<Tensor>());
        std::vector<Tensor> wd, fwd;
        for (int i = 0; i < N_TW; ++i) wd.push_back(saved[N_IN + i].detach());
        for (int i = 0; i <= N_OUT; ++i) fwd.push_back(saved[N_IN + N_TW + i]);
        at::AutoDispatchBelowADInplaceOrView guard;
        const Tensor g_col = go[0].defined() ? go[0] : at::zeros_like(fwd[0]);
        const OptT g_cc = go[12].defined() ? OptT(go[12]) : OptT();
        std::vector<Tensor> g = call_bwd(in_l, wd, fwd, g_col, g_cc, ctx->saved_data["SR"].toInt(), ctx->saved_data["ks"].toIntVector(), ctx->saved_data["r2"].toDouble(),
                                         ctx->saved_data["vz"].toDouble(), ctx->saved_data["rmu"].toInt(), ctx->saved_data["order"].toInt(), ctx->saved_data["slope"].toDouble(),
                                         ctx->saved_data["cap"].toInt());
        auto shp = ctx->saved_data["leaf_shapes"].toIntVector();
        torch::autograd::variable_list res;
        size_t pos = 0;
        for (int i = 0; i < N_IN; ++i) {
            if (i >= I_EMB && i <= I_COLOR) {
                const int64_t nd = shp[pos++];
                std::vector<int64_t> sz(shp.begin() + pos, shp.begin() + pos + nd); pos += nd;
                res.push_back(g[i - I_EMB].reshape(sz));
            } else res.emplace_back();
        }
        for (int i = 0; i < N_TW; ++i) res.push_back(g[4 + i].numel() > 0 ? g[4 + i] : Tensor());
        for (int i = 0; i < 11; ++i) res.emplace_back();                        // grid, drop_lut, ray_drop, the scalars
        return res;
    }
};

// an absent optional input travels through the autograd node as an EMPTY tensor on the rays' device (its input list must hold defined tensors)
std::vector<Tensor> render_train_autograd(int64_t grid, const OptList &in_l, c10::ArrayRef<Tensor> weights, const OptT &drop_lut, const OptT &ray_drop, int64_t SR,
                                          at::IntArrayRef kernel_size, double radius2, double vsize_z, int64_t raydist_mode_unit, int64_t knn_order, double slope,
                                          int64_t cap_samples)
{
    TORCH_CHECK(in_l.size() == N_IN && weights.size() == N_TW, "hnr::render_train: 15 inputs and 44 weights expected");
    std::vector<Tensor> in;
    for (size_t i = 0; i < in_l.size(); ++i) { std::optional<Tensor> t = in_l.get(i); in.push_back(t.has_value() ? *t : Tensor()); }
    TORCH_CHECK(in[I_RAYDIR].defined(), "hnr::render_train: raydir is required");
    for (auto &t : in) if (!t.defined()) t = at::empty({0}, in[I_RAYDIR].options().dtype(at::kFloat));
    return RenderTrainFn::apply(at::TensorList(in), weights, grid, drop_lut, ray_drop, SR, kernel_size.vec(), radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples);
}
// no-grad / inference dispatch of the same op: the forward call alone (the workspace dies with the call)
std::vector<Tensor> render_train_nograd(int64_t grid, const OptList &in_l, c10::ArrayRef<Tensor> weights, const OptT &drop_lut, const OptT &ray_drop, int64_t SR,
                                        at::IntArrayRef kernel_size, double radius2, double vsize_z, int64_t raydist_mode_unit, int64_t knn_order, double slope, int64_t cap_samples)
{
    std::vector<Tensor> outs = call_fwd(grid, in_l, weights, drop_lut, ray_drop, SR, kernel_size, radius2, vsize_z, raydist_mode_unit, knn_order, slope, cap_samples);
    outs.pop_back();
    return outs;
}

}  // namespace

TORCH_LIBRARY(hnr, m)
{
    m.def("grid_build(Tensor xyz, float[] origin, float[] cell, int[] dims, int[] query_size, int P, int max_o) -> int");
    m.def("grid_free(int grid) -> ()");
    m.def("march_query(int grid, Tensor campos, Tensor raydir, Tensor tmid, int SR, int K, float radius2, int[] kernel_size, bool pad, int knn_order) -> "
          "(Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("render_forward(int grid, Tensor xyz, Tensor conf, Tensor dir, Tensor color, Tensor point_table, Tensor? point_records, Tensor[] packed, Tensor campos, "
          "Tensor camrot, Tensor raydir, Tensor tmid, Tensor bg_color, Tensor? w2c, Tensor? intrinsic, Tensor? campos_nearest, Tensor? featmap, Tensor? frame_w, int SR, "
          "int K, int[] kernel_size, float radius2, float vsize_z, int raydist_mode_unit, int knn_order, float slope, int cap_samples) -> Tensor[]");
    m.def("render_train(int grid, Tensor?[] inputs, Tensor[] weights, Tensor? drop_lut, Tensor? ray_drop, int SR, int[] kernel_size, float radius2, float vsize_z, "
          "int raydist_mode_unit, int knn_order, float slope, int cap_samples) -> Tensor[]");
    m.def("render_train_fwd(int grid, Tensor?[] inputs, Tensor[] weights, Tensor? drop_lut, Tensor? ray_drop, int SR, int[] kernel_size, float radius2, float vsize_z, "
          "int raydist_mode_unit, int knn_order, float slope, int cap_samples) -> Tensor[]");
    m.def("render_train_bwd(Tensor?[] inputs, Tensor[] weights, Tensor[] fwd, Tensor g_raycolor, Tensor? g_conf_coefficient, int SR, int[] kernel_size, float radius2, "
          "float vsize_z, int raydist_mode_unit, int knn_order, float slope, int cap_samples) -> Tensor[]");
}
TORCH_LIBRARY_IMPL(hnr, CompositeExplicitAutograd, m)
{
    m.impl("grid_free", &grid_free);
    m.impl("render_train", &render_train_nograd);       // below autograd: the forward op alone (redispatches: CUDA kernel or shape function)
}
TORCH_LIBRARY_IMPL(hnr, CUDA, m)
{
    m.impl("grid_build", &grid_build);
    m.impl("march_query", &march_query);
    m.impl("render_forward", &render_forward);
    m.impl("render_train_fwd", &render_train_fwd);
    m.impl("render_train_bwd", &render_train_bwd);
}
TORCH_LIBRARY_IMPL(hnr, Autograd, m)
{
    m.impl("render_train", &render_train_autograd);
}
