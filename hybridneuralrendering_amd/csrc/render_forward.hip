// One call = one pass of NeuralPointsRayMarching.forward + fill_invalid over R rays
// (/root/reference/models/neural_points_volumetric_model.py:257-391, :87-126): query -> gather / aggregate -> composite.
//
// The reference synchronises with the host three times per 2304-ray chunk (query_point_indices_worldcoords.py:645-646, :705;
// point_aggregators.py:1092) because every stage sizes its tensors from a count the previous stage produced.  Here every launch is
// issued by this function, back to back on the caller's stream: the kernels read their work sizes from the query's device counters,
// the buffers live in ONE caller-provided workspace sized for a capacity of `cap_samples` valid shading samples, and an overflow of
// that capacity is reported in *d_status (device) instead of being discovered on the host.  No host read, no allocation.
#include <limits.h>

#include "hnr_common.h"

using namespace hnr;

namespace {

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

struct Layout {
    int32_t *work, *vs_item, *vs_off, *vs_cnt, *scratch, *row_s;
    void *chain_ws;
    float *X5, *sigma, *CF, *pre, *X6, *vmask, *M1, *X7, *Y1;
    size_t bytes;
};

Layout carve(void *ws, size_t ws_bytes, const hnr_render_params *p, bool *ok)
{
    Carver c{(char *)ws, 0, ws_bytes, true};
    Layout L;
    const size_t cap = (size_t)p->cap_samples, V = (size_t)p->V;
    L.work = c.take<int32_t>((size_t)hnr_query_work_elems(p->R, p->SR));
    L.vs_item = c.take<int32_t>(cap + 1); L.vs_off = c.take<int32_t>(cap + 1); L.vs_cnt = c.take<int32_t>(cap + 1);
    L.scratch = c.take<int32_t>(3 * (((size_t)p->R * p->SR + 1023) / 1024) + 3);
    L.chain_ws = c.take<char>((size_t)hnr_chain_workspace_bytes(p->cap_samples));
    L.X5 = c.take<float>(cap * 280); L.sigma = c.take<float>(cap + 1);
    L.CF = c.take<float>(cap * 128); L.pre = c.take<float>(cap * 64);
    // the per-(view, sample) rows exist only on the un-fused merge path (V != 4): hnr_merge_stage keeps them on chip (12 GB of the bench frame's
    // worst-case carve otherwise)
    const size_t VR = V == 4 ? 0 : V;
    L.X6 = c.take<float>(VR * cap * 48); L.vmask = c.take<float>(VR * cap + 1); L.row_s = c.take<int32_t>(VR * cap + 1);
    L.M1 = c.take<float>(VR * cap * 64);
    L.X7 = c.take<float>(cap * 92); L.Y1 = c.take<float>(cap * 48);
    L.bytes = (c.off + 255) & ~(size_t)255;
    if (ok) *ok = c.ok;
    return L;
}

__global__ void status_kernel(unsigned long long *counts, int cap_samples, int32_t *status)
{
    // status[0] = 1: more valid shading samples than the workspace capacity -- the extra ones are dropped (the frame is incomplete);
    // status[1] = the true count.  The counter the downstream kernels size themselves from is clamped to the capacity.
    const unsigned long long nv = counts[HNR_CNT_SAMPLES_VALID];
    status[1] = (int32_t)nv;
    status[0] = nv > (unsigned long long)cap_samples ? 1 : 0;
    if (nv > (unsigned long long)cap_samples) counts[HNR_CNT_SAMPLES_VALID] = (unsigned long long)cap_samples;
}

}  // namespace

extern "C" int64_t hnr_render_workspace_bytes(const hnr_render_params *p)
{
    if (!p || p->R <= 0 || p->SR <= 0 || p->K != 8 || p->V < 0 || p->cap_samples <= 0) return -1;
    return (int64_t)carve(nullptr, 0, p, nullptr).bytes + 256;
}

extern "C" int hnr_render_forward(const hnr_grid *grid, const hnr_render_params *p, const hnr_render_cloud *cl, const hnr_render_weights *w,
                                  const hnr_render_camera *cam, const hnr_render_views *vw, void *d_workspace, int64_t workspace_bytes,
                                  const hnr_render_outputs *o, void *stream)
{
    if (!grid || !p || !cl || !w || !cam || !o || (p->V > 0 && !vw)) { set_error("hnr_render_forward: NULL argument block"); return HNR_ERR_BADARG; }
    if (p->K != 8) { set_error("hnr_render_forward: built for K = 8 (got %d); drive the per-stage entry points for other K", p->K); return HNR_ERR_BADARG; }
    if (p->R <= 0 || p->SR <= 0 || p->cap_samples <= 0 || p->V < 0 || p->V > 8) { set_error("hnr_render_forward: bad sizes (R=%d SR=%d cap_samples=%d V=%d)", p->R, p->SR, p->cap_samples, p->V); return HNR_ERR_BADARG; }
    if (!d_workspace || ((uintptr_t)d_workspace & 255)) { set_error("hnr_render_forward: workspace must be 256-byte aligned"); return HNR_ERR_BADARG; }
    if (!o->d_raycolor || !o->d_opacity || !o->d_is_background || !o->d_ray_mask || !o->d_decoded || !o->d_sample_pidx || !o->d_sample_loc_w ||
        !o->d_ray_nsamp || !o->d_counts || !o->d_status) { set_error("hnr_render_forward: NULL output pointer"); return HNR_ERR_BADARG; }
    bool ok = true;
    const Layout L = carve(d_workspace, (size_t)workspace_bytes, p, &ok);
    if (!ok) { set_error("hnr_render_forward: workspace too small (%lld bytes, need %lld)", (long long)workspace_bytes, (long long)hnr_render_workspace_bytes(p)); return HNR_ERR_BADARG; }
    hipStream_t st = (hipStream_t)stream;
    const int R = p->R, SR = p->SR, K = p->K, cap = p->cap_samples, V = p->V;
    int rc, stage = 0;
    auto mark = [&]() -> int {
        if (o->stage_events && o->stage_events[stage]) { if (hipEventRecord((hipEvent_t)o->stage_events[stage], st) != hipSuccess) return 1; }
        ++stage;
        return 0;
    };
#define HNR_MARK() do { if (mark()) { set_error("hnr_render_forward: hipEventRecord failed"); return HNR_ERR_HIP; } } while (0)
    HNR_MARK();
    // ---- query (march + first-SR compaction + k-NN), un-padded outputs
    hnr_query_params q;
    q.R = R; q.D = p->D; q.SR = SR; q.K = K; q.radius2 = p->radius2; q.tmid_stride = p->tmid_stride; q.pad_outputs = 0; q.knn_order = p->knn_order;
    for (int i = 0; i < 3; ++i) q.kernel_size[i] = p->kernel_size[i];
    if ((rc = hnr_march_query(grid, cam->d_campos, cam->d_raydir, cam->d_tmid, &q, o->d_sample_pidx, o->d_sample_loc_w, o->d_ray_nsamp, o->d_ray_mask,
                              L.work, o->d_counts, stream)) != HNR_OK) return rc;
    HNR_MARK();
    // ---- plan: the list of valid samples, those with more than four neighbours first (8 row slots each), then the small ones (4 slots)
    if ((rc = hnr_chain_plan(L.work, o->d_sample_pidx, o->d_counts, K, R * SR, hnr_chain_classes(), L.vs_item, cap, L.scratch, stream)) != HNR_OK) return rc;
    status_kernel<<<1, 1, 0, st>>>(reinterpret_cast<unsigned long long *>(o->d_counts), cap, o->d_status);
    HNR_LAUNCH_CHECK();
    HNR_HIP_CHECK(hipMemsetAsync(o->d_decoded, 0, (size_t)R * SR * 4 * sizeof(float), st));
    HNR_MARK();
    // ---- per-neighbour chain
    rc = cl->d_rec ? hnr_chain_gather_rec(cl->d_rec, o->d_sample_pidx, o->d_sample_loc_w, cam->d_raydir, cam->d_campos, cam->d_camrot, L.vs_item, o->d_counts,
                                          SR, K, cap, L.chain_ws, L.X5, 280, o->d_weight, o->d_conf_coefficient, stream)
                   : hnr_chain_gather(cl->d_xyz, cl->d_conf, cl->d_dir, cl->d_color, o->d_sample_pidx, o->d_sample_loc_w, cam->d_raydir, cam->d_campos, cam->d_camrot,
                                      L.vs_item, o->d_counts, SR, K, cap, L.chain_ws, L.X5, 280, o->d_weight, o->d_conf_coefficient, stream);
    if (rc != HNR_OK) return rc;
    HNR_MARK();
    if ((rc = hnr_chain_forward(L.chain_ws, cl->d_point_table, cl->ldt, w->d_chain, o->d_counts, cap, w->slope, L.X5, 280, L.sigma, nullptr, 0, stream)) != HNR_OK) return rc;
    HNR_MARK();
    // ---- per-sample MLPs
    // colour feature 280 -> 128 -> 128 -> 128, and on its tail the colour-feature columns of aux_merge_weight_block.0 (128 -> 64, once per sample)
    const int cfN[4] = {128, 128, 128, 64}, cfK[4] = {280, 128, 128, 128}, act1110[4] = {1, 1, 1, 0}, act111[3] = {1, 1, 1};
    if ((rc = hnr_mlp3_forward(L.X5, 280, cap, o->d_counts, HNR_CNT_SAMPLES_VALID, 1, 0, w->d_mlp_cf, V > 0 ? 4 : 3, cfN, cfK, act1110, w->slope, nullptr, nullptr, 0,
                               L.CF, 128, L.pre, 64, stream)) != HNR_OK) return rc;
    HNR_MARK();
    if (V > 0 && vw->featmap_ready) HNR_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)vw->featmap_ready, 0));      // the feature map may have been built on another stream
    if (V == 4) {
        // reprojection + feature gather + merge-weight MLP + weighted merge in one launch: nothing per (view, sample) row reaches HBM
        HNR_MARK();
        if ((rc = hnr_merge_stage(o->d_sample_loc_w, L.vs_item, o->d_counts, vw->d_w2c, vw->d_intrinsic, cam->d_campos, vw->d_campos_nearest, vw->d_featmap, V,
                                  vw->H, vw->W, vw->d_frame_w, L.pre, 64, w->d_mlp_mw, w->d_mw_last_w, w->d_mw_last_b, L.CF, 128, cap, w->slope, L.X7, 92,
                                  stream)) != HNR_OK) return rc;
        HNR_MARK();
    } else if (V > 0) {
        if ((rc = hnr_proj_rows(o->d_sample_loc_w, L.vs_item, o->d_counts, vw->d_w2c, vw->d_intrinsic, cam->d_campos, vw->d_campos_nearest, vw->d_featmap,
                                V, vw->H, vw->W, L.CF, 128, cap, L.X6, 48, L.vmask, L.row_s, stream)) != HNR_OK) return rc;
        HNR_MARK();
        const int mwN[3] = {64, 64, 64}, mwK[3] = {48, 64, 64};
        if ((rc = hnr_mlp3_forward(L.X6, 48, (int64_t)V * cap, o->d_counts, HNR_CNT_SAMPLES_VALID, V, cap, w->d_mlp_mw, 3, mwN, mwK, act111, w->slope, L.pre, L.row_s, 64,
                                   L.M1, 64, nullptr, 0, stream)) != HNR_OK) return rc;
        HNR_MARK();
        if ((rc = hnr_merge(L.X6, 48, L.M1, 64, w->d_mw_last_w, w->d_mw_last_b, L.vmask, vw->d_frame_w, L.CF, 128, o->d_counts, V, cap, L.X7, 92,
                            nullptr, nullptr, 0, stream)) != HNR_OK) return rc;
    } else {
        // use_nearest = 0 (scene241.sh): the image branch is off, merged = 0 (point_aggregators.py:1257-1258)
        HNR_HIP_CHECK(hipMemsetAsync(L.X7, 0, (size_t)cap * 92 * sizeof(float), st));
        HNR_HIP_CHECK(hipMemcpy2DAsync(L.X7, 92 * sizeof(float), L.CF, 128 * sizeof(float), 45 * sizeof(float), cap, hipMemcpyDeviceToDevice, st));
        HNR_MARK(); HNR_MARK();
    }
    HNR_MARK();
    // color_mixup_block + residual + color_final_block + decode in one launch: the mix-up output never reaches HBM
    if ((rc = hnr_mixup_stage(L.X7, 92, w->d_mlp_mx, L.CF, 128, w->d_fin_w, w->d_fin_b, L.sigma, L.vs_item, o->d_counts, cap, w->slope, nullptr, 0,
                              o->d_decoded, stream)) != HNR_OK) return rc;
    HNR_MARK();
    HNR_MARK();
    // ---- composite + fill_invalid
    rc = hnr_composite(o->d_decoded, o->d_sample_loc_w, o->d_sample_pidx, o->d_ray_mask, o->d_ray_nsamp, cam->d_campos, cam->d_camrot, cam->d_bg_color,
                         R, SR, K, p->vsize_z, p->raydist_mode_unit, o->d_raycolor, o->d_opacity, o->d_is_background, o->d_blend_weight, stream);
    if (rc != HNR_OK) return rc;
    HNR_MARK();
#undef HNR_MARK
    return HNR_OK;
}
