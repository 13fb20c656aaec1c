/*
 * hnr.h -- C ABI of libhnr_hip.so: the MI355X-native replacement for the hot path of
 * CVMI-Lab/HybridNeuralRendering (per-ray voxel k-NN neural-point query -> point/image
 * feature gather + aggregation MLP -> front-to-back alpha composite).
 *
 * Conventions
 *   - every pointer whose name starts with d_ is a DEVICE pointer (HBM); everything else is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); no entry point synchronises
 *     the host unless its comment says so;
 *   - return value: 0 = HNR_OK, negative = error (never aborts, never throws);
 *   - all floating point is fp32, all indices int32, masks int8; the batch dimension B of the
 *     reference is always 1 and is dropped;
 *   - the caller owns every buffer; the only internal allocations are the index arrays owned by an
 *     hnr_grid handle and a small per-process scratch of counters;
 *   - ONE PROCESS PER GPU is the supported deployment (the multi-GPU path is one rank per GPU over RCCL): kernel attributes (large dynamic
 *     LDS) and the CU count are cached per device, but the training calls' side streams and the small per-process scratch belong to the
 *     device that was current when they were first used.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference repo).
 *
 * STABILITY.  Two tiers:
 *   STABLE  -- what a reference-side binding needs and what later versions keep source- and binary-compatible: hnr_version,
 *              hnr_last_error, hnr_points_bounds, hnr_grid_* (build / destroy / stats / bytes), hnr_march_query, hnr_ray_compact*, hnr_point_records,
 *              hnr_image_features*, hnr_render_forward* (+ workspace sizing), hnr_render_train_forward / _backward (+ sizing), hnr_shipped_loss*,
 *              hnr_composite, hnr_ray_march, hnr_voxel_downsample*, hnr_probe_select, hnr_blur_*.
 *   STAGE   -- everything else (hnr_chain_*, hnr_mlp3_*, hnr_merge*, hnr_mixup_stage, hnr_proj_*, hnr_h2*, hnr_linear_*, hnr_gather_*, hnr_ksum*,
 *              hnr_segment_*, hnr_absmax, hnr_div_probe, ...): the individual stages the two single-call entries are built from.  They are exported so
 *              that tests/ can compare every stage with the oracle and so that tools/ can time them alone; their signatures, workspace layouts and
 *              packed-image formats follow the kernels and MAY CHANGE from one version to the next (round 4 changed what hnr_chain_forward leaves
 *              in the workspace's row scalars, for example).  Do not bind to them from outside this repository.
 */
#ifndef HNR_H
#define HNR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNR_OK            0
#define HNR_ERR_BADARG   -1   /* NULL pointer, non-positive size, unsupported K/SR ...          */
#define HNR_ERR_HIP      -2   /* a HIP runtime call failed (hnr_last_error() has the string)     */
#define HNR_ERR_TOOBIG   -3   /* grid volume or an index would overflow 32 bits                 */
#define HNR_ERR_NOMEM    -4   /* device allocation failed                                       */
#define HNR_NEED_REBUILD  1   /* hnr_grid_grow: not an error -- the update cannot be done in place, nothing was changed; call hnr_grid_build */

#define HNR_MAX_K        32   /* neighbours per shading sample (reference scripts: 8)           */

/* Library / build identification: "hnr-hip <version> gfx950".  */
const char *hnr_version(void);
/* Text of the last error on this thread ("" if none). */
const char *hnr_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Stage 0: bounds of the point cloud.
 * Replaces the torch.min/torch.max pair of lighting_fast_querier.get_hyperparameters
 * (models/neural_points/query_point_indices_worldcoords.py:56).
 *   d_xyz [n,3] -> d_out6 = {min_x,min_y,min_z,max_x,max_y,max_z}
 */
int hnr_points_bounds(const float *d_xyz, int n, float *d_out6, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 1: voxel grid over the point cloud.  Built ONCE per point-cloud version and reused by
 * every query (the reference rebuilds it for every 2304-ray chunk).
 * Replaces build_occ_vox = claim_occ + map_coor2occ + fill_occ2pnts
 * (query_point_indices_worldcoords.py:540-602, kernels :237-381).
 */
typedef struct hnr_grid hnr_grid;

typedef struct {
    float origin[3];      /* d_coord_shift  = ranges_np[:3]        (:67, :611)                  */
    float cell[3];        /* d_voxel_size   = scaled_vsize_np      (:58)                         */
    int   dims[3];        /* d_grid_size    = scaled_vdim_np       (:71)                         */
    int   query_size[3];  /* occupancy dilation, opt.query_size     (:616 passes query_size_gpu) */
    int   P;              /* max points listed per voxel            (opt.P)                      */
    int   max_o;          /* max occupied voxels                    (opt.max_o)                  */
} hnr_grid_params;

typedef struct {
    int64_t n_points;        /* points handed in                                               */
    int64_t n_inbounds;      /* points whose voxel lies inside dims                            */
    int64_t n_occ;           /* occupied voxels kept (<= max_o)                                */
    int64_t n_dropped_voxels;/* voxels beyond max_o (reference: random replacement)            */
    int64_t n_cells_over_P;  /* voxels holding more than P points (lists truncated to first P) */
    int64_t n_dilated;       /* cells set in the dilated march mask                            */
    int64_t n_words;         /* 4x4x4 bricks (64-bit words) covering dims                      */
    int64_t bytes;           /* HBM bytes owned by the handle                                  */
} hnr_grid_stats;

/* Synchronises the host once (it must size the index arrays). */
int hnr_grid_build(const float *d_xyz, int n_points, const hnr_grid_params *p, void *stream, hnr_grid **out);
int hnr_grid_free(hnr_grid *g);
int hnr_grid_get_stats(const hnr_grid *g, hnr_grid_stats *out);
int hnr_grid_get_params(const hnr_grid *g, hnr_grid_params *out);

/* After grow_points (models/neural_points/neural_points.py:376-402: new points are APPENDED to the cloud): extends the tables of a grid in place instead of
 * rebuilding them.  d_xyz [n_points,3] is the grown cloud; its first hnr_grid_get_stats().n_points rows must be the points the grid describes, unchanged,
 * and the grid parameters (origin / cell / dims, i.e. the cloud's bounding box and opt) must still apply -- the caller checks both
 * (querier.lighting_fast_querier.grow).  On HNR_OK the grid is logically what hnr_grid_build(d_xyz, n_points, same parameters) returns: same dilated
 * mask, same per-cell lists and 3x3x3 neighbourhood runs in the same order, same counters; physically the changed lists / runs were appended in the
 * slack the build leaves behind its tables (HNR_GRID_SLACK percent, default 25) and the superseded ones stay as holes.  Returns HNR_NEED_REBUILD (> 0,
 * nothing changed) when that slack is used up, when max_o would be exceeded or was, or for grids without neighbourhood lists (P > 63).  Synchronises the
 * host twice (sizes); launches already queued keep reading the old tables.  The reference has no counterpart: it rebuilds its tables for every
 * 2304-ray chunk and leaves the process after growing (run/train_ft.py:926-952). */
int hnr_grid_grow(hnr_grid *g, const float *d_xyz, int n_points, void *stream);
/* Test hook: per cell of dims, the length of its 3x3x3 neighbourhood run (-1: the cell is not in the dilated mask) and an order-dependent hash of the
 * run's records -- two grids with equal hnr_grid_export_dense and hnr_grid_export_runs outputs answer every query identically. */
int hnr_grid_export_runs(const hnr_grid *g, int32_t *d_run_len, uint64_t *d_run_hash, void *stream);

/* Test hook: expands the device tables into the reference's dense layout so they can be compared
 * with the oracle: d_coor_occ [X*Y*Z] u8 (dilated mask), d_cell_count [X*Y*Z] i32 (-1 = voxel not
 * occupied, else min(P, points listed)), d_cell_first [X*Y*Z] i32 (first listed point id or -1). */
int hnr_grid_export_dense(const hnr_grid *g, uint8_t *d_coor_occ, int32_t *d_cell_count,
                          int32_t *d_cell_first, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 2: ray march + first-SR compaction + layered k-NN.
 * Replaces mask_raypos (:384-408) + the torch cumsum compaction (:645-655) + get_shadingloc
 * (:411-433) + query_neigh_along_ray_layered (:436-522), and fuses away the materialised
 * raypos tensor of near_far_linear_ray_generation (models/rendering/diff_ray_marching.py:386).
 */
typedef struct {
    int   R;              /* rays in this launch                                                */
    int   D;              /* marched samples per ray (opt.z_depth_dim = 400)                    */
    int   SR;             /* shading samples kept per ray (opt.SR)                              */
    int   K;              /* neighbours per shading sample (opt.K), <= HNR_MAX_K                */
    int   kernel_size[3]; /* k-NN neighbourhood (opt.kernel_size); layers = (kernel_size[0]+1)/2 */
    float radius2;        /* radius_limit^2, 0 = unlimited (:493, :685)                         */
    int   tmid_stride;    /* 0: d_tmid is one [D] table shared by all rays (jitter = 0);
                             D: d_tmid is [R,D], one depth table per ray (train-time jitter)    */
    int   pad_outputs;    /* 1: d_sample_pidx / d_sample_loc_w are fully written, -1 / 0 in the unused slots
                             (the reference's torch.full / torch.zeros tensors, :647-648);
                             0: only the first d_ray_nsamp[r] slots of a ray are written (fused path: consumers
                             take d_ray_nsamp; saves 1 kB of padding stores per ray)                */
    int   knn_order;      /* 0: the K slots of a sample are filled exactly as the reference kernel fills them (insertion
                             history, farthest-first replacement: query_point_indices_worldcoords.py:494-513);
                             1: the same neighbour SET in canonical order, ascending (d^2, enumeration order) -- a sorted
                             insertion instead of the replay of that rule (K = 8, 3x3x3 neighbourhood); every consumer of
                             the path sums over the K slots, so only the fp32 summation order of a sample changes.
                             Caveat: with EXACT d^2 ties at the current maximum of a full list the retained point can differ
                             (the reference evicts the first maximum in slot order, the sorted list the latest-enumerated
                             of the tied entries): use 0 for golden / PSNR comparisons on clouds with duplicate distances */
} hnr_query_params;

/* counters written by hnr_march_query (device, int64[HNR_NCOUNTS]) */
enum {
    HNR_CNT_RAYS_HIT = 0,     /* rays with >= 1 occupied marched sample (:645-646)              */
    HNR_CNT_SAMPLES,          /* shading samples kept over all rays                             */
    HNR_CNT_RAYS_VALID,       /* rays with >= 1 neighbour (:705-706), filled by hnr_ray_compact_plan */
    HNR_CNT_NEIGHBOURS,       /* sample_pidx entries >= 0                                       */
    HNR_CNT_CELLS_VISITED,    /* occupied cells whose lists were scanned by the k-NN            */
    HNR_CNT_CANDIDATES,       /* points distance-tested by the k-NN                             */
    HNR_CNT_SAMPLES_VALID,    /* shading samples with >= 1 neighbour                            */
    HNR_CNT_SAMPLES_SMALL,    /* written by hnr_chain_plan: valid samples of its second class (4 row slots: at most 4 neighbours
                                 with classes = 1, 3..4 with classes = 2), listed AFTER the first class in its d_vs_item (0 after
                                 hnr_march_query: every sample counts as a full one)                                      */
    HNR_CNT_SAMPLES_TINY,     /* written by hnr_chain_plan(classes = 2): valid samples with 1..2 neighbours (2 row slots), listed last */
    HNR_NCOUNTS = 9
};

/*
 * Outputs are in the UN-COMPACTED ray order (row r = input ray r):
 *   d_sample_pidx  [R,SR,K] i32, -1 padded         (reference: sample_pidx before :708)
 *   d_sample_loc_w [R,SR,3] f32, 0 padded          (reference: sample_loc before :709)
 *   d_ray_nsamp    [R]      i32  shading samples kept on the ray
 *   d_ray_mask     [R]      i8   1 iff the ray has >= 1 neighbour (reference: ray_mask, :707,:711)
 *   d_work         i32[hnr_query_work_elems(R,SR)]  scratch; its first counts[HNR_CNT_SAMPLES] entries are the
 *                  packed (ray*SR+slot) list of kept samples in (ray, slot) order
 *   d_counts       [HNR_NCOUNTS] i64
 * d_campos [3], d_raydir [R,3], d_tmid see tmid_stride.  No host synchronisation.
 */
int64_t hnr_query_work_elems(int R, int SR);
int hnr_march_query(const hnr_grid *g, const float *d_campos, const float *d_raydir, const float *d_tmid,
                    const hnr_query_params *q,
                    int32_t *d_sample_pidx, float *d_sample_loc_w, int32_t *d_ray_nsamp, int8_t *d_ray_mask,
                    int32_t *d_work, int64_t *d_counts, void *stream);

/*
 * Second compaction of the reference (:705-709), the ray-direction expansion of query_points
 * (:91) and the camera-perspective sample coordinates (x/z, y/z, z) of lighting_fast_querier.w2pers
 * (:96-103): rows of rays with ray_mask = 1, in ray order.
 *   hnr_ray_compact_plan : d_ray_row [R] i32 = compact row of ray r or -1; d_counts[HNR_CNT_RAYS_VALID].
 *   hnr_ray_compact      : d_out_* sized for n_valid rows (read d_counts on the host in between).
 * d_scratch: int32[(R+1023)/1024 + 1].
 */
int hnr_ray_compact_plan(const int8_t *d_ray_mask, int R, int32_t *d_ray_row, int32_t *d_scratch,
                         int64_t *d_counts, void *stream);
int hnr_ray_compact(const int32_t *d_ray_row, int R, int SR, int K,
                    const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir,
                    const float *d_campos /*[3]*/, const float *d_camrotc2w /*[3,3] row-major*/,
                    int32_t *d_out_pidx /*[R',SR,K]*/, float *d_out_loc_w /*[R',SR,3]*/,
                    float *d_out_loc_pers /*[R',SR,3] = w2pers(loc_w), :96-103*/,
                    float *d_out_raydir /*[R',SR,3]*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 3a: dense layers of the aggregation MLP on the matrix cores (fp32 in, fp32 accumulate).
 * Replaces the nn.Linear(+LeakyReLU) layers of PointAggregator.viewmlp
 * (models/aggregators/point_aggregators.py:948 block1, :972 block3, :1037 color_feature_branch,
 *  :1199 aux_merge_weight_block, :1292 color_mixup_block).
 *   C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]),  act: 0 = none, 1 = LeakyReLU(slope)
 * W is the torch nn.Linear weight ([out,in], row-major).  It is packed ONCE per checkpoint into a
 * zero-padded [N_pad,K_pad] image (+ bias[N_pad]) by hnr_linear_pack; sizes from hnr_linear_packed_dims.
 * lda must be a multiple of 4 floats and A 16-byte aligned; ldc >= N.
 */
int hnr_linear_packed_dims(int N, int K, int *N_pad, int *K_pad);
int hnr_linear_pack(const float *d_W, const float *d_bias /*may be NULL*/, int N, int K,
                    float *d_Wp /*[N_pad*K_pad]*/, float *d_bias_p /*[N_pad]*/, void *stream);
int hnr_linear_f32(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, float *d_C, int ldc,
                   int M, int N, int K, int act, float slope, void *stream);
/* Same, with a gathered per-row addend:  C[m,:] = act(A[m,:] W^T + bias + R[ridx[m], :])  (R row stride ldr >= N).
 * Used to split block1.0 (point_aggregators.py:948): its input row is [emb | PE(emb) | PE(dists)] (:931-939) and the first
 * 224 columns depend on the POINT only, so R = [emb | PE(emb)] W[:, :224]^T is computed once per point (hnr_point_rows +
 * hnr_linear_f32) and each (sample, neighbour) row only multiplies its 60 distance-encoding columns. */
int hnr_linear_f32_gather_add(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, const float *d_R,
                              const int32_t *d_ridx, int ldr, float *d_C, int ldc, int M, int N, int K, int act, float slope,
                              void *stream);
/* General form of the side operand, applied to the output columns < r_cols only (d_ridx may be NULL = row m itself):
 *   r_mode 0: C = act(A W^T + bias + R[ridx[m], :])          (d_R == d_C gives an accumulating "+=" layer)
 *   r_mode 1: C = (A W^T + bias) * (R[m, :] > 0 ? 1 : slope)  LeakyReLU derivative of the stored activation R; act must be 0.
 * r_mode 1 is the input-gradient GEMM of the backward pass (torch autograd of nn.Linear + LeakyReLU in the reference):
 * with W^T packed as the weight, dZ_prev = (dZ W) * LeakyReLU'(Y_prev). */
int hnr_linear_f32_side(const float *d_A, int lda, const float *d_Wp, const float *d_bias_p, const float *d_R,
                        const int32_t *d_ridx, int ldr, int r_cols, int r_mode, float *d_C, int ldc, int M, int N, int K,
                        int act, float slope, void *stream);
/* ------------------------------------------------------------------------------------------------
 * Stage 3b: everything of the gather / aggregate / composite path that is not a dense layer.
 * Row order is the reference's boolean-mask order, (ray, slot, k) ascending, so packed rows line up
 * with `feat[pnt_mask_flat]` / `[ray_valid]` of point_aggregators.py:924-935, :1014-1026.
 * Every function reads its work sizes from d_counts on the device (no host sync); `cap_*` arguments
 * are the capacities of the caller's buffers (grid sizes); hnr_sample_plan raises *d_overflow when a
 * capacity is too small.
 */

/* Compact lists over the kept samples of hnr_march_query (d_work, d_counts[HNR_CNT_SAMPLES]):
 *   d_vs_item[s] = ray*SR+slot of the s-th VALID sample (>= 1 neighbour; `ray_valid`, :1441),
 *   d_vs_off[s]  = index of its first neighbour row, d_vs_cnt[s] = its neighbour count.
 * d_scratch: int32[2*ceil(max_items/1024)].  max_items = R*SR. */
int hnr_sample_plan(const int32_t *d_work, const int32_t *d_sample_pidx, const int64_t *d_counts, int K, int max_items,
                    int32_t *d_vs_item, int32_t *d_vs_off, int32_t *d_vs_cnt, int cap_samples, int cap_rows,
                    int32_t *d_scratch, int32_t *d_overflow, void *stream);

/* NeuralPoints gather (neural_points.py:709-720) + w2pers (:607-613) + dists (point_aggregators.py:1472-1480)
 * + inverse-distance weights (:825-833, :1500-1501, x clamp(conf) :1508-1512) + positional encodings
 * (:930, :938) written straight into the packed MLP inputs:
 *   d_X1[row, 0:284]     = [emb32 | PE3(emb) 192 | PE5(dists6) 60]            (block1 input)
 *   d_X3[row, 256:263]   = [color3 | dir - viewdir | dir . viewdir]            (block3 extras, :957-971)
 *   d_wagg[row]          = normalised weight * clamp(conf, 1e-4, 1)
 *   optional d_weight_out / d_conf_out [R,SR,K]: the reference's `weight` and `conf_coefficient` outputs.
 * Point buffers: xyz [N,3], emb [N,32], conf [N], dir [N,3], color [N,3].
 * d_row_pid != NULL selects the SPLIT layout: d_X1[row, 0:60] = PE5(dists6) only (ld1 >= 60) and d_row_pid[row] = point id;
 * the point-only columns [emb32 | PE3(emb)] come from hnr_point_rows through hnr_linear_f32_gather_add. */
int hnr_gather_rows(const float *d_xyz, const float *d_emb, const float *d_conf, const float *d_dir, const float *d_color,
                    int F, const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir,
                    const float *d_campos, const float *d_camrot, const int32_t *d_vs_item, const int32_t *d_vs_off,
                    const int32_t *d_vs_cnt, const int64_t *d_counts, int SR, int K, int cap_samples,
                    float *d_X1, int ld1, float *d_X3, int ld3, float *d_wagg, float *d_weight_out, float *d_conf_out,
                    int32_t *d_row_pid, void *stream);

/* d_E[p, 0:224] = [emb32 | PE3(emb) 192] for every point (point_aggregators.py:931-938): the point-only columns of
 * block1's input row.  lde >= 224, multiple of 4.  With d_ids the rows are those of the listed points only (training:
 * the points a batch touches, hnr_unique_points). */
int hnr_point_rows(const float *d_emb, const int32_t *d_ids /*NULL or [n_points] point ids*/, int n_points, int F, float *d_E, int lde,
                   void *stream);

/* The materialised gather of NeuralPoints.forward (neural_points.py:709-720) for the drop-in 14-tuple only:
 * n_entries = R'*SR*K entries of d_sample_pidx; empty entries (-1) read point 0 (the reference clamps the index);
 * outputs [n,3] [n,3] [n] [n,F] [n,3] [n,3] and d_o_mask [n] u8 = (pidx >= 0). */
int hnr_gather_points(const int32_t *d_sample_pidx, int64_t n_entries, const float *d_xyz, const float *d_emb,
                      const float *d_conf, const float *d_dir, const float *d_color, int F, const float *d_campos,
                      const float *d_camrot, float *d_o_color, float *d_o_dir, float *d_o_conf, float *d_o_emb,
                      float *d_o_xyz_pers, float *d_o_xyz, uint8_t *d_o_mask, void *stream);

/* alpha branch + softplus(x-1) (:1005, :471-476) + K-weighted sums (:1008-1026) + view-direction encoding:
 *   d_X5[s, 0:280] = [sum_k w feat_k (256) | sin(viewdir 2^f) 12 | cos 12]   (color_feature_branch input, :1028-1036)
 *   d_sigma[s]     = sum_k w softplus(alpha_k - 1) */
int hnr_ksum(const float *d_H4, int ldh, const float *d_wagg, const float *d_alpha_w, const float *d_alpha_b,
             const int32_t *d_vs_item, const int32_t *d_vs_off, const int32_t *d_vs_cnt, const float *d_raydir,
             const int64_t *d_counts, int SR, int cap_samples, float *d_X5, int ld5, float *d_sigma, void *stream);

/* Reference-view feature pyramid, once per frame (:1047-1067, :1089): conv_w/conv_b are HOST arrays of 6 device
 * pointers (aux_block_s1.{0,2}, s2.{0,2}, s3.{0,2}); d_img [V,H,W,3]; d_featmap [V,H,W,48] channels-last
 * (45 used: RGB | up(s1) 6 | up(s2) 12 | up(s3) 24), pixel (0,0) zeroed.
 * d_scratch: float[hnr_image_features_scratch_elems(V,H,W)]. */
int64_t hnr_image_features_scratch_elems(int V, int H, int W);
int hnr_image_features(const float *d_img, int V, int H, int W, const float *const *conv_w, const float *const *conv_b,
                       float slope, float *d_scratch, float *d_featmap, void *stream);

/* Merge-weight rows: reprojection into the V reference views (neural_points_volumetric_model.py:248-255,
 * d_w2c[v] = inverse(c2w_nearest[v]) row-major), truncation to a pixel + bounds rule (:1077-1088), feature
 * gather, delta view directions (:296-310):  d_X6[v*cap+s, 0:176] = [imgfeat45 | colfeat128 | ddir3], d_vmask.
 * d_row_sample != NULL selects the SPLIT layout: rows are [imgfeat45 | ddir3] (ld6 >= 48), d_row_sample[row] = s and the
 * colour feature is not copied -- its 128 columns of aux_merge_weight_block.0 are shared by the V views of a sample, so they
 * are multiplied once per sample and added per row by hnr_linear_f32_gather_add. */
int hnr_proj_rows(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                  const float *d_intrinsic, const float *d_campos, const float *d_campos_nearest, const float *d_featmap,
                  int V, int H, int W, const float *d_CF, int ldcf, int cap_samples, float *d_X6, int ld6, float *d_vmask,
                  int32_t *d_row_sample, void *stream);

/* Probe (parity evidence): q_hnr[i] = the library's divide-free correctly rounded quotient (hnr_div, csrc/hnr_common.h: what chain_gather_kernel and
 * train_ksum_bwd_kernel divide with instead of the v_div_scale / v_div_fmas expansion), q_ieee[i] = num[i] / den[i] as the compiler expands it. */
int hnr_div_probe(const float *d_num, const float *d_den, int n, float *d_q_hnr, float *d_q_ieee, void *stream);
/* ... and for the other two divisions of the device code: cell_hnr[i] = the query kernels' cell index floor((p[i] - origin) / c[i]) (hnr_div_cell inside
 * cell_coord; INT32_MIN beyond +-2e9 / NaN), cell_ieee[i] = the same expression with the compiler's division; q64_* = hnr_div64 (the loss kernels' means)
 * beside the compiler's fp64 division on operands derived from p, c. */
int hnr_div_probe2(const float *d_p, const float *d_c, float origin, int n, int32_t *d_cell_hnr, int32_t *d_cell_ieee, double *d_q64_hnr, double *d_q64_ieee,
                   void *stream);

/* Probe (parity evidence, not on the render path): the integer pixel every (view v, valid sample s) row of the merge stage gathers,
 * d_pix[(v * cap_samples + s) * 2 + {0,1}] = (px, py), or (-1, -1) where the reference's bounds rule masks the row -- computed by the one
 * device function (hnr_project_pixel, csrc/hnr_common.h) that hnr_proj_rows, hnr_merge_stage and hnr_proj_rows_bwd call, so that a test or
 * bench.py can tell a sample that truncates to another pixel than the CPU oracle's projection (point_aggregators.py:1077-1078: `.to(torch.int32)`
 * of a coordinate within an ulp of a pixel border) from an arithmetic difference. */
int hnr_proj_pixels(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                    const float *d_intrinsic, int V, int H, int W, int cap_samples, int32_t *d_pix, void *stream);

/* Last layer + sigmoid of aux_merge_weight_block, weighted merge (:1199-1217) and the mix-up input (:1286-1292):
 *   d_X7[s, 0:90] = [colfeat[:45] | sum_v w_v f_v / (sum_v w_v + 1e-6)].  d_frame_w: optional [V].
 * d_ray_drop: optional [R] u8; samples on flagged rays get merged = 0 (train-time patch drop, :1222-1237; the caller
 * builds the flags from drop_patch_rays :14-23 over the valid-ray rows). */
int hnr_merge(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last,
              const float *d_vmask, const float *d_frame_w, const float *d_CF, int ldcf, const int64_t *d_counts, int V,
              int cap_samples, float *d_X7, int ld7, const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The per-neighbour chain of PointAggregator.viewmlp fused into one kernel (csrc/chain.hip): block1 (:948), block3 on
 * [block1_out | colour | dir - viewdir | dir.viewdir] (:957-972), alpha_branch + softplus(x - 1) (:1005, :471-476) and the
 * K-weighted sums per shading sample (:1008-1026).  Replaces hnr_gather_rows + four hnr_linear_* launches + hnr_ksum of the
 * per-layer path for K = 8: activations stay on chip, only [S_v, 256] feature sums and [S_v] densities are written.
 * fp32 in / fp32 out; dense arithmetic: every operand split into two fp16 terms (22 bits), three 16-bit MFMAs per product,
 * fp32 accumulation, exact power-of-two scaling per activation row / per layer (fp32-class error, tests/test_chain_gpu.py).
 *
 * hnr_chain_pack: the four nn.Linear weights -> the kernel's image, once per checkpoint.  d_w_b1_0_dist = block1.0.weight[:, 224:284]
 *   (row stride ldw0; the point-only columns [:, :224] go into the per-point table, hnr_point_rows + hnr_linear_f32), d_w_b1_2
 *   [256,256], d_w_b3_0 [256,263], d_w_b3_2 [256,256] contiguous, biases [256], alpha_branch.0 weight [256] / bias [1].
 * hnr_chain_gather: gather + geometry + positional encoding of the distances for the valid samples d_vs_item[0 .. n_valid)
 *   (n_valid = d_counts[HNR_CNT_SAMPLES_VALID], read on the device; cap_samples bounds it) into d_workspace
 *   (hnr_chain_workspace_bytes(cap_samples)); also d_X5[s, 256:280] = view-direction encoding (:909-913) and, optionally, the
 *   reference's `weight` / `conf_coefficient` outputs [R,SR,K] (written for valid neighbour slots only).
 * hnr_chain_plan: the list of valid samples the chain works on (the packing by `pnt_mask_flat` / `sampled_Rw2c` masks of
 *   models/aggregators/point_aggregators.py:924,935,961-970), d_vs_item[s] = ray * SR + slot, from the kept-sample work list of
 *   hnr_march_query.  classes = 0: in (ray, slot) order, exactly hnr_sample_plan's list.  classes = 1: the samples with more than four
 *   neighbours first, then those with 1..4 (each class in (ray, slot) order), and d_counts[HNR_CNT_SAMPLES_SMALL] = size of the second
 *   class: the gather and the chain kernel give a small sample 4 row slots instead of 8 (82 % of the bench frame's samples have 8
 *   neighbours, 11 % at most 4: 5.5 % fewer rows through the four dense layers; the sums are bit-identical, a slot without a neighbour
 *   contributes an exact zero).  classes = 2: three classes -- more than 4 neighbours (8 slots), 3..4 (4 slots,
 *   HNR_CNT_SAMPLES_SMALL), 1..2 (2 slots, HNR_CNT_SAMPLES_TINY; 7.5 % of the bench frame's samples): another 2 % fewer rows.
 *   Everything downstream is per sample and order-free.  hnr_chain_classes() = the largest `classes` the chain kernel selected in
 *   this process supports (0 for the older kernel variants).  d_scratch: int32[3 * ceil(max_items / 1024) + 3].
 * hnr_chain_forward: d_X5[s, 0:256] = sum_k w_k block3(...)_k, d_sigma[s] = sum_k w_k softplus(alpha_k - 1).
 *   d_point_table [N, ldt >= 256] = [emb | PE3(emb)] block1.0.weight[:, :224]^T.  d_dbg (probe, may be NULL): the post-activation
 *   output of layer dbg_layer (0..3) as [rows = 8 per valid sample, 256]. */
int64_t hnr_chain_packed_bytes(void);
int64_t hnr_chain_workspace_bytes(int cap_samples);
int hnr_chain_classes(void);
int hnr_chain_plan(const int32_t *d_work, const int32_t *d_sample_pidx, int64_t *d_counts, int K, int max_items, int classes,
                   int32_t *d_vs_item, int cap_samples, int32_t *d_scratch, void *stream);
int hnr_chain_pack(const float *d_w_b1_0_dist, int ldw0, const float *d_b_b1_0, const float *d_w_b1_2, const float *d_b_b1_2,
                   const float *d_w_b3_0, const float *d_b_b3_0, const float *d_w_b3_2, const float *d_b_b3_2,
                   const float *d_alpha_w, const float *d_alpha_b, void *d_packed, void *stream);
int hnr_chain_gather(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color,
                     const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir, const float *d_campos,
                     const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR, int K, int cap_samples,
                     void *d_workspace, float *d_X5, int ld5, float *d_weight_out, float *d_conf_out, void *stream);
/* hnr_point_records: the four per-point buffers the gather reads (NeuralPoints' xyz / points_conf / points_dir / points_color, indexed by
 *   sample_pidx in models/neural_points/neural_points.py:709-720), interleaved once per cloud version into 48-byte records
 *   d_rec [N][12] f32 = {x y z conf | dir.x dir.y dir.z r | g b 0 0}; hnr_chain_gather_rec = hnr_chain_gather reading them (the same values,
 *   bit-identical outputs): one or two 64-byte sectors per neighbour instead of four scattered reads (6.0 -> 1.8 GB fetched per frame,
 *   profiles/r02_traffic.json). */
int hnr_point_records(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color, int N, float *d_rec, void *stream);
int hnr_chain_gather_rec(const float *d_rec, const int32_t *d_sample_pidx, const float *d_sample_loc_w, const float *d_raydir,
                         const float *d_campos, const float *d_camrot, const int32_t *d_vs_item, const int64_t *d_counts, int SR, int K,
                         int cap_samples, void *d_workspace, float *d_X5, int ld5, float *d_weight_out, float *d_conf_out, void *stream);
int hnr_chain_forward(const void *d_workspace, const float *d_point_table, int ldt, const void *d_packed, const int64_t *d_counts,
                      int cap_samples, float slope, float *d_X5, int ld5, float *d_sigma, float *d_dbg, int dbg_layer, void *stream);

/* Three consecutive nn.Linear (+ LeakyReLU) layers of width <= 128 fused into one kernel (csrc/mlp.hip), fp32 in / fp32 out, the
 * same two-term fp16 split arithmetic as the chain above: the per-sample MLPs of viewmlp -- color_feature_branch (:1028-1037),
 * the first three layers of aux_merge_weight_block (:1199) and color_mixup_block (:1285-1292).
 *   hnr_mlp3_pack: d_W[l] [N[l], K[l]] (row stride ldw[l]), d_bias[l] [N[l]] or NULL; K[l] = N[l-1]; K[0] <= 288, N <= 128.
 *   hnr_mlp3_forward: d_C[m, 0:N[2]] = L2(L1(L0(d_A[m, 0:K[0]]) )), act[l] != 0 applies LeakyReLU(slope) after layer l; optional addend of
 *   layer 0 before its activation: d_R[d_ridx[m], 0:N[0]] (the colour-feature part of aux_merge_weight_block.0, shared by a sample's
 *   views).  Rows: M = min(M_cap, d_counts[count_index] * count_mult) when d_counts != NULL (device-side count), else M_cap.
 *   seg_stride > 0: the rows are count_mult SEGMENTS of d_counts[count_index] rows each, segment v starting at row v * seg_stride
 *   (the (view, sample) rows of hnr_proj_rows: row = v * cap_samples + s).
 *   n_layers = 4 adds a TAIL: a fourth layer reading layer 2's output, written to d_C2[m, 0:N[3]] (K[3] = N[2]) -- the colour-feature
 *   columns of aux_merge_weight_block.0 (128 -> 64, no activation) ride on the colour-feature launch.
 *   Built for the k-step tuples of these uses: K = (280,128,128[,128]), (48,64,64), (90,45,45). */
int64_t hnr_mlp3_packed_bytes(int n_layers, const int *K);
int hnr_mlp3_pack(int n_layers, const float *const *d_W, const int *ldw, const int *N, const int *K, const float *const *d_bias, void *d_packed,
                  void *stream);
int hnr_mlp3_forward(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                     const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                     const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, void *stream);

/* The merge stage of the image branch in ONE launch (V = 4): reprojection of every valid sample into the reference views + truncation +
 * feature gather + delta view directions (hnr_proj_rows: neural_points_volumetric_model.py:248-255, :296-310; point_aggregators.py:1077-1088),
 * aux_merge_weight_block on [imgfeat45 | colfeat128 | ddir3] (:1199; the colour-feature columns enter as the per-sample addend d_pre [S, 64],
 * bias included -- the tail output of the colour-feature launch), sigmoid, validity / frame weights, the weighted merge over the views
 * (:1217) and the mix-up row d_X7[s, 0:90] = [colfeat[:45] | merged45] (:1286-1292).  Replaces hnr_proj_rows + hnr_mlp3_forward + hnr_merge:
 * the [4 S, 48] rows, the [4 S, 64] activations and the per-row weights stay on chip.  d_mlp_mw: hnr_mlp3_pack image of the three layers
 * 48 -> 64 -> 64 -> 64 (first layer = the image-feature and direction columns, no bias). */
int hnr_merge_stage(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c, const float *d_intrinsic,
                    const float *d_campos, const float *d_campos_nearest, const float *d_featmap, int V, int H, int W, const float *d_frame_w,
                    const float *d_pre, int ldpre, const void *d_mlp_mw, const float *d_w_last, const float *d_b_last, const float *d_CF, int ldcf,
                    int cap_samples, float slope, float *d_X7, int ld7, void *stream);

/* Mix-up stage in one launch: color_mixup_block (90 -> 45 -> 45 -> 45, :1285-1292; d_mlp_mx = its hnr_mlp_pack image), learn_residuals (:1294),
 * color_final_block + sigmoid*1.002-0.001 (:1295, :1334, :478-482), scattered with sigma into d_decoded [R*SR,4] (:1337-1338).  Equals
 * hnr_mlp3_forward + hnr_final_color bit for bit.  d_X7 [S, ld7 >= 92] (ld7 a multiple of 4, 16-B aligned), d_CF [S, ldcf >= 128];
 * d_Y (optional) receives the mix-up output rows [S, ldy >= 48]. */
int hnr_mixup_stage(const float *d_X7, int ld7, const void *d_mlp_mx, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                    const float *d_sigma, const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples, float slope, float *d_Y, int ldy,
                    float *d_decoded, void *stream);

/* Residual + color_final_block + sigmoid*1.002-0.001 (:1294-1295, :1334, :478-482), scattered with sigma into
 * d_decoded [R*SR,4] (pre-zeroed by the caller; :1337-1338). */
int hnr_final_color(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                    const float *d_sigma, const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples,
                    float *d_decoded, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 4: ray_dist (neural_points_volumetric_model.py:331-339) + ray_march with radiance render / alpha
 * blend (models/rendering/diff_ray_marching.py:508-557) + fill_invalid (:87-126), in input-ray order:
 *   d_raycolor [R,3] (bg colour where ray_mask = 0), d_opacity [R,SR], d_is_background [R] (T_end; 1 where
 *   ray_mask = 0), optional d_blend_weight [R,SR].
 * d_ray_nsamp: NULL when the query outputs are padded (pad_outputs = 1); else slots >= d_ray_nsamp[r] are taken as
 * the padding values (position 0, no neighbour) without being read. */
int hnr_composite(const float *d_decoded, const float *d_sample_loc_w, const int32_t *d_sample_pidx, const int8_t *d_ray_mask,
                  const int32_t *d_ray_nsamp,
                  const float *d_campos, const float *d_camrot, const float *d_bg_color, int R, int SR, int K, float vsize_z,
                  int raydist_mode_unit, float *d_raycolor, float *d_opacity, float *d_is_background, float *d_blend_weight,
                  void *stream);

/* Hole-probing outputs of opt.prob == 1 (models/neural_points_volumetric_model.py:392-416; consumed by the point-growing step
 * of run/train_ft.py:450-569): per ray the sample of maximum opacity -> d_max_opacity [R], its position d_max_loc_w [R,3],
 * the distance to the nearest of its K listed points d_far_dist [R] (empty slots read point 0, like the reference's clamped
 * gather), and sum_k (weight * conf_coefficient) x {color, dir, conf, embedding} of those points.  Rows = input rays. */
int hnr_probe_outputs(const float *d_opacity, const float *d_sample_loc_w, const int32_t *d_sample_pidx, const float *d_weight,
                      const float *d_conf_coefficient, const float *d_xyz, const float *d_emb, const float *d_conf, const float *d_dir,
                      const float *d_color, int F, int R, int SR, int K, float *d_max_opacity, float *d_max_loc_w, float *d_far_dist,
                      float *d_avg_color, float *d_avg_dir, float *d_avg_conf, float *d_avg_emb, void *stream);

/* ray_march alone (models/rendering/diff_ray_marching.py:508-557; radiance render + alpha blend) on existing tensors:
 * d_ray_dist [R,SR], d_ray_valid [R,SR] u8, d_features [R,SR,4] = (sigma, rgb), d_bg_color [3] or NULL ->
 * d_ray_color [R,3], d_opacity / d_acc_transmission / d_blend_weight [R,SR], d_bg_transmission [R]. */
int hnr_ray_march(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *d_bg_color, int R,
                  int SR, float *d_ray_color, float *d_opacity, float *d_acc_transmission, float *d_blend_weight,
                  float *d_bg_transmission, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stage 5: backward of stages 3b and 4 (hnr_gather_aggregate_bwd / hnr_composite_bwd), one entry point per
 * forward kernel.  The reference obtains all of these from torch autograd over its eager ops; file:line below name the
 * forward code whose derivative each one is.  g_* / d_g* are gradients; weight gradients marked "atomics" are ADDED to
 * the caller's (zero-initialised or accumulating) buffers.
 */

/* ray_march + alpha blend (models/rendering/diff_ray_marching.py:508-557) and fill_invalid (:87-126):
 * d_g_raycolor [R,3] -> d_g_decoded [R,SR,4] = (d sigma, d rgb), zeros for invalid samples / rays. */
int hnr_composite_bwd(const float *d_decoded, const float *d_sample_loc_w, const int32_t *d_sample_pidx, const int8_t *d_ray_mask,
                      const int32_t *d_ray_nsamp, const float *d_campos, const float *d_camrot, const float *d_bg_color,
                      int R, int SR, int K, float vsize_z, int raydist_mode_unit, const float *d_g_raycolor,
                      float *d_g_decoded, void *stream);

/* color_final_block + sigmoid*1.002-0.001 + residual (point_aggregators.py:1294-1295, :1334, :478-482):
 * d_g_decoded -> d_gY [S,45] (mix-up output), d_gCF [S,128] (OVERWRITTEN), d_g_sigma [S]; weights: atomics. */
int hnr_final_color_bwd(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin,
                        const int32_t *d_vs_item, const int64_t *d_counts, int cap_samples, const float *d_g_decoded,
                        float *d_gY, int ldgy, float *d_gCF, int ldgcf, float *d_g_sigma, float *d_g_w_fin, float *d_g_b_fin,
                        void *stream);

/* weighted merge + last layer of aux_merge_weight_block (:1199-1217, patch drop :1222-1237, mix-up input :1286-1292):
 * d_gX7 [S,90] -> d_gF [V*cap,48] (d image-feature columns), d_gZ3 [V*cap,64] (d pre-activation of the last hidden layer),
 * d_gCF[:, :45] += ; last-layer weights: atomics. */
int hnr_merge_bwd(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last,
                  const float *d_vmask, const float *d_frame_w, const int64_t *d_counts, int V, int cap_samples, float slope,
                  const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR, const float *d_gX7, int ldg7,
                  float *d_gF, int ldgf, float *d_gZ3, int ldgz, float *d_gCF, int ldgcf, float *d_g_w_last, float *d_g_b_last,
                  void *stream);

/* pixel gather (:1077-1089, :1193) + F.interpolate (:1064-1067) transposed: the rows' image-feature gradients (two sources
 * added: d_gFa from hnr_merge_bwd, optional d_gFb from the merge-weight MLP's first layer) are first added to their pixel
 * of d_g_featmap ([V,H,W,48], ZERO-INITIALISED; d_bbox int32[V,4] initialised to {W,H,-1,-1} receives the touched
 * rectangle; rows are added to their pixel with float atomics, row strides lda/ldb >= 48; d_key_scratch: 3 * V * cap_samples int32,
 * d_sort_scratch is no longer used), then gathered with the bilinear weights into the s1/s2/s3 slots of d_g_pyramid, a ZERO-INITIALISED buffer
 * laid out like the forward scratch of hnr_image_features. */
int hnr_proj_rows_bwd(const float *d_sample_loc_w, const int32_t *d_vs_item, const int64_t *d_counts, const float *d_w2c,
                      const float *d_intrinsic, int V, int H, int W, int cap_samples, const float *d_gFa, int lda,
                      const float *d_gFb, int ldb, float *d_g_featmap, int32_t *d_bbox, float *d_g_pyramid,
                      int32_t *d_key_scratch /* int32[3*V*cap_samples] */, void *d_sort_scratch,
                      int64_t sort_scratch_bytes /* hnr_sort_rows_scratch_bytes(V*cap_samples) */, void *stream);

/* Sum-by-key of gradient rows (torch autograd's index_add / scatter in the reference) without per-element atomics:
 *   hnr_sort_rows_by_key : stable radix sort of (key, row index); keys < 0 sort first and are skipped later;
 *   hnr_segment_sum_rows : d_dst[key * dst_stride + c] += sum over the rows with that key of (A[row,c] + B[row,c]),
 *                          c < n_cols (multiple of 4, <= 256; d_B may be NULL).
 * Used for d(per-point table) (rows -> touched point) and d(feature map) (rows -> pixel). */
int64_t hnr_sort_rows_scratch_bytes(int64_t M);
int hnr_sort_rows_by_key(const int32_t *d_keys, int64_t M, int32_t *d_keys_sorted, int32_t *d_perm, void *d_scratch,
                         int64_t scratch_bytes, void *stream);
int hnr_segment_sum_rows(const float *d_A, int lda, const float *d_B, int ldb, const int32_t *d_keys_sorted,
                         const int32_t *d_perm, int64_t M, int n_cols, float *d_dst, int64_t dst_stride, void *stream);

/* aux_block_s1..3 (:1047-1063): d_scratch is the forward scratch (activations), d_g_pyramid as above (used as workspace);
 * g_conv_w / g_conv_b: HOST arrays of 6 device pointers, atomics. */
int hnr_image_features_bwd(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope,
                           const float *d_scratch, float *d_g_pyramid, float *const *g_conv_w, float *const *g_conv_b,
                           void *stream);

/* NeuralPoints gather (neural_points.py:709-720), block3 extras (:957-971), conf straight-through clamp (:1422-1424, :1508-1512) transposed:
 * d_gX3[:, 256:263], d_g_wagg, optional d_g_conf_out [R,SR,K] (gradient of the conf_coefficient output) -> per-row contributions
 * [d color 3 | d dir 3 | d conf | 0] in d_G8 [rows, 8]; hnr_segment_sum_rows_det over the rows sorted by touched-point index
 * (hnr_sort_rows_by_key) adds them per point in a fixed order: bit-identical gradients run to run, no float atomics. */
int hnr_gather_rows_bwd_rows(const int32_t *d_sample_pidx, const float *d_raydir, const int32_t *d_vs_item, const int32_t *d_vs_off,
                             const int32_t *d_vs_cnt, const int64_t *d_counts, int SR, int K, int cap_samples, const float *d_gX3,
                             int ldg3, const float *d_g_wagg, const float *d_weight, const float *d_g_conf_out, float *d_G8, void *stream);
/* dst[(dst_index ? dst_index[k] : k), 0:n_cols] (+)= sum of the rows A[perm[e], :] with keys_sorted[e] == k, for the DENSE keys k = 0 .. n_keys-1;
 * one wave per key, rows added in sorted (= original row) order, no atomics.  n_cols a multiple of 4, <= 256. */
int hnr_segment_sum_rows_det(const float *d_A, int lda, const int32_t *d_keys_sorted, const int32_t *d_perm, int64_t M, int n_cols,
                             int n_keys, const int32_t *d_dst_index, float *d_dst, int64_t dst_stride, int accumulate, void *stream);
/* The form the training step uses (torch autograd's index_add for the point-buffer gradients, models/neural_points/neural_points.py:709-720
 * transposed): the rows of key k are listed, in ANY order, in d_row_list[d_seg_start[k] .. + d_seg_count[k]) (a point-major list built without a
 * sort); dst[k, 0:n_cols] = their sum, added in an order that depends on the row indices only (one workgroup per key: row indices rank-sorted
 * in LDS, four partial sums over the sorted quarters added in wave order) -- bit-identical run to run, no float atomics.  Row indices of a
 * segment must be distinct.  Optional second matrix d_A2 (n_cols2 <= 256) summed into d_dst2 over the same segments; optional d_absmax:
 * max |dst| as a float bit pattern with atomicMax (exponent-exact: see csrc/hnr_common.h absmax_publish).  n_cols, n_cols2 multiples of 4. */
int hnr_segment_sum_rows_csr(const float *d_A, int lda, const int32_t *d_row_list, const int32_t *d_seg_start, const int32_t *d_seg_count,
                             int n_cols, int n_keys, float *d_dst, int64_t dst_stride, const float *d_A2, int lda2, int n_cols2,
                             float *d_dst2, int64_t dst_stride2, uint32_t *d_absmax, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Dense layers of the TRAINING step on the 16-bit matrix pipe (csrc/h2gemm.hip), fp32 in / fp32 out, the chain's two-term fp16 split
 * arithmetic.  They replace what torch autograd runs for the nn.Linear (+ LeakyReLU) layers of PointAggregator.viewmlp in the reference's
 * train step (models/aggregators/point_aggregators.py:948, :972, :1037, :1199, :1292; loss.backward() at
 * models/mvs_points_volumetric_model.py:111-131).  Every row count is read on the DEVICE: rows = min(M_cap, *d_m) (d_m may be NULL).
 * n_seg > 1: the rows are n_seg SEGMENTS of min(*d_m, seg_stride) rows each, segment v starting at physical row v * seg_stride -- the
 * (view, sample) rows of hnr_proj_rows (row = v * cap_samples + s); n_seg = 1: plain rows.
 *   hnr_h2lin_pack   n_jobs <= 16 weight matrices -> kernel images, two launches for all of them.  Element (n, k) of job j is
 *                    d_W[j][n * rs[j] + k * cs[j]]: (rs, cs) = (ld, 1) packs an nn.Linear weight [N, K], (1, ld) its transpose (input
 *                    gradients: dX = dZ W is the layer "dZ (W^T)^T").  N <= 256, K <= 288; image bytes: hnr_h2lin_packed_bytes(K).
 *   hnr_h2lin        mode 0: C = act(A W^T + bias)  (act != 0: LeakyReLU(slope));
 *                    mode 1: C = (A W^T) * (side > 0 ? 1 : slope)   -- side [M, ld_side] is the stored forward activation: the input
 *                    gradient through a LeakyReLU.  K in {<= 48, <= 64, <= 128, <= 224, <= 256}.  d_absmax (optional): max |C| is
 *                    atomically max-ed into it as a bit pattern (the scale of a later hnr_h2wgrad).
 *   hnr_h2lin_dgrad_bits   mode 1 of hnr_h2lin for the per-neighbour chain (N = 256, whole 32-row tiles: M_cap % 32 == 0): LeakyReLU' comes from the
 *                    SIGN WORDS the training forward leaves (one word per (32-row tile, wave = 64-column group, lane): bit 31 - i = (value i of
 *                    the lane's 32 columns 64 wave + 16 (lane >> 5) + {0..15, 32..47} of row (lane & 31) is > 0)) instead of the stored
 *                    activation.  K = 256 runs the weight-stationary kernel (csrc/h2lin_ws.hip); results equal hnr_h2lin's bit for bit.
 *   hnr_h2wgrad      dW[N, K] (row stride lddw) = dZ[M, N]^T X[M, K], db[N] = column sums of dZ (d_db may be NULL); accumulate != 0 adds.
 *                    d_absmax_z / d_absmax_x: bit patterns of (an upper bound of) max |dZ|, max |X| (hnr_absmax or a producer's output).
 *                    N <= 256, K <= 287.  Deterministic (fixed-order partials, no atomics); d_scratch: hnr_h2wgrad_scratch_bytes(N, K).
 *   hnr_absmax       *d_out = max(*d_out, max |A[0:M, 0:N]|) as a bit pattern. */
int64_t hnr_h2lin_packed_bytes(int K);
int hnr_h2lin_pack(int n_jobs, const float *const *d_W, const int64_t *rs, const int64_t *cs, const int *N, const int *K,
                   const float *const *d_bias /*may be NULL*/, void *const *d_packed, void *stream);
int hnr_h2lin(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, const void *d_packed, int N, int K, int mode,
              int act, float slope, const float *d_side, int ld_side, float *d_C, int ldc, uint32_t *d_absmax, void *stream);
int hnr_h2lin_dgrad_bits(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, int N, int K, float slope,
                         const uint32_t *d_side_bits, float *d_C, int ldc, uint32_t *d_absmax, void *stream);
int64_t hnr_h2wgrad_scratch_bytes(int N, int K);
int hnr_h2wgrad(const float *d_dZ, int ldz, const float *d_X, int ldx, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, int N, int K,
                const uint32_t *d_absmax_z, const uint32_t *d_absmax_x, float *d_dW, int lddw, float *d_db, int accumulate,
                void *d_scratch, void *stream);
int hnr_absmax(const float *d_A, int lda, int64_t M_cap, const int64_t *d_m, int n_seg, int64_t seg_stride, int N, uint32_t *d_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The whole forward path in ONE call (csrc/render_forward.hip): one pass of NeuralPointsRayMarching.forward + fill_invalid
 * (models/neural_points_volumetric_model.py:257-391, :87-126) over R rays -- query, gather / aggregate (fused chain + fused
 * per-sample MLPs), composite -- with every launch issued here, back to back on the caller's stream.  The reference's three host
 * synchronisations per chunk (query_point_indices_worldcoords.py:645-646, :705; point_aggregators.py:1092) have no counterpart:
 * the kernels read their work sizes from the query's device counters, the intermediate buffers live in the caller's workspace
 * (hnr_render_workspace_bytes) sized for `cap_samples` valid shading samples (R*SR is always enough), and d_status[0] becomes 1 when
 * that capacity was exceeded (the extra samples are dropped; d_status[1] holds the true count's low 32 bits).  K = 8 only.
 * All pointers are device pointers; nothing is allocated, nothing is read back. */
typedef struct {
    int   R, SR, K, D;            /* rays, opt.SR, opt.K (8), opt.z_depth_dim                                          */
    int   tmid_stride;            /* 0: cam->d_tmid is one [D] table; D: [R,D] per-ray tables (train-time jitter)       */
    int   kernel_size[3];
    float radius2;                /* radius_limit^2                                                                    */
    float vsize_z;                /* opt.vsize[2] (ray_dist rule, :331-339)                                            */
    int   raydist_mode_unit;
    int   V;                      /* reference views (0: use_nearest = 0, image branch off)                            */
    int   cap_samples;            /* capacity of the workspace in valid shading samples                                */
    int   knn_order;              /* hnr_query_params.knn_order                                                        */
} hnr_render_params;
typedef struct {
    const float *d_xyz, *d_conf, *d_dir, *d_color;      /* [N,3] [N] [N,3] [N,3]                                       */
    const float *d_point_table; int ldt;                /* [N, ldt >= 256] per-point addend of block1.0                 */
    const float *d_rec;                                 /* optional hnr_point_records image of the four buffers above ([N,12]); NULL: the
                                                           gather reads the four buffers                                */
} hnr_render_cloud;
typedef struct {
    const void  *d_chain;                               /* hnr_chain_pack                                               */
    const void  *d_mlp_cf, *d_mlp_mw, *d_mlp_mx;        /* hnr_mlp3_pack images: colour feature (4 layers: + the 128 -> 64 colour-feature
                                                           columns of aux_merge_weight_block.0 with its bias), merge weights, mix-up */
    const float *d_mw_last_w, *d_mw_last_b;             /* aux_merge_weight_block.6: [64], [1]                           */
    const float *d_fin_w, *d_fin_b;                     /* color_final_block.0: [3,128], [3]                             */
    float slope;                                        /* LeakyReLU slope                                               */
} hnr_render_weights;
typedef struct { const float *d_campos, *d_camrot, *d_raydir, *d_tmid, *d_bg_color; } hnr_render_camera;
typedef struct {
    const float *d_w2c, *d_intrinsic, *d_campos_nearest; /* [V,4,4] inverse(c2w_nearest), [3,3], [V,3]                  */
    const float *d_featmap; int H, W;                    /* hnr_image_features output [V,H,W,48]                         */
    const float *d_frame_w;                              /* optional [V]                                                 */
    void        *featmap_ready;                          /* optional hipEvent_t: the launch stream waits for it before the first kernel that
                                                            reads d_featmap (the merge stage), so the caller may build the feature map on
                                                            another stream while the query and the per-neighbour chain run            */
} hnr_render_views;
typedef struct {
    float   *d_raycolor, *d_opacity, *d_is_background;   /* [R,3] [R,SR] [R]  (fill_invalid applied)                     */
    float   *d_blend_weight;                             /* optional [R,SR]                                              */
    int8_t  *d_ray_mask;                                 /* [R]                                                          */
    float   *d_decoded;                                  /* [R,SR,4] (sigma, rgb)                                        */
    int32_t *d_sample_pidx; float *d_sample_loc_w; int32_t *d_ray_nsamp;    /* the query outputs, un-padded               */
    int64_t *d_counts;                                   /* [HNR_NCOUNTS]                                                */
    int32_t *d_status;                                   /* [2]                                                          */
    float   *d_weight, *d_conf_coefficient;              /* optional [R,SR,K] (both or neither)                          */
    void   **stage_events;                               /* optional: HNR_RENDER_NSTAGES + 1 hipEvent_t handles recorded at the
                                                            stage boundaries (query, plan, chain_gather, chain, mlp_colorfeat,
                                                            proj_rows, mlp_merge, merge, mlp_mixup, final_color, composite) */
} hnr_render_outputs;
#define HNR_RENDER_NSTAGES 11
int64_t hnr_render_workspace_bytes(const hnr_render_params *p);
int hnr_render_forward(const hnr_grid *grid, const hnr_render_params *p, const hnr_render_cloud *cloud, const hnr_render_weights *weights,
                       const hnr_render_camera *camera, const hnr_render_views *views, void *d_workspace, int64_t workspace_bytes,
                       const hnr_render_outputs *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The TRAINING step of the path as two calls (csrc/render_train.hip): hnr_render_train_forward = one pass of NeuralPointsRayMarching.forward in
 * train mode + fill_invalid (models/neural_points_volumetric_model.py:257-427, :87-126: jittered depths through cam->d_tmid with
 * tmid_stride = D, query_point_indices_worldcoords.py:87; patch drop point_aggregators.py:1222-1237; straight-through conf clamp :1422-1424),
 * keeping in the caller's workspace what the backward pass needs; hnr_render_train_backward = what loss.backward() makes torch autograd compute
 * for it (models/mvs_points_volumetric_model.py:111-131; there is no custom autograd.Function in the reference): the gradients of
 * points_embeding / points_conf / points_dir / points_color and of every aggregator parameter of the order-2 hybrid path, from the gradients of
 * coarse_raycolor [R,3] and (optionally) conf_coefficient [R,SR,K].  Every launch is issued by the library on the caller's stream; every work
 * size (valid samples, neighbour rows, touched points) is read from device counters; nothing is allocated, nothing is read back.
 * K = 8, point_features_dim = 32.  The weights are the raw nn.Linear / nn.Conv2d tensors under the reference's names (their kernel images are
 * re-packed every step -- all of them, the backward call's transposed images included, by the FORWARD call: hnr_render_train_backward needs the
 * forward call of the same step, with the same weights and workspace, to have run, as it does for the stored activations).  Weight gradients are OVERWRITTEN, point gradients too ([N,32] [N] [N,3] [N,3], zero
 * outside the batch's points).  Per-point sums are formed in a fixed order: bit-identical gradients run to run.
 *   d_drop_lut [R] u8 (optional): patch-drop pattern indexed by VALID-ray row (drop_patch_rays, point_aggregators.py:14-23, :1225-1233);
 *   d_ray_drop [R] u8 (optional, wins): explicit per-ray flags (a rank's slice of a batch-wide pattern); both are ANDed with ray_mask.
 * `out` as for hnr_render_forward (padded query outputs; d_blend_weight, d_weight, d_conf_coefficient required; stage_events: optional HIP events recorded at the stage boundaries, as in hnr_render_forward -- bench.py reads them);
 * d_status[0] = 1 when cap_samples was exceeded (R * SR always suffices). */
typedef struct {
    int   R, SR, K, D;
    int   tmid_stride;
    int   kernel_size[3];
    float radius2, vsize_z;
    int   raydist_mode_unit;
    int   V, H, W;                /* reference views and their size (V = 0: use_nearest = 0, image branch off)             */
    int   n_points;               /* N                                                                                    */
    int   cap_samples;
    int   knn_order;
    float slope;                  /* LeakyReLU slope                                                                      */
} hnr_train_params;
typedef struct { const float *d_xyz, *d_emb, *d_conf, *d_dir, *d_color; } hnr_train_cloud;          /* [N,3] [N,32] [N] [N,3] [N,3] */
typedef struct { float *d_emb, *d_conf, *d_dir, *d_color; } hnr_train_cloud_grads;
typedef struct {   /* aggregator.{block1.0, block1.2, block3.0, block3.2, alpha_branch.0, color_feature_branch.{0,2,4}, aux_merge_weight_block.{0,2,4,6},
                      color_mixup_block.{0,2,4}, color_final_block.0, aux_block_s1.{0,2}, s2.{0,2}, s3.{0,2}} .weight / .bias (device pointers) */
    const float *block1_0_w, *block1_0_b, *block1_2_w, *block1_2_b, *block3_0_w, *block3_0_b, *block3_2_w, *block3_2_b, *alpha_w, *alpha_b;
    const float *cf_w[3], *cf_b[3], *mw_w[4], *mw_b[4], *mx_w[3], *mx_b[3], *fin_w, *fin_b, *conv_w[6], *conv_b[6];
} hnr_train_weights;           /* the gradient block handed to hnr_render_train_backward has the same layout (its buffers are written)      */
typedef struct { const float *d_w2c, *d_intrinsic, *d_campos_nearest, *d_images /*[V,H,W,3]*/, *d_frame_w /*optional [V]*/; } hnr_train_views;
int64_t hnr_render_train_workspace_bytes(const hnr_train_params *p);
/* STAGE tier (tools only): byte offset / size pairs of the forward pass's intermediate tensors inside the workspace, in the order
 * Xd H1 X3 H3 H4 E Tu X5 sigma T1 T2 CF X6 vmask M1 M2 M3 X7 Y1 Y2 Y3 fm row_pid vs_item fm_scratch (out[2 i], out[2 i + 1]); returns the number of entries or -1.
 * The training-mode counterpart of reading PointAggregator.viewmlp's locals in a debugger (point_aggregators.py:892-1338). */
int hnr_render_train_debug_layout(const hnr_train_params *p, int64_t *out, int max_entries);
/* After hnr_render_train_forward: the batch's touched points inside the workspace -- *d_ids = ascending point ids [*capacity], *d_count = their number
 * (one int64 on the device, already clamped to the capacity).  The reference has no counterpart (autograd's index_select backward writes dense
 * [N, C] gradients, neural_points.py:712-720); a rank's gradient exchange (parallel.PointGradExchange) and a sparse optimiser step read it instead of
 * torch.unique(sample_pidx).  Host pointers only: nothing is launched. */
int hnr_render_train_touched(const hnr_train_params *p, void *d_workspace, int64_t workspace_bytes, const int32_t **d_ids, const int64_t **d_count,
                             int64_t *capacity);
int hnr_render_train_forward(const hnr_grid *grid, const hnr_train_params *p, const hnr_train_cloud *cloud, const hnr_train_weights *weights,
                             const hnr_render_camera *camera, const hnr_train_views *views, const uint8_t *d_drop_lut, const uint8_t *d_ray_drop,
                             void *d_workspace, int64_t workspace_bytes, const hnr_render_outputs *out, void *stream);
int hnr_render_train_backward(const hnr_train_params *p, const hnr_train_cloud *cloud, const hnr_train_weights *weights, const hnr_render_camera *camera,
                              const hnr_train_views *views, void *d_workspace, int64_t workspace_bytes, const hnr_render_outputs *forward_out,
                              const float *d_g_raycolor, const float *d_g_conf_coefficient /*may be NULL*/, const hnr_train_cloud_grads *cloud_grads,
                              const hnr_train_weights *weight_grads, void *stream);

/* ------------------------------------------------------------------------------------------------
 * "Next" row: hole probing, run/train_ft.py:527-549 (`probe_hole`) + :571-581 (`bloat_inds`).  Per probed frame: a cast
 * ray that found no neighbour although its ground-truth pixel is not background (|gt - bg| > 0.002) marks a hole; every hit ray
 * whose max-opacity sample exceeds opacity_thresh and that lies in the 3x3 neighbourhood of a hole -- or, when far_thresh > 0, whose
 * nearest neighbour is farther than far_thresh while its colour is within 0.1 of the ground truth -- becomes a new point.
 * d_sel [h*w] receives ray id + 1 at the selected pixels (0 elsewhere); the caller compacts it in row-major order.
 * d_pixel_idx [R,2] (x, y) as floats, d_ray_mask [R] float, bg3_host: 3 HOST floats, d_miss_scratch int32[h*w]. */
int hnr_probe_select(const float *d_pixel_idx, const float *d_ray_mask, const float *d_gt, const float *d_raycolor, const float *bg3_host,
                     const float *d_far_dist, const float *d_opacity, int R, int h, int w, float far_thresh, float opacity_thresh,
                     int32_t *d_miss_scratch, int32_t *d_sel, void *stream);

/* ------------------------------------------------------------------------------------------------
 * "Next" row: blur-handling module with pre-defined kernels, models/base_rendering_model.py:677-745
 * (`blur_update_output`, called at mvs_points_volumetric_model.py:145-146).  d_color / d_gt / d_out: [S*S,3] with
 * S = patch_num * patch_size in the dilated-patch ray layout; d_kernels [n_kernels, ks, ks]; per patch the candidate
 * (n_kernels blurred versions, normalised at the borders, + the un-blurred patch as candidate n_kernels) closest in L1 to
 * the ground truth replaces the patch; d_select [patch_num^2] records the choice for the backward.
 * patch_num < 0: the buffers hold -patch_num whole patches packed patch-major (patch, y, x) -- a rank's share of a batch
 * that is sharded over GPUs by whole patches (parallel.shard_patches); d_select then has -patch_num entries. */
int hnr_blur_select(const float *d_color, const float *d_gt, const float *d_kernels, int n_kernels, int kernel_size,
                    int patch_num, int patch_size, float *d_out, int32_t *d_select, void *stream);
int hnr_blur_select_bwd(const float *d_g_out, const float *d_kernels, const int32_t *d_select, int n_kernels, int kernel_size,
                        int patch_num, int patch_size, float *d_g_in, void *stream);

/* "Next" row, learnable blur kernels: models/base_rendering_model.py:827-1020 (`learnable_blur_update_output`,
 * faster_version; called at mvs_points_volumetric_model.py:143-144 when the aggregator returns a blur predictor).
 *   hnr_blur_gray_patches      d_gray [n_patches, 2, ps, ps]: channel 0 = grey ground-truth patch, 1 = grey rendered patch (:886-893),
 *                              the predictor's input; _bwd returns the gradient w.r.t. the rendered colours.
 *   hnr_blur_apply             every patch convolved with ITS kernel d_kernels[patch] (grouped F.conv2d, zero padding ks/2) under
 *                              opt.boundary_mode 0 / 1 / 2 (:915-923); _bwd gives the gradients w.r.t. the colours and the kernels.
 * Ray layout and the patch_num < 0 (patch-major) convention as for hnr_blur_select.  kernel_size odd, <= 15; patch_size <= 16. */
int hnr_blur_gray_patches(const float *d_color, const float *d_gt, int patch_num, int patch_size, float *d_gray, void *stream);
int hnr_blur_gray_patches_bwd(const float *d_g_gray, int patch_num, int patch_size, float *d_g_color, void *stream);
int hnr_blur_apply(const float *d_color, const float *d_kernels, int kernel_size, int patch_num, int patch_size, int boundary_mode,
                   float *d_out, void *stream);
int hnr_blur_apply_bwd(const float *d_g_out, const float *d_color, const float *d_kernels, int kernel_size, int patch_num, int patch_size,
                       int boundary_mode, float *d_g_color, float *d_g_kernels, void *stream);

/* ------------------------------------------------------------------------------------------------
 * "Next" row (second half): the loss terms of the shipped training configurations, value + gradients.
 * BaseRenderingModel.compute_losses, models/base_rendering_model.py: colour item `ray_masked_coarse_raycolor` (:1113-1118:
 * MSE over the rays with ray_mask > 0, 0 when none; `+ 1e-6` per item :1198; `* frame_weight` :1204-1205) and the zero-one
 * regulariser on conf_coefficient (:1228-1240: mean(log v + log(1 - v)), v = clamp(x, eps, 1 - eps)).
 *   d_out4 = {total, colour MSE, zero-one mean, number of valid rays};  total = (mse * w_color + 1e-6) * frame_weight + zo * w_zero_one
 *   d_g_color [R,3], d_g_conf [n_conf]: d total / d input (both NULL = value only).  d_scratch: hnr_shipped_loss_scratch_bytes(). */
int64_t hnr_shipped_loss_scratch_bytes(void);
int hnr_shipped_loss(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int64_t n_conf,
                     float zero_epsilon, float w_color, float w_zero_one, float frame_weight, float *d_out4, float *d_g_color,
                     float *d_g_conf, void *d_scratch, void *stream);
/* The same with d_conf [R, conf_per_ray] holding a row per ray of the batch (the layout hnr_render_train_forward writes): only the rows of rays with
 * ray_mask > 0 enter the mean (and receive a gradient) -- the reference's conf_coefficient holds the R' valid rays only
 * (PointAggregator.forward's return, models/aggregators/point_aggregators.py:1519-1522) -- without a masked copy or a host read of the number of valid rays. */
int hnr_shipped_loss_rows(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int conf_per_ray,
                          float zero_epsilon, float w_color, float w_zero_one, float frame_weight, float *d_out4, float *d_g_color,
                          float *d_g_conf, void *d_scratch, void *stream);
/* ... with the item's frame weight (:1204-1205) read from the device, d_frame_weight[0]: a training step captured in a hipGraph (train.CapturedTrainStep)
 * is replayed for dataset items with different weights. */
int hnr_shipped_loss_rows_fw(const float *d_color, const float *d_gt, const int8_t *d_ray_mask, int R, const float *d_conf, int conf_per_ray,
                             float zero_epsilon, float w_color, float w_zero_one, const float *d_frame_weight, float *d_out4, float *d_g_color,
                             float *d_g_conf, void *d_scratch, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU training (BASELINE config C5): sparse exchange of the point-buffer gradients of a patch-sharded step.  No reference counterpart
 * (its DataParallel wrapper runs on gpu_ids[0] only, models/neural_points_volumetric_model.py:161-167; autograd writes dense [N, C] gradients).
 *   hnr_point_grad_pack   d_ids / d_count: the batch's touched points (hnr_render_train_touched); d_n_valid: this rank's number of valid rays (one
 *                         float: d_out4[3] of hnr_shipped_loss_rows); the four dense gradients of hnr_render_train_backward.
 *                         d_rec [capacity + 2][40] floats: row 0 = {records, valid rays, overflow flag}; rows 1..capacity = {point id (int32 bits) |
 *                         emb 32 | conf | dir 3 | colour 3}, unused slots id -1; row capacity + 1 = point 0 when the batch did not touch it (the empty
 *                         neighbour slots' d conf_coefficient lands there, neural_points.py:711).
 *   hnr_point_grad_apply  d_all_rec [n_ranks][capacity + 2][40] (ONE all-gather of the ranks' d_rec): rewrites the dense gradients in place as
 *                         sum_r (n_r / n) g_r -- the gradient of the loss's mean over ALL ranks' valid rays -- adding the ranks' rows in rank order (the
 *                         same bits on every rank); only rows some rank touched are written.  d_out2 (optional) = {n, overflow flag of any rank}. */
int hnr_point_grad_pack(const int32_t *d_ids, const int64_t *d_count, int capacity, const float *d_n_valid, const float *d_g_emb, const float *d_g_conf,
                        const float *d_g_dir, const float *d_g_color, float *d_rec, void *stream);
int hnr_point_grad_apply(const float *d_all_rec, int n_ranks, int capacity, float *d_g_emb, float *d_g_conf, float *d_g_dir, float *d_g_color, int n_points,
                         float *d_out2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * "Next" row: voxel down-sampling of the initial point cloud, models/mvs/mvs_utils.py:537-563
 * (`construct_vox_points_closest`, called at run/train_ft.py:164 and :725; needs torch_scatter in the reference).
 *   cell = floor((xyz - space_min) / vox_size)  (fp32 subtract, fp32 divide);  voxels in the lexicographic order of torch.unique(dim=0);
 *   d_centroid [V,3] per-voxel mean (sequential fp32 sum in point-id order), d_grid_idx [V,3] the cells, d_min_idx [V] the point
 *   closest to its voxel's centroid (first one on ties), d_inverse [n] voxel of every point (may be NULL), d_count[0] = V.
 * Output arrays are sized for n voxels.  space_min: 3 host floats.  Synchronises the stream (it reports out-of-range points). */
int64_t hnr_voxel_downsample_scratch_bytes(int64_t n);
int hnr_voxel_downsample(const float *d_xyz, int n, const float *space_min, float vox_size, float *d_centroid, int32_t *d_grid_idx,
                         int32_t *d_min_idx, int32_t *d_inverse, int64_t *d_count, void *d_scratch, int64_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HNR_H */
