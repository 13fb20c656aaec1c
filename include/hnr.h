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
 *     hnr_grid handle and a small per-process scratch of counters.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference repo).
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
    HNR_NCOUNTS = 8
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

#ifdef __cplusplus
}
#endif
#endif /* HNR_H */
