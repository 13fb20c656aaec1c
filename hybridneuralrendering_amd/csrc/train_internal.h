// Library-internal entry points of the training step (csrc/render_train.hip drives them): forms of the stage launchers whose work sizes are
// read on the device, and the training variants of the fused forward kernels that keep their activations.
#pragma once
#include "hnr_common.h"

namespace hnr {

// csrc/chain.hip
int chain_gather_train(const float *d_xyz, const float *d_conf, const float *d_dir, const float *d_color, const int32_t *d_sample_pidx,
                       const float *d_sample_loc_w, const float *d_raydir, const float *d_campos, const float *d_camrot, const int32_t *d_vs_item,
                       const int64_t *d_counts, int SR, int K, int cap_samples, void *d_workspace, float *d_X5, int ld5, float *d_weight_out,
                       float *d_conf_out, float *d_Xd, int32_t *d_row_pid, void *stream);
int chain_forward_train(const void *d_workspace, const float *d_point_table, int ldt, const int32_t *d_uidx, const void *d_packed, const int64_t *d_counts,
                        int cap_samples, float slope, float *d_X5, int ld5, float *d_sigma, float *const *d_H, const int *ldh, uint32_t *d_hmax, uint32_t *d_x5max, void *stream,
                        const int32_t *d_row_u = nullptr, int ucap = 0, uint32_t *d_hbits = nullptr, long long hbits_stride = 0);
// csrc/h2gemm.hip: dX = (dZ W) * LeakyReLU'(forward activation), the activation's signs as the chain kernels' bit words (ChainArgs::hbits)
int h2lin_dgrad_bits(const float *d_dZ, int ldz, int64_t M_cap, const int64_t *d_m, const void *d_packed, int N, int K, float slope, const uint32_t *d_side_bits,
                     float *d_C, int ldc, uint32_t *d_absmax, void *stream);
// csrc/mlp.hip
int mlp3_forward_train(const float *d_A, int lda, int64_t M_cap, const int64_t *d_counts, int count_index, int count_mult, int seg_stride,
                       const void *d_packed, int n_layers, const int *N, const int *K, const int *act, float slope, const float *d_R,
                       const int32_t *d_ridx, int ldr, float *d_C, int ldc, float *d_C2, int ldc2, float *d_T0, int ldt0, float *d_T1, int ldt1,
                       uint32_t *d_tmax, void *stream);
// csrc/aggregate.hip
int point_rows_dc(const float *d_emb, const int32_t *d_ids, int n_cap, const long long *d_n, float *d_E, int lde, hipStream_t st);
// csrc/backward.hip
int unique_points_dc(const int32_t *d_row_pid, int64_t M_cap, const long long *d_m, int n_points, int32_t *d_uidx, int32_t *d_ulist, int cap,
                     int32_t *d_row_u, int32_t *d_count, int32_t *d_scratch, int32_t *d_seg_start, int32_t *d_seg_count, int32_t *d_row_list, hipStream_t st);
int point_small_grads_dc(const float *d_P8, const int32_t *d_ulist, int U_cap, const long long *d_u, float *d_g_conf, float *d_g_dir, float *d_g_color, hipStream_t st);
int point_rows_bwd_dc(const float *d_gE, int ldg, const float *d_E, int lde, const int32_t *d_ids, int n_cap, const long long *d_n, float *d_g_emb, hipStream_t st);
int final_color_bwd_max(const float *d_Y, int ldy, const float *d_CF, int ldcf, const float *d_w_fin, const float *d_b_fin, const int32_t *d_vs_item,
                        const int64_t *d_counts, int cap_samples, const float *d_g_decoded, float *d_gY, int ldgy, float *d_gCF, int ldgcf, float *d_g_sigma,
                        float *d_g_w_fin, float *d_g_b_fin, uint32_t *d_gY_max, void *stream);
int merge_bwd_max(const float *d_X6, int ld6, const float *d_Hm, int ldh, const float *d_w_last, const float *d_b_last, const float *d_vmask, const float *d_frame_w,
                  const int64_t *d_counts, int V, int cap_samples, float slope, const uint8_t *d_ray_drop, const int32_t *d_vs_item, int SR, const float *d_gX7,
                  int ldg7, float *d_gF, int ldgf, float *d_gZ3, int ldgz, float *d_gCF, int ldgcf, float *d_g_w_last, float *d_g_b_last, uint32_t *d_gZ3_max,
                  void *stream);
int dleaky_add_dc(float *d_g, int ldg, const float *d_add, int lda, int n_add, const float *d_y, int ldy, int64_t M_cap, const long long *d_m, int N, float slope,
                  uint32_t *d_absmax, hipStream_t st);
int sum_views_dc(const float *d_in, int ldi, int V, int cap, const long long *d_n, int N, float *d_out, int ldo, uint32_t *d_absmax, hipStream_t st);
int image_features_bwd_bbox(const float *d_img, int V, int H, int W, const float *const *conv_w, float slope, const float *d_scratch, float *d_g_pyramid,
                            float *const *g_conv_w, float *const *g_conv_b, const int32_t *d_bbox, void *stream);
// csrc/segment.hip
int sort_rows_by_key_bits(const int32_t *d_keys, int64_t M, int bits, int32_t *d_keys_sorted, int32_t *d_perm, void *d_scratch, int64_t scratch_bytes, hipStream_t st);
int segment_starts(const int32_t *d_keys_sorted, int64_t M, int32_t *d_start, hipStream_t st);
int segment_sum_rows_det_dc(const float *d_A, int lda, const int32_t *d_keys_sorted, const int32_t *d_perm, int64_t M, int n_cols, int keys_cap,
                            const long long *d_nkeys, const int32_t *d_start, float *d_dst, int64_t dst_stride, hipStream_t st);
int segment_sum_rows_csr_dc(const float *d_A, int lda, const int32_t *d_row_list, const int32_t *d_seg_start, const int32_t *d_seg_count, int n_cols, int keys_cap,
                            const long long *d_nkeys, float *d_dst, int64_t dst_stride, const float *d_A2, int lda2, int n_cols2, float *d_dst2, int64_t dst_stride2,
                            uint32_t *d_absmax, hipStream_t st);

}  // namespace hnr
