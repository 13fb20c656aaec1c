"""ctypes binding of libhnr_hip.so (the C ABI declared in include/hnr.h).

There is NO fallback: if the HIP library is missing or a call fails, this raises.  PyTorch only
supplies device memory (`tensor.data_ptr()`) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HNR_LIB_PATH: another build of the same library (A/B timing of kernel changes); there is no other fallback
LIB_PATH = os.environ.get("HNR_LIB_PATH") or os.path.join(_HERE, "libhnr_hip.so")
NCOUNTS = 9
CNT = dict(RAYS_HIT=0, SAMPLES=1, RAYS_VALID=2, NEIGHBOURS=3, CELLS_VISITED=4, CANDIDATES=5, SAMPLES_VALID=6, SAMPLES_SMALL=7, SAMPLES_TINY=8)


class HnrError(RuntimeError):
    pass


class GridParams(ctypes.Structure):
    _fields_ = [("origin", ctypes.c_float * 3), ("cell", ctypes.c_float * 3), ("dims", ctypes.c_int * 3),
                ("query_size", ctypes.c_int * 3), ("P", ctypes.c_int), ("max_o", ctypes.c_int)]


class GridStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("n_points", "n_inbounds", "n_occ", "n_dropped_voxels",
                                              "n_cells_over_P", "n_dilated", "n_words", "bytes")]


class QueryParams(ctypes.Structure):
    _fields_ = [("R", ctypes.c_int), ("D", ctypes.c_int), ("SR", ctypes.c_int), ("K", ctypes.c_int),
                ("kernel_size", ctypes.c_int * 3), ("radius2", ctypes.c_float), ("tmid_stride", ctypes.c_int),
                ("pad_outputs", ctypes.c_int), ("knn_order", ctypes.c_int)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float


class RenderParams(ctypes.Structure):
    _fields_ = [("R", _I), ("SR", _I), ("K", _I), ("D", _I), ("tmid_stride", _I), ("kernel_size", _I * 3), ("radius2", _F), ("vsize_z", _F),
                ("raydist_mode_unit", _I), ("V", _I), ("cap_samples", _I), ("knn_order", _I)]


class RenderCloud(ctypes.Structure):
    _fields_ = [("d_xyz", _P), ("d_conf", _P), ("d_dir", _P), ("d_color", _P), ("d_point_table", _P), ("ldt", _I), ("d_rec", _P)]


class RenderWeights(ctypes.Structure):
    _fields_ = [("d_chain", _P), ("d_mlp_cf", _P), ("d_mlp_mw", _P), ("d_mlp_mx", _P),
                ("d_mw_last_w", _P), ("d_mw_last_b", _P), ("d_fin_w", _P), ("d_fin_b", _P), ("slope", _F)]


class RenderCamera(ctypes.Structure):
    _fields_ = [("d_campos", _P), ("d_camrot", _P), ("d_raydir", _P), ("d_tmid", _P), ("d_bg_color", _P)]


class RenderViews(ctypes.Structure):
    _fields_ = [("d_w2c", _P), ("d_intrinsic", _P), ("d_campos_nearest", _P), ("d_featmap", _P), ("H", _I), ("W", _I), ("d_frame_w", _P), ("featmap_ready", _P)]


class RenderOutputs(ctypes.Structure):
    _fields_ = [("d_raycolor", _P), ("d_opacity", _P), ("d_is_background", _P), ("d_blend_weight", _P), ("d_ray_mask", _P), ("d_decoded", _P),
                ("d_sample_pidx", _P), ("d_sample_loc_w", _P), ("d_ray_nsamp", _P), ("d_counts", _P), ("d_status", _P), ("d_weight", _P),
                ("d_conf_coefficient", _P), ("stage_events", ctypes.POINTER(_P))]


class TrainParams(ctypes.Structure):
    _fields_ = [("R", _I), ("SR", _I), ("K", _I), ("D", _I), ("tmid_stride", _I), ("kernel_size", _I * 3), ("radius2", _F), ("vsize_z", _F),
                ("raydist_mode_unit", _I), ("V", _I), ("H", _I), ("W", _I), ("n_points", _I), ("cap_samples", _I), ("knn_order", _I), ("slope", _F)]


class TrainCloud(ctypes.Structure):
    _fields_ = [("d_xyz", _P), ("d_emb", _P), ("d_conf", _P), ("d_dir", _P), ("d_color", _P)]


class TrainCloudGrads(ctypes.Structure):
    _fields_ = [("d_emb", _P), ("d_conf", _P), ("d_dir", _P), ("d_color", _P)]


class TrainWeights(ctypes.Structure):
    """hnr_train_weights: raw parameter pointers under the reference's names; the gradient block has the same layout."""
    _fields_ = [(n, _P) for n in ("block1_0_w", "block1_0_b", "block1_2_w", "block1_2_b", "block3_0_w", "block3_0_b", "block3_2_w", "block3_2_b", "alpha_w", "alpha_b")] + \
               [("cf_w", _P * 3), ("cf_b", _P * 3), ("mw_w", _P * 4), ("mw_b", _P * 4), ("mx_w", _P * 3), ("mx_b", _P * 3), ("fin_w", _P), ("fin_b", _P),
                ("conv_w", _P * 6), ("conv_b", _P * 6)]


class TrainViews(ctypes.Structure):
    _fields_ = [("d_w2c", _P), ("d_intrinsic", _P), ("d_campos_nearest", _P), ("d_images", _P), ("d_frame_w", _P)]


RENDER_STAGES = ("query", "plan_gather", "chain_gather", "chain", "mlp_colorfeat", "proj_rows", "mlp_merge", "merge", "mlp_mixup", "final_color",
                 "composite")

# stage boundaries of the two training calls (their stage_events hook: one more event than stages)
TRAIN_FWD_STAGES = ("pack", "query_plan", "featmap", "gather_table", "chain", "per_sample", "composite")
TRAIN_BWD_STAGES = ("zero_pack", "composite_mixup", "merge_mlp", "proj_conv", "colorfeat", "ksum", "block3", "point_sums", "block1")

# name -> (restype, argtypes); must list every symbol include/hnr.h declares (tests check this)
SIGNATURES = {
    "hnr_version": (ctypes.c_char_p, []),
    "hnr_last_error": (ctypes.c_char_p, []),
    "hnr_points_bounds": (_I, [_P, _I, _P, _P]),
    "hnr_grid_build": (_I, [_P, _I, ctypes.POINTER(GridParams), _P, ctypes.POINTER(_P)]),
    "hnr_grid_free": (_I, [_P]),
    "hnr_grid_get_stats": (_I, [_P, ctypes.POINTER(GridStats)]),
    "hnr_grid_get_params": (_I, [_P, ctypes.POINTER(GridParams)]),
    "hnr_grid_export_dense": (_I, [_P, _P, _P, _P, _P]),
    "hnr_grid_grow": (_I, [_P, _P, _I, _P]),
    "hnr_grid_export_runs": (_I, [_P, _P, _P, _P]),
    "hnr_query_work_elems": (ctypes.c_int64, [_I, _I]),
    "hnr_march_query": (_I, [_P, _P, _P, _P, ctypes.POINTER(QueryParams), _P, _P, _P, _P, _P, _P, _P]),
    "hnr_ray_compact_plan": (_I, [_P, _I, _P, _P, _P, _P]),
    "hnr_ray_compact": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hnr_linear_packed_dims": (_I, [_I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "hnr_linear_pack": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "hnr_linear_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "hnr_linear_f32_gather_add": (_I, [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "hnr_linear_f32_side": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "hnr_sample_plan": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "hnr_gather_rows": (_I, [_P] * 5 + [_I] + [_P] * 9 + [_I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P]),
    "hnr_point_rows": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "hnr_gather_points": (_I, [_P, ctypes.c_int64] + [_P] * 5 + [_I] + [_P] * 9 + [_P]),
    "hnr_ksum": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P]),
    "hnr_image_features_scratch_elems": (ctypes.c_int64, [_I, _I, _I]),
    "hnr_image_features": (_I, [_P, _I, _I, _I, ctypes.POINTER(_P), ctypes.POINTER(_P), _F, _P, _P, _P]),
    "hnr_proj_rows": (_I, [_P] * 8 + [_I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P]),
    "hnr_proj_pixels": (_I, [_P] * 5 + [_I, _I, _I, _I, _P, _P]),
    "hnr_div_probe": (_I, [_P, _P, _I, _P, _P, _P]),
    "hnr_div_probe2": (_I, [_P, _P, _F, _I, _P, _P, _P, _P, _P]),
    "hnr_merge": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _I, _P]),
    "hnr_chain_packed_bytes": (ctypes.c_int64, []),
    "hnr_chain_workspace_bytes": (ctypes.c_int64, [_I]),
    "hnr_chain_pack": (_I, [_P, _I] + [_P] * 9 + [_P, _P]),
    "hnr_chain_gather": (_I, [_P] * 11 + [_I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "hnr_chain_classes": (_I, []),
    "hnr_chain_plan": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "hnr_point_records": (_I, [_P, _P, _P, _P, _I, _P, _P]),
    "hnr_chain_gather_rec": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "hnr_chain_forward": (_I, [_P, _P, _I, _P, _P, _I, _F, _P, _I, _P, _P, _I, _P]),
    "hnr_mlp3_packed_bytes": (ctypes.c_int64, [_I, ctypes.POINTER(_I)]),
    "hnr_mlp3_pack": (_I, [_I, ctypes.POINTER(_P), ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_P), _P, _P]),
    "hnr_mlp3_forward": (_I, [_P, _I, ctypes.c_int64, _P, _I, _I, _I, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), _F, _P, _P, _I, _P, _I, _P, _I, _P]),
    "hnr_render_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(RenderParams)]),
    "hnr_render_forward": (_I, [_P, ctypes.POINTER(RenderParams), ctypes.POINTER(RenderCloud), ctypes.POINTER(RenderWeights),
                                ctypes.POINTER(RenderCamera), ctypes.POINTER(RenderViews), _P, ctypes.c_int64, ctypes.POINTER(RenderOutputs), _P]),
    "hnr_merge_stage": (_I, [_P] * 8 + [_I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _F, _P, _I, _P]),
    "hnr_final_color": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "hnr_mixup_stage": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P, _P]),
    "hnr_composite": (_I, [_P] * 8 + [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P]),
    "hnr_probe_outputs": (_I, [_P] * 10 + [_I, _I, _I, _I] + [_P] * 7 + [_P]),
    "hnr_ray_march": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "hnr_blur_select": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "hnr_blur_select_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "hnr_shipped_loss_scratch_bytes": (ctypes.c_int64, []),
    "hnr_shipped_loss": (_I, [_P, _P, _P, _I, _P, ctypes.c_int64, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "hnr_shipped_loss_rows": (_I, [_P, _P, _P, _I, _P, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P]),
    "hnr_shipped_loss_rows_fw": (_I, [_P, _P, _P, _I, _P, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    "hnr_voxel_downsample_scratch_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "hnr_voxel_downsample": (_I, [_P, _I, ctypes.POINTER(_F), _F, _P, _P, _P, _P, _P, _P, ctypes.c_int64, _P]),
    "hnr_blur_gray_patches": (_I, [_P, _P, _I, _I, _P, _P]),
    "hnr_blur_gray_patches_bwd": (_I, [_P, _I, _I, _P, _P]),
    "hnr_blur_apply": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "hnr_blur_apply_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    # backward
    "hnr_composite_bwd": (_I, [_P] * 8 + [_I, _I, _I, _F, _I, _P, _P, _P]),
    "hnr_final_color_bwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "hnr_merge_bwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P]),
    "hnr_proj_rows_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, ctypes.c_int64, _P]),
    "hnr_sort_rows_scratch_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "hnr_sort_rows_by_key": (_I, [_P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P]),
    "hnr_segment_sum_rows": (_I, [_P, _I, _P, _I, _P, _P, ctypes.c_int64, _I, _P, ctypes.c_int64, _P]),
    "hnr_image_features_bwd": (_I, [_P, _I, _I, _I, ctypes.POINTER(_P), _F, _P, _P, ctypes.POINTER(_P), ctypes.POINTER(_P), _P]),
    "hnr_gather_rows_bwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P]),
    "hnr_segment_sum_rows_det": (_I, [_P, _I, _P, _P, ctypes.c_int64, _I, _I, _P, _P, ctypes.c_int64, _I, _P]),
    "hnr_segment_sum_rows_csr": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, ctypes.c_int64, _P, _I, _I, _P, ctypes.c_int64, _P, _P]),
    "hnr_probe_select": (_I, [_P, _P, _P, _P, ctypes.POINTER(_F), _P, _P, _I, _I, _I, _F, _F, _P, _P, _P]),
    # training-step dense layers on the 16-bit matrix pipe (csrc/h2gemm.hip)
    "hnr_h2lin_packed_bytes": (ctypes.c_int64, [_I]),
    "hnr_h2lin_pack": (_I, [_I, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(_I), ctypes.POINTER(_I),
                            ctypes.POINTER(_P), ctypes.POINTER(_P), _P]),
    "hnr_h2lin": (_I, [_P, _I, ctypes.c_int64, _P, _I, ctypes.c_int64, _P, _I, _I, _I, _I, _F, _P, _I, _P, _I, _P, _P]),
    "hnr_h2wgrad_scratch_bytes": (ctypes.c_int64, [_I, _I]),
    "hnr_h2lin_dgrad_bits": (_I, [_P, _I, ctypes.c_int64, _P, _P, _I, _I, _F, _P, _P, _I, _P, _P]),
    "hnr_h2wgrad": (_I, [_P, _I, _P, _I, ctypes.c_int64, _P, _I, ctypes.c_int64, _I, _I, _P, _P, _P, _I, _P, _I, _P, _P]),
    "hnr_absmax": (_I, [_P, _I, ctypes.c_int64, _P, _I, ctypes.c_int64, _I, _P, _P]),
    "hnr_point_grad_pack": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "hnr_point_grad_apply": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    # the training step as two calls (csrc/render_train.hip)
    "hnr_render_train_workspace_bytes": (ctypes.c_int64, [ctypes.POINTER(TrainParams)]),
    "hnr_render_train_debug_layout": (_I, [ctypes.POINTER(TrainParams), ctypes.POINTER(ctypes.c_int64), _I]),
    "hnr_render_train_touched": (_I, [ctypes.POINTER(TrainParams), _P, ctypes.c_int64, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    "hnr_render_train_forward": (_I, [_P, ctypes.POINTER(TrainParams), ctypes.POINTER(TrainCloud), ctypes.POINTER(TrainWeights), ctypes.POINTER(RenderCamera),
                                      ctypes.POINTER(TrainViews), _P, _P, _P, ctypes.c_int64, ctypes.POINTER(RenderOutputs), _P]),
    "hnr_render_train_backward": (_I, [ctypes.POINTER(TrainParams), ctypes.POINTER(TrainCloud), ctypes.POINTER(TrainWeights), ctypes.POINTER(RenderCamera),
                                       ctypes.POINTER(TrainViews), _P, ctypes.c_int64, ctypes.POINTER(RenderOutputs), _P, _P, ctypes.POINTER(TrainCloudGrads),
                                       ctypes.POINTER(TrainWeights), _P]),
}

_lib = None


def lib():
    """Load libhnr_hip.so once.  Raises HnrError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HnrError(
            "libhnr_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hybridneuralrendering_amd/csrc`; there is no CPU or PyTorch fallback." % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HnrError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise HnrError("libhnr_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().hnr_last_error().decode("utf-8", "replace")
        raise HnrError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor as void*."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HnrError("expected a tensor on the GPU, got device=%s" % t.device)
    if not t.is_contiguous():
        raise HnrError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HnrError("%s must be a tensor on the GPU (the HIP path has no CPU fallback)" % name)
    if dtype is not None and t.dtype != dtype:
        raise HnrError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


# ---- raw HIP events (profiling hooks of hnr_render_forward: recorded by the library on the launch stream) ------------------
_hip = None


def hip_runtime():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(_P)]
        _hip.hipEventDestroy.argtypes = [_P]
        _hip.hipEventSynchronize.argtypes = [_P]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(_F), _P, _P]
    return _hip


class StageEvents:
    """len(stages) + 1 hipEvent_t handles; elapsed_ms() after the stream has been synchronised."""

    def __init__(self, stages=RENDER_STAGES):
        H = hip_runtime()
        self.stages = tuple(stages)
        self.n = len(self.stages) + 1
        self.arr = (_P * self.n)()
        for i in range(self.n):
            e = _P()
            if H.hipEventCreate(ctypes.byref(e)) != 0:
                raise HnrError("hipEventCreate failed")
            self.arr[i] = e

    def elapsed_ms(self):
        H = hip_runtime()
        out = {}
        for i, name in enumerate(self.stages):
            ms = _F()
            if H.hipEventElapsedTime(ctypes.byref(ms), self.arr[i], self.arr[i + 1]) != 0:
                raise HnrError("hipEventElapsedTime failed (events not recorded / not complete)")
            out[name] = float(ms.value)
        return out

    def __del__(self):
        try:
            H = hip_runtime()
            for i in range(self.n):
                if self.arr[i]:
                    H.hipEventDestroy(self.arr[i])
        except Exception:
            pass
