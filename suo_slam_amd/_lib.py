"""ctypes binding of libsuo_hip.so (include/suo_hip.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUO_HIP_LIB") or os.path.join(HERE, "libsuo_hip.so")      # SUO_HIP_LIB: kernel-tuning variants only

_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int)
c_i64p = C.POINTER(C.c_int64)
VP = C.c_void_p


class SuoError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/suo_hip.h declares
SIGNATURES = {
    "suo_last_error": (C.c_char_p, []),
    "suo_version": (C.c_int, []),
    "suo_device_count": (C.c_int, []),
    "suo_net_create": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(VP), C.POINTER(VP), c_i32p, C.c_int, C.POINTER(VP)]),
    "suo_net_destroy": (None, [VP]),
    "suo_net_set_graph": (C.c_int, [VP, C.c_int]),
    "suo_net_prepare": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "suo_net_workspace_bytes": (C.c_size_t, [VP]),
    "suo_net_schedule_bytes": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "suo_net_forward": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP]),
    "suo_net_forward_frames": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP]),
    "suo_net_forward_prior_kp": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_render_priors": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "suo_net_backbone": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "suo_upload": (C.c_int, [VP, VP, C.c_size_t, VP]),
    "suo_decode_heatmaps": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP, VP]),
    "suo_classifier": (C.c_int, [VP, VP, VP, C.c_int, VP, VP, VP]),
    "suo_keypoint_masks": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_float, C.c_float, VP, VP]),
    "suo_roi_align_concat": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, C.c_int, VP, VP, VP]),
    "suo_pack_gemm_weight": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_pack_conv_weight": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_conv1x1": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_pack_gemm_weight_bf16x3": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "suo_conv1x1_bf16x3": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_conv1x1_bf16x3_pool": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "suo_conv1x1_bf16x3_ex": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_conv_kxk": (C.c_int, [C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP]),
    "suo_net_get_pipe": (C.c_int, [VP]),
    "suo_net_set_pipe": (C.c_int, [VP, C.c_int]),
    "suo_net_range_exceeded": (C.c_int, [VP]),
    "suo_pack_gemm_weight_f16x2": (C.c_int, [VP, C.c_int, C.c_int, VP, VP]),
    "suo_conv1x1_f16x2_ex": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "suo_conv1x1_f16x2_pool": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP]),
    "suo_pack_wino_weight_f16x2": (C.c_int, [VP, C.c_int, C.c_int, VP, VP]),
    "suo_conv3x3_wino_f16x2_n": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP]),
    "suo_pack_tail_weight_f16x2": (C.c_int, [VP, C.c_int, C.c_int, VP, VP]),
    "suo_conv3x3_wino_f16x2_conv1x1_skip_up": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_conv3x3_wino_f16x2_conv1x1_skip_up_next": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_pack_wino_weight_bf16x3": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "suo_conv3x3_wino_x3": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP]),
    "suo_conv3x3_wino_x3_n": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP]),
    "suo_pack_tail_weight_bf16x3": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "suo_conv3x3_wino_x3_conv1x1_skip_up": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, VP, VP]),
    "suo_pack_wino_weight": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_conv1x1_pool": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "suo_conv3x3_wino": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP]),
    "suo_conv3x3_wino_conv1x1_skip": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP]),
    "suo_conv3x3_wino_conv1x1_skip_up": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_conv3x3_conv1x1_skip": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP]),
    "suo_pack_res_block": (C.c_int, [VP, VP, VP, VP, VP, VP, VP]),
    "suo_res_block": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_pack_res_block_bf16x3": (C.c_int, [VP, VP, VP, VP, VP, VP, VP]),
    "suo_res_block_bf16x3": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_pack_res_block_f16x2": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_res_block_f16x2": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_conv1x1_chain_head_f16x2": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, VP, VP, VP, VP, C.c_int, C.c_int, VP, VP]),
    "suo_pack_stem_weight_bf16x3": (C.c_int, [VP, C.c_int, VP, VP]),
    "suo_pack_stem_weight_f16x2": (C.c_int, [VP, C.c_int, VP, VP, VP]),
    "suo_stem_f16x2": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP]),
    "suo_stem_f16x2_next": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_stem_x3": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, VP, VP, VP, VP]),
    "suo_maxpool2": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_upsample2_add": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "suo_pnp_batch": (C.c_int, [C.c_int, VP, VP, VP, C.c_double, C.c_uint64, C.c_int, VP, VP, VP, VP]),
    "suo_pnp_replay": (C.c_int, [C.c_int, VP, VP, VP, C.c_double, VP, C.c_int, C.c_int, VP, VP, VP, VP, VP]),
    "suo_pnp": (C.c_int, [VP, VP, C.c_int, C.c_double, VP]),
    "suo_optimize": (C.c_int, [VP]),
    "suo_optimize_batch": (C.c_int, [VP, C.c_int]),
    "suo_ba_ctx_create": (C.c_int, [VP, C.POINTER(VP)]),
    "suo_ba_ctx_destroy": (None, [VP]),
    "suo_ba_ctx_ns": (C.c_int, [VP]),
    "suo_ba_classify": (C.c_int, [VP, C.c_int, VP]),
    "suo_ba_linearize": (C.c_int, [VP, C.c_int, VP]),
    "suo_ba_schur": (C.c_int, [VP, C.c_double, VP]),
    "suo_ba_solve_update": (C.c_int, [VP, C.c_double, C.c_int, VP, VP]),
    "suo_ba_restore": (C.c_int, [VP]),
    "suo_ba_ctx_download": (C.c_int, [VP, VP]),
    "suo_ba_classify_dev": (C.c_int, [VP, C.c_int, VP, VP]),
    "suo_ba_linearize_dev": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP]),
    "suo_ba_schur_dev": (C.c_int, [VP, C.c_double, VP, VP]),
    "suo_ba_solve_update_dev": (C.c_int, [VP, C.c_double, C.c_int, C.c_int, VP, VP, VP, VP]),
    "suo_ba_restore_dev": (C.c_int, [VP, VP]),
    "suo_ba_lm_begin_dev": (C.c_int, [VP, VP, C.c_int, C.c_int, VP]),
    "suo_ba_lm_linearize_dev": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP]),
    "suo_ba_lm_schur_dev": (C.c_int, [VP, VP, VP, VP, VP]),
    "suo_ba_lm_solve_update_dev": (C.c_int, [VP, C.c_int, C.c_int, VP, VP, VP, VP, VP]),
    "suo_ba_lm_unit_one_rank_dev": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP, VP]),
    "suo_ba_lm_decide_dev": (C.c_int, [VP, VP, VP, VP]),
    "suo_debug_ba_jacobians": (C.c_int, [VP, C.c_int, VP, VP]),
    "suo_debug_cholesky_solve": (C.c_int, [VP, VP, C.c_int, VP, VP]),
    "suo_frame_geom_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(VP)]),
    "suo_frame_geom_destroy": (None, [VP]),
    "suo_frame_geom_launch": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "suo_frame_geom_fetch": (C.c_int, [VP, VP]),
    "suo_frame_geom_ready": (C.c_int, [VP]),
    "suo_frame_geom_device_result": (C.c_int, [VP, VP]),
    "suo_slam_vote": (C.c_int, [C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, VP, VP, C.c_int, C.c_double, C.c_double, C.c_int, VP, VP, VP, VP]),
    "suo_mesh_db_create": (C.c_int, [C.c_int, VP, VP, C.POINTER(VP)]),
    "suo_mesh_db_destroy": (None, [VP]),
    "suo_pose_errors": (C.c_int, [VP, C.c_int, VP, VP, VP, VP, VP]),
    "suo_slam_store_create": (VP, [C.c_int]),
    "suo_slam_store_destroy": (None, [VP]),
    "suo_slam_store_put": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "suo_slam_score": (C.c_int, [VP, C.c_int, VP, C.c_double, C.c_double, VP]),
}


class BaProblem(C.Structure):
    """ctypes mirror of suo_ba_problem (include/suo_hip.h)."""
    _fields_ = [("n_cam", C.c_int), ("n_obj", C.c_int), ("n_edge", C.c_int),
                ("cam_T", VP), ("cam_fixed", VP), ("obj_T", VP), ("obj_fixed", VP),
                ("edge_cam", VP), ("edge_obj", VP), ("edge_camk", VP), ("edge_p", VP), ("edge_uv", VP), ("edge_info", VP),
                ("edge_inlier", VP), ("edge_chi2", VP),
                ("its", C.c_int * 8), ("n_rounds", C.c_int), ("init_with_outliers", C.c_int),
                ("chi2_thr", C.c_double), ("huber_delta", C.c_double), ("stats", C.c_int * 4)]


class FrameGeomParams(C.Structure):
    """ctypes mirror of suo_frame_geom_params."""
    _fields_ = [("pnp_threshold", C.c_double), ("seed", C.c_uint64), ("use_cov", C.c_int), ("do_lm", C.c_int), ("its", C.c_int * 4),
                ("n_rounds", C.c_int), ("chi2_thr", C.c_double), ("huber_delta", C.c_double), ("seed_dev", VP)]


class FrameGeomResult(C.Structure):
    """ctypes mirror of suo_frame_geom_result."""
    _fields_ = [("n_frames", C.c_int), ("n_crops", C.c_int), ("T_pnp", VP), ("T_opt", VP), ("chi2", VP), ("pnp_status", VP),
                ("pnp_best_inliers", VP), ("pnp_iterations", VP), ("n_kp", VP), ("lm_stats", VP), ("accepted", VP), ("inlier", VP),
                ("uv", VP), ("cov", VP), ("mask", VP)]


def register(extra):
    SIGNATURES.update(extra)


def lib():
    """Load the HIP extension; raise if it is missing (the product path has no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SuoError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64; load it FIRST so that this library binds to the same HIP
        # runtime (two runtimes in one process cannot share the device: torch would then see "No HIP GPUs").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().suo_last_error().decode("utf-8", "replace")
        raise SuoError(f"{what} failed (code {rc}): {msg}")


def require_gpu():
    n = lib().suo_device_count()
    if n <= 0:
        raise SuoError("no HIP device visible: the suo_slam_amd product path needs an MI355X (no CPU fallback)")
    return n


def current_stream_ptr():
    """torch's current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Stream object behind three device-index
    look-ups (13 us on the GPU boxes, eight to ten times per SLAM view on the host's critical path); the raw query underneath it is a microsecond."""
    import torch
    try:
        return int(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:                                  # (a torch without the private entry points: the public route)
        return int(torch.cuda.current_stream().cuda_stream)
