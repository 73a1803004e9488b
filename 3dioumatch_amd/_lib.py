"""ctypes binding of lib3dioumatch_hip.so (the C ABI declared in include/*.h).  PyTorch is used by the callers for device memory and streams only.

There is NO CPU fallback: if the shared library is missing this module raises ImportError,
and every device entry point raises RuntimeError for non-GPU tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib3dioumatch_hip%s.so" % os.environ.get("PN2_LIB_SUFFIX", ""))  # (build.py)

_c_int, _c_float, _vp, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPE)
_SIGNATURES = {
    "pn2_furthest_point_sampling": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_furthest_point_sampling_ws": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _sz, _vp],
    "pn2_fps_workspace_bytes": [_c_int, _c_int, _c_int],
    "pn2_gather_points": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_gather_points_grad": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_ball_query": [_c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _vp, _vp, _sz, _vp],
    "pn2_ball_query_workspace_bytes": [_c_int, _c_int, _c_int, _c_int],
    "pn2_group_points": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_group_points_grad": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_three_nn": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate_affine_supported": [_c_int, _c_int, _c_int],
    "pn2_three_interpolate_affine": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate_grad": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_nn_weights": [ctypes.c_longlong, _vp, _vp, _vp],
    "pn2_multi_copy": [_c_int, _vp, ctypes.c_longlong, _vp],
    "pn2_group_inverse_supported": [_c_int, _c_int, _c_int],
    "pn2_group_inverse_entries": [_c_int, _c_int],
    "pn2_group_inverse_build": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp],
    "pn2_group_points_grad_sorted": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp,
                                     _vp, _vp],
    "pn2_three_interpolate_into": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp],
    "pn2_three_interpolate_grad_from": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp],
    "pn2_query_and_group": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _vp, _vp,
                            _vp, _vp, _vp, _vp, _sz, _vp],
    "pn2_group_concat": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _vp, _vp,
                         _vp, _vp, _vp, _vp],
    "pn2_grid_bytes": [_c_int, _c_int],
    "pn2_grid_launch_order": [_c_int, _c_int, _vp, _vp, _vp, _vp],
    "pn2_grid_build": [_c_int, _c_int, _c_float, _vp, _vp, _sz, _vp],
    "pn2_ball_query_prebuilt": [_c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _vp, _vp, _sz, _vp],
    "pn2_query_and_group_prebuilt": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "pn2_query_and_group_picks": [_c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "pn2_fps_grid_supported": [_c_int],
    "pn2_furthest_point_sampling_grid": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _sz, _c_float, _vp,
                                         _sz, _vp],
    "pn2_fps_ties_supported": [_c_int],
    "pn2_furthest_point_sampling_ties": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _sz, _c_float, _vp,
                                         _sz, _vp, _vp],
    "pn2_furthest_point_sampling_prefix": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _sz, _vp, _vp],
    "pn2_error_string": [_c_int],
    "mlp_bn_workspace_floats": [_c_int, _c_int, _c_int],
    "mlp_bn_train_stats": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _c_float, _c_float, _vp, _vp, _vp,
                           _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_eval_coeff": [_c_int, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_relu_apply": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_relu_pool": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_relu_backward": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_relu_pool_backward": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_relu_backward_stats": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_forward": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp],
    "mlp_gemm_forward_stats_parts": [_c_int, _c_int, _c_int, _c_int, _vp],
    "mlp_gemm_forward_stats": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_finalize_pairs": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _c_float, _c_float, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_bn_finalize_pairs_scratch_bytes": [_c_int],
    "mlp_gemm_dgrad": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                       _vp, _vp, _vp, _vp],
    "mlp_gemm_wgrad": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                       _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_dgrad_nt": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp, _vp],
    "mlp_gemm_dgrad_pooled_nt": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_dgrad_pooled": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_wgrad_pooled": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_wgrad_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
    "mlp_gemm_forward_stats_pool_supported": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_gemm_forward_stats_pool": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int,
                                    _vp, _vp, _vp],
    "mlp_pool_gram_supported": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_pool_gram_parts": [_c_int, _c_int],
    "mlp_pool_gram_workspace_floats": [_c_int, _c_int],
    "mlp_pool_gram_backward": [_c_int, _c_int, _c_int] + [_vp] * 19,
    "mlp_pool_gram256_supported": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_pool_gram256_parts": [_c_int, _c_int],
    "mlp_pool_gram256_workspace_floats": [_c_int, _c_int, _c_int],
    "mlp_pool_gram256_backward": [_c_int, _c_int, _c_int] + [_vp] * 19,
    "mlp_chain_lin4_parts": [_c_int, _c_int, _c_int, _c_int, _vp],
    "mlp_chain_lin4_image_bytes": [],
    "mlp_chain_lin4_prepare": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_chain_lin4_stats": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "mlp_chain_lin4_forward": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_chain_finalize": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _c_float, _c_float, _vp, _vp, _vp, _vp, _vp,
                           _vp, _vp],
    "mlp_bn_pool_from_extrema": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_wgrad_first4_workspace_bytes": [_c_int, _c_int],
    "mlp_wgrad_first4": [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_backward_fused_lin4_gated": [],
    "mlp_wgrad_first4_from_gated": [_c_int] + [_vp] * 9,
    "mlp_first4_moments_doubles": [],
    "mlp_first4_moments": [_c_int, _c_int, _vp, _vp, _vp],
    "mlp_first4_bn": [_vp, ctypes.c_double, _vp, _vp, _vp, _c_float, _c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                      _vp],
    "mlp_gemm_forward_stats_lin4": [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_backward_fused_supported": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_gemm_backward_fused_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
    "mlp_gemm_backward_fused": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _c_int,
                                _vp, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_backward_fused_stats_parts": [_c_int, _c_int, _c_int, _c_int],
    "mlp_bn_backward_finalize": [_c_int, _c_int, ctypes.c_double, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp],
    "mlp_gemm_backward_small_supported": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_weight_image_elems": [_c_int, _c_int],
    "mlp_weight_images_build": [_c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_image_supported": [_c_int, _c_int],
    "mlp_gemm_forward_img": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp],
    "mlp_gemm_backward_small_img": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_gemm_backward_small": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mlp_pregather_supported": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "mlp_pregather_pack": [_c_int, _c_int, _c_int, _c_int, ctypes.c_float, _vp, _vp, _vp, _vp, _vp],
    "mlp_pregather_unpack_grad": [_c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp],
    "mlp_pregather_forward": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "mlp_pregather_backward": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp],
    "mlp_defer_weight_reductions": [_c_int],
    "mlp_flush_weight_reductions": [],
    "lhs_nms3d_aabb": [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _c_int, _c_int, _vp,
                       _vp],
    "votenet_decode_scores": [_c_int] * 5 + [_vp] * 13,
    "votenet_decode_scores_grad": [_c_int] * 5 + [_vp] * 13,
    "lhs_pseudo_select": [_vp, _vp],
    "lhs_pseudo_finish": [_vp, _vp],
    "lhs_nms_samecls": [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _c_int, _vp, _vp],
    "iou3d_boxes_overlap_bev": [_c_int, _vp, _c_int, _vp, _vp, _vp],
    "iou3d_boxes_iou_bev": [_c_int, _vp, _c_int, _vp, _vp, _vp],
    "iou3d_boxes_iou3d": [_c_int, _vp, _c_int, _vp, _vp, _vp],
    "iou3d_scene_best_iou3d": [_c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _vp],
    "votenet_loss_decode": [_vp, _vp],
    "votenet_loss_forward_backward": [_vp, _vp],
    "votenet_loss_scratch_floats": [_vp],
    "votenet_bbox_jitter": [_c_int, _c_int, _c_int, _c_int] + [_vp] * 15,
    "votenet_gridconv_points": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "votenet_channel_normalize": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp],
    "votenet_channel_normalize_grad": [_c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp],
    "votenet_adam_step": [ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double,
                          ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp, _vp],
    "iou3d_corners_iou3d": [_c_int, _vp, _c_int, _vp, _vp, _vp],
    "iou3d_corners_best_match": [_c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "iou3d_nms_mask": [_vp, _vp, _c_int, _c_float, _vp],
    "iou3d_nms_normal_mask": [_vp, _vp, _c_int, _c_float, _vp],
    "iou3d_nms": [_vp, _c_int, _c_float, _c_int, _vp, _vp, _vp, _vp],
    "iou3d_boxes_iou_bev_cpu": [_c_int, _vp, _c_int, _vp, _vp],
}
_RESTYPE = {"pn2_ball_query_workspace_bytes": _sz, "pn2_grid_bytes": _sz, "pn2_fps_workspace_bytes": _sz, "mlp_bn_workspace_floats": _sz, "mlp_gemm_wgrad_workspace_floats": _sz, "mlp_wgrad_first4_workspace_bytes": _sz, "mlp_gemm_backward_fused_workspace_floats": _sz, "mlp_bn_finalize_pairs_scratch_bytes": _sz, "mlp_chain_lin4_image_bytes": _sz, "mlp_weight_image_elems": _sz, "mlp_pool_gram_workspace_floats": _sz, "mlp_pool_gram256_workspace_floats": _sz, "pn2_error_string": ctypes.c_char_p}

EXPORTS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "3dioumatch_amd: %s is missing -- build it with `python 3dioumatch_amd/build.py` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime; load it FIRST so that this library binds to the
    # runtime that owns the process's device context (dlopen-ing the library before `import torch`
    # pulled in /opt/rocm's copy: two runtimes in one process, "no ROCm-capable device" at the
    # first launch -- seen with __graft_entry__.build() followed by smoke() in one interpreter)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI mismatch, fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, _c_int)
    return lib


lib = _load()


def check(code, what):
    """Turn a non-zero hipError_t return into a Python exception (never exit())."""
    if code != 0:
        msg = lib.pn2_error_string(int(code))
        raise RuntimeError("%s failed: HIP error %d (%s)" %
                           (what, code, msg.decode() if msg else "?"))


def current_stream_ptr(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
