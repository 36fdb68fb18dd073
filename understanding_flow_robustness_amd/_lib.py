"""ctypes binding of lib/libufr_hip.so (C ABI declared in include/ufr_hip.h).

The product path has NO CPU fallback: if the shared library is missing, cannot be
loaded, or a tensor is not a HIP tensor, every operator raises.  Only device
pointers, sizes and the current HIP stream cross this boundary.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UFR_HIP_LIB") or os.path.join(_HERE, "lib", "libufr_hip.so")   # UFR_HIP_LIB: another BUILD of the same library (same-box A/B of two kernels)
UFR_F32, UFR_F64, UFR_F16 = 0, 1, 2
ABI_VERSION = 9            # UFR_ABI_VERSION of include/ufr_hip.h these ctypes mirrors were written against
_lib = None


class CorrParams(C.Structure):
    """ufr_corr_params (include/ufr_hip.h) -- the 12 ints of correlation_sampler.cpp:59-87."""
    _fields_ = [(n, C.c_int) for n in (
        "kH", "kW", "patchH", "patchW", "padH", "padW", "dilationH", "dilationW",
        "dilation_patchH", "dilation_patchW", "dH", "dW")]


UFR_MAX_LEVELS = 8


class Pyramid(C.Structure):
    """ufr_pyramid (include/ufr_hip.h)."""
    _fields_ = [("num_levels", C.c_int),
                ("vol", C.c_void_p * UFR_MAX_LEVELS),
                ("grad_vol", C.c_void_p * UFR_MAX_LEVELS),
                ("Hl", C.c_int * UFR_MAX_LEVELS),
                ("Wl", C.c_int * UFR_MAX_LEVELS)]


class AltCorrLevels(C.Structure):
    """ufr_altcorr_levels (include/ufr_hip.h)."""
    _fields_ = [("num_levels", C.c_int), ("fmap2", C.c_void_p * 4), ("fmap2_grad", C.c_void_p * 4),
                ("H2", C.c_int * 4), ("W2", C.c_int * 4), ("coord_scale", C.c_float * 4)]


class AltCorrPlaneLevels(C.Structure):
    """ufr_altcorr_plane_levels (include/ufr_hip.h)."""
    _fields_ = [("num_levels", C.c_int), ("planes", C.c_void_p * 4), ("plane_stride", C.c_long * 4),
                ("H2", C.c_int * 4), ("W2", C.c_int * 4), ("coord_scale", C.c_float * 4)]


UFR_IGEMM_MAX_TAPS = 25


class IgemmPhase(C.Structure):
    """ufr_igemm_phase (include/ufr_hip.h)."""
    _fields_ = [("ntaps", C.c_int), ("oy0", C.c_int), ("ox0", C.c_int), ("w_off", C.c_long),
                ("dy", C.c_byte * UFR_IGEMM_MAX_TAPS), ("dx", C.c_byte * UFR_IGEMM_MAX_TAPS)]


class IgemmDesc(C.Structure):
    """ufr_igemm_desc (include/ufr_hip.h), field for field."""
    _fields_ = [("x", C.c_void_p), ("x_plane_stride", C.c_long), ("in_chunk0", C.c_int), ("KC", C.c_int),
                ("B", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("in_sy", C.c_int), ("in_sx", C.c_int),
                ("in_x0", C.c_void_p), ("in_x0_stride", C.c_int), ("in_x0_div", C.c_int), ("in_xw", C.c_int),
                ("w", C.c_void_p), ("w_plane_stride", C.c_long), ("Npad", C.c_int), ("N", C.c_int),
                ("Hr", C.c_int), ("Wr", C.c_int),
                ("row_x0", C.c_void_p), ("row_x0_stride", C.c_int), ("row_x0_div", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("out_sy", C.c_int), ("out_sx", C.c_int),
                ("nphase", C.c_int),
                ("phase", IgemmPhase * 4),
                ("act", C.c_int), ("bias", C.c_void_p), ("slope", C.c_float),
                ("add", C.c_void_p), ("add_chunk0", C.c_int),
                ("mask", C.c_void_p), ("mask_chunk0", C.c_int),
                ("out_planes", C.c_void_p), ("out_plane_stride", C.c_long), ("out_chunk0", C.c_int),
                ("out_f32", C.c_void_p), ("out_f32_chunk0", C.c_int),
                ("tail", C.c_void_p), ("tail_n0", C.c_int), ("tail_accumulate", C.c_int),
                ("splitk", C.c_int), ("ws", C.c_void_p),
                ("products", C.c_int), ("variant", C.c_int), ("k_order", C.c_int),
                ("out_rowmajor", C.c_void_p), ("out_ld", C.c_long),
                ("planes_chunks", C.c_int), ("f32_first_chunk", C.c_int), ("no_reduce", C.c_int),
                ("tickets", C.c_void_p)]


UFR_MAX_CONE_LAYERS = 8


class ConeChain(C.Structure):
    """ufr_cone_chain (include/ufr_hip.h)."""
    _fields_ = [("n_layers", C.c_int),
                ("kernel", C.c_int * UFR_MAX_CONE_LAYERS), ("stride", C.c_int * UFR_MAX_CONE_LAYERS),
                ("pad", C.c_int * UFR_MAX_CONE_LAYERS),
                ("n_taps", C.c_int),
                ("tap_layer", C.c_int * UFR_MAX_CONE_LAYERS), ("tap_margin", C.c_int * UFR_MAX_CONE_LAYERS)]


_vp, _i, _f, _l, _d = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_double
# name -> argtypes; every function returns int.  Kept in one table so tests can check that each
# symbol declared in include/ufr_hip.h is exported by the built library.
LOSS_PARTIALS = 1024     # UFR_LOSS_PARTIALS (include/ufr_hip.h)
SIGNATURES = {
    "ufr_corr_forward": [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(CorrParams), _vp],
    "ufr_corr_forward_fused": [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(CorrParams), _f, _f, _vp],
    "ufr_corr_backward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(CorrParams), _vp],
    "ufr_altcorr_forward": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_altcorr_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_corr_lookup_forward": [C.POINTER(Pyramid), _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_corr_lookup_backward": [C.POINTER(Pyramid), _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_resample2d_forward": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_resample2d_backward_owner": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_resample2d_backward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_channelnorm_forward": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ufr_channelnorm_backward": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ufr_patch_paste": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _l, _i, _f, _f, _vp],
    "ufr_patch_update": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _l, _f, _f, _f, _f, _vp, _vp],
    "ufr_attack_gate": [_vp, _vp, _f, _vp],
    "ufr_gru_gates_forward": [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _vp],
    "ufr_gru_gates_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _l, _vp],
    "ufr_gru_blend_forward": [_vp, _vp, _vp, _vp, _l, _vp],
    "ufr_gru_blend_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp],
    "ufr_flow_loss_ex": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "ufr_universal_update": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _i, _i, _i, _i,
                             _i, _vp],
    "ufr_flow_loss": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp],
    "ufr_flow2_upsampled_loss": [_vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp],
    "ufr_patch_grad_crop": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_patch_apply": [_vp, _i, _vp, _vp, _i, _i, _f, _f, _vp, _vp],
    "ufr_patch_grad_crop_window": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_patch_paste_placed_rect": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "ufr_patch_paste_placed": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "ufr_cone_window": [_vp, _i, _l, _i, _i, _i, C.POINTER(ConeChain), _i, _i, _vp, _vp, _vp],
    "ufr_window_gather": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_window_scatter": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_conv1_pack_planes": [_vp, _vp, _vp, _l, _i, _i, _i, _i, _vp, _vp],
    "ufr_conv1_unpack_grad": [_vp, _vp, _i, _i, _i, _vp],
    "ufr_nchw_grad_to_planes": [_vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _f, _vp],
    "ufr_window_gather_chunks": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_normalize_frames": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ufr_corr_backward_window_fused": [_vp, _vp, _vp, _i, _f, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "ufr_corr_backward_window": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp],
    "ufr_bias_leaky_forward": [_vp, _vp, _i, _i, _l, _f, _vp],
    "ufr_leaky_backward": [_vp, _vp, _vp, _l, _f, _vp],
    "ufr_conv3x3_c2_forward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_conv3x3_c2_backward_data": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_deconv4x4s2_c2_forward": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_deconv4x4s2_c2_backward_data": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_pwc_warp_forward": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_pwc_warp_backward": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_pwc_warp_backward_owner": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_raft_normalize_pair": [_vp, _vp, _vp, _l, _vp],
    "ufr_raft_normalize_pair_backward": [_vp, _vp, _vp, _l, _vp],
    "ufr_raft_fmap_pyramid_forward": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ufr_raft_fmap_pyramid_backward": [_vp, _i, _vp, _i, _i, _i, _i, _vp],
    "ufr_igemm": [C.POINTER(IgemmDesc), _vp],
    "ufr_flow_upscale4_forward": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "ufr_flow_upscale4_backward": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "ufr_fn2_stage_pack": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "ufr_fn2_stage_unpack_grad": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_fn2_stage_finish_grad": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "ufr_fn2_fusion_pack": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_fn2_fusion_unpack_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_fn2_fusion_finish_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_rowmajor_to_planes": [_vp, _l, _l, _i, _f, _vp, _l, _i, _l, _vp],
    "ufr_nchw_to_planes": [_vp, _vp, _l, _i, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "ufr_window_scatter_planes": [_vp, _vp, _l, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_chunks_to_nchw": [_vp, _l, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _vp],
    "ufr_nchw_cat_to_planes": [_vp, _vp, _vp, _i, _vp, _l, _i, _i, _i, _i, _i, _vp],
    "ufr_chunks_to_nchw_cat": [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _f, _f, _i, _i, _i, _vp],
    "ufr_grad_finalize": [_vp, _i, _vp, _i, _vp, _l, _i, _l, _i, _f, _vp],
    "ufr_flow_head_planes_forward": [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "ufr_flow_head_planes_forward_mfma": [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "ufr_deconv_flow_tail_backward_mfma": [_vp, _l, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "ufr_flow_head_planes_backward": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_flow_head_planes_backward_finalize": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _l, _i, _i, _f, _vp],
    "ufr_unshuffle_pack_planes": [_vp, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_unshuffle_unpack_grad": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_conv3x3s2_c3_planes": [_vp, _vp, _vp, _f, _vp, _l, _i, _i, _i, _i, _i, _vp],
    "ufr_conv3x3_c16_planes": [_vp, _l, _i, _vp, _vp, _f, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_cm_norm_stats": [_vp, _vp, _vp, _l, _i, _i, _f, _vp],
    "ufr_cm_norm_apply": [_vp, _vp, _vp, _l, _i, _vp, _l, _i, _l, _i, _i, _i, _i, _vp],
    "ufr_cm_norm_stats_apply": [_vp, _vp, _vp, _f, _vp, _l, _i, _vp, _l, _i, _l, _i, _i, _i, _i, _vp],
    "ufr_cm_norm_backward": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _l, _i, _l, _i, _i, _i, _vp],
    "ufr_cm_masked_copy": [_vp, _vp, _l, _vp, _l, _vp],
    "ufr_raft_flow_patches": [_vp, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_raft_motion_finish_slabs": [_vp, _i, _i, _i, _vp, _f, _vp, _l, _vp, _l, _i, _vp, _i, _i, _i, _vp],
    "ufr_raft_motion_finish": [_vp, _l, _vp, _l, _i, _vp, _i, _i, _i, _vp],
    "ufr_raft_coords_step": [_vp, _vp, _vp, _vp, _vp, _l, _vp],
    "ufr_grad_finalize_consume": [_vp, _vp, _vp, _l, _l, _i, _f, _vp],
    "ufr_gru_gates_cm_forward": [_vp, _vp, _l, _i, _vp, _l, _i, _l, _i, _vp],
    "ufr_gru_gates_cm_forward_slabs": [_vp, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _i, _l, _i, _vp],
    "ufr_gru_blend_cm_forward_slabs": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _i, _l, _i, _vp],
    "ufr_gru_blend_cm_forward": [_vp, _vp, _vp, _l, _i, _vp, _l, _i, _l, _i, _vp],
    "ufr_gru_blend_cm_backward": [_vp, _vp, _vp, _l, _i, _vp, _vp, _l, _i, _vp, _vp, _l, _i, _vp, _vp],
    "ufr_gru_gates_cm_backward": [_vp, _vp, _l, _i, _vp, _vp, _vp, _l, _i, _vp, _l, _i, _i, _vp, _vp],
    "ufr_altcorr_pyramid_forward": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "ufr_altcorr_pyramid_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ufr_altcorr_pyramid_backward_cm": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "ufr_upfeat_planes_forward_mfma": [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "ufr_upfeat_planes_backward": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_flow_up_planes_forward": [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp],
    "ufr_flow_up_planes_backward": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_corr_forward_planes_window": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _i, _vp],
    "ufr_corr_forward_planes": [_vp, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp],
    "ufr_convex_upsample_forward": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_convex_upsample_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ufr_affine_resample_f64": [_vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _d, _d, _d, _i, _vp],
    "ufr_patch_place": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ufr_patch_crop_f64": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ufr_resample_u8_horizontal": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "ufr_resample_u8_vertical": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "ufr_u8_to_tensor": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "ufr_kitti_flow_decode": [_vp, _vp, _i, _i, _vp],
    "ufr_host_png_unfilter": [_vp, _vp, _i, _i, _i],
    "ufr_igemm_clock_probe": [_vp, _i],
    "ufr_altcorr_planes_prepare": [_vp, _vp, _l, _l, _i, _vp],
    "ufr_altcorr_planes_forward": [_vp, _l, C.POINTER(AltCorrPlaneLevels), _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "ufr_conv1_direct": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _l, _i, _vp],
}
PLAIN = {"ufr_abi_version": (C.c_int, []), "ufr_last_error": (C.c_char_p, []),
         "ufr_device_count": (C.c_int, []), "ufr_build_manifest": (C.c_char_p, []),
         "ufr_igemm_variant_fallbacks": (C.c_int, []),
         "ufr_conv3x3_c2_workspace_floats": (C.c_long, [_i, _i, _i, _i]),
         "ufr_cm_norm_workspace_doubles": (C.c_long, [_l, _i, _i]),
         "ufr_resample2d_backward_workspace_bytes": (C.c_long, [_i, _i, _i]),
         "ufr_pwc_warp_backward_workspace_bytes": (C.c_long, [_i, _i, _i]),
         "ufr_altcorr_pyramid_workspace_bytes": (C.c_long, [_i, _i, _i, _i, _i, _i])}


def lib():
    """Load libufr_hip.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C understanding_flow_robustness_amd/csrc` (hipcc, gfx950). "
                "There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = args, C.c_int
        for name, (res, args) in PLAIN.items():
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = args, res
        if handle.ufr_abi_version() != ABI_VERSION:      # a stale .so with another ufr_igemm_desc layout would read garbage
            raise RuntimeError(f"{LIB_PATH} has ABI version {handle.ufr_abi_version()}, these bindings are written for "
                               f"{ABI_VERSION}: rebuild it (`make -C understanding_flow_robustness_amd/csrc`)")
        verify_build(handle.ufr_build_manifest().decode("ascii", "replace"))
        _lib = handle
    return _lib


_CSRC = os.path.join(_HERE, "csrc")
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "ufr_hip.h")


def source_checksums(csrc: str = _CSRC, header: str = _HEADER) -> dict:
    """{translation unit: md5 of csrc/<unit>.hip + csrc/ufr_common.h + include/ufr_hip.h}: what csrc/Makefile embeds in each object."""
    import hashlib
    with open(os.path.join(csrc, "ufr_common.h"), "rb") as f:
        shared = f.read()
    with open(header, "rb") as f:
        shared += f.read()
    sums = {}
    for name in sorted(os.listdir(csrc)):
        if name.endswith(".hip"):
            with open(os.path.join(csrc, name), "rb") as f:
                sums[name[:-4]] = hashlib.md5(f.read() + shared).hexdigest()
    return sums


def verify_build(manifest: str, csrc: str = _CSRC, header: str = _HEADER) -> None:
    """Refuse a library whose objects were not built from THIS tree's sources (`ufr_build_manifest`, include/ufr_hip.h): every
    object embeds the checksum of what it was compiled from.  A stale object, a missing one or one of a deleted source raises and
    is named.  Skipped when the sources are not there (an installed library without its tree), and when `UFR_HIP_LIB` points at
    another build on purpose (same-box A/B of two kernels) unless UFR_HIP_LIB_VERIFY=1."""
    if os.environ.get("UFR_HIP_LIB") and os.environ.get("UFR_HIP_LIB_VERIFY") != "1":
        return
    if not (os.path.isdir(csrc) and os.path.exists(header) and os.path.exists(os.path.join(csrc, "ufr_common.h"))):
        return
    built = dict(ln.split() for ln in manifest.splitlines() if ln.strip())
    want = source_checksums(csrc, header)
    stale = sorted(n for n in want if n in built and built[n] != want[n])
    missing = sorted(n for n in want if n not in built)
    extra = sorted(n for n in built if n not in want)
    if stale or missing or extra:
        what = "; ".join(s for s in (stale and f"built from other sources: {', '.join(stale)}",
                                     missing and f"not in the library: {', '.join(missing)}",
                                     extra and f"objects of sources that no longer exist: {', '.join(extra)}") if s)
        raise RuntimeError(f"{LIB_PATH} does not match the sources under {csrc} ({what}): rebuild it "
                           "(`make -C understanding_flow_robustness_amd/csrc`) before running anything on it")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ufr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg} (ufr error {rc})" if what else f"{msg} (ufr error {rc})")


def require_hip(t: torch.Tensor, name: str, contiguous: bool = True) -> None:
    """The reference's CHECK_INPUT (correlation_sampler.cpp:31-34) minus its CPU branch."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (HIP device tensor); the MI355X build has no CPU path")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    """The current torch stream's hipStream_t, so launches order with torch ops and get captured."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return UFR_F32
    if t.dtype == torch.float64:
        return UFR_F64
    if t.dtype == torch.float16:
        return UFR_F16
    raise RuntimeError(f"unsupported dtype {t.dtype}: the gfx950 build implements float32, float64 and (correlation only) float16")


# ---------------------------------------------------------------------------------------------------------------------
# When a forward leaves the hand-written engines (its convolutions then run as torch operators on the vendor library) it says
# so: one RuntimeWarning per (module class, reason) and a process-wide counter.  The engines serve the attack's configuration
# (frozen parameters, eval mode, HIP float32); everything else -- a training-mode module, parameters that want weight
# gradients -- is the reference's own torch spelling.  Frame sides: FlowNetC / PWC-Net / FlowNet2 need multiples of 64 IN THE
# REFERENCE TOO (their `torch.cat` of decoder and encoder features raises otherwise, models/FlowNetC.py:167-182,
# models/PWCNet.py:300-360), so there is no silent size fallback to report for them; RAFT needs multiples of 8 like the reference.
VENDOR_FALLBACKS: dict = {}
_SHAPE_PROBE = [0]


class shape_probe:
    """Around a forward that only asks for output SHAPES on a toy frame (PatchAttackStep._setup_cone): the engines decline the
    toy size by design, which is not a fallback of the workload, so nothing is reported or counted inside."""
    def __enter__(self):
        _SHAPE_PROBE[0] += 1

    def __exit__(self, *exc):
        _SHAPE_PROBE[0] -= 1


def engine_refusal(module, x, multiple: int, spatial_scale: int = 1):
    """None when the native engines serve `module` on tensor `x` (a feature map at 1/spatial_scale of the frame whose sides
    must be multiples of `multiple`), else the reason they do not."""
    import os
    if os.environ.get("UFR_ENGINE", "1") != "1":
        return "UFR_ENGINE=0"
    if not (x.is_cuda and x.dtype == torch.float32):
        return "not a HIP float32 tensor"
    if module.training:
        return "module in training mode"
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return "parameters require gradients (the engines compute data gradients only)"
    m = max(1, multiple // spatial_scale)
    if x.shape[2] % m or x.shape[3] % m:
        return f"frame sides are not multiples of {multiple}"
    return None


def engine_gate(module, x, multiple: int, spatial_scale: int = 1, extra=None) -> bool:
    """True = run on the engines.  False = the torch / vendor spelling, reported once per (class, reason) and counted in
    VENDOR_FALLBACKS (explicit opt-outs -- UFR_ENGINE=0, CPU tensors -- are not reported)."""
    reason = extra or engine_refusal(module, x, multiple, spatial_scale)
    if reason is None:
        return True
    if reason not in ("UFR_ENGINE=0", "not a HIP float32 tensor") and not _SHAPE_PROBE[0]:
        key = (type(module).__name__, reason)
        VENDOR_FALLBACKS[key] = VENDOR_FALLBACKS.get(key, 0) + 1
        if VENDOR_FALLBACKS[key] == 1:
            import warnings
            warnings.warn(f"understanding_flow_robustness_amd: {key[0]} runs its convolutions as torch operators on the vendor "
                          f"library, not on the hand-written engines: {reason} (reported once; _lib.VENDOR_FALLBACKS counts)",
                          RuntimeWarning, stacklevel=3)
    return False


class EngineCache(dict):
    """Per-module cache of native engines / schedules / captured steps (static device buffers, ctypes descriptors).  It is
    state OF the process, not of the module: `copy.deepcopy(net)`, `pickle` and `torch.save(net)` get an empty one and the copy
    builds its own engines on first use."""

    MAX_ENTRIES = 6                     # engines hold GBs of static planes: a validation loop over many frame sizes must not keep them all
    evictions = 0                       # process-wide count: a loop that thrashes the cache shows here (and warns once per cache)

    def get(self, key, default=None):
        """A hit makes the entry the most recently USED one."""
        if key in self:
            value = super().pop(key)
            super().__setitem__(key, value)
            return value
        return default

    def __setitem__(self, key, value):
        """Least-recently-USED eviction.  A step that captured graphs over an evicted engine's buffers holds the engine object
        itself, so its memory stays valid; the cache only stops handing it out."""
        if key in self:
            super().pop(key)
        elif len(self) >= self.MAX_ENTRIES:
            del self[next(iter(self))]
            EngineCache.evictions += 1
            if not getattr(self, "_warned", False):
                self._warned = True
                import warnings
                warnings.warn(f"understanding_flow_robustness_amd: more than {self.MAX_ENTRIES} engine configurations alive on one "
                              "module; the least recently used one is rebuilt on its next use (EngineCache.evictions counts)",
                              RuntimeWarning, stacklevel=3)
        super().__setitem__(key, value)

    def __deepcopy__(self, memo):
        return EngineCache()

    def __reduce__(self):
        return (EngineCache, ())


# The engines keep their results and input gradients in STATIC buffers; their autograd Functions hand autograd CLONES, because a
# caller may keep a flow or a `.grad` across the next forward.  A composition whose every consumer reads the tensor at once and
# keeps nothing (FlowNet2's native path: fn2_glue.py Functions and engine Functions feeding each other) says so with
# `with static_handoff():` around its sub-network calls; the Functions then pass aliases of the static buffers, forward and
# (the choice is remembered in ctx) backward -- ~30 copy kernels per FlowNet2 iteration less.
# Round 6: the state is THREAD-LOCAL (another thread's forward must not inherit it), and the input GRADIENTS are aliased only
# where the composition says the differentiated tensor has ONE consumer (`input_grads=True`, the default inside the context): a gradient
# that feeds a tensor with several consumers -- FlowNet2's stacked frames `x`, read by FlowNetC, FlowNet-SD, two warp stages and the
# fusion input -- goes into autograd's accumulation buffer, where the FIRST arrival may be kept by reference: such calls are wrapped in
# `static_handoff(input_grads=False)` and hand autograd clones, so no `.grad` can end up a view of an engine buffer that the next
# forward overwrites (tests/test_fn2_glue_gpu.py::test_input_gradients_survive_the_next_forward).
import threading

_STATIC_HANDOFF = threading.local()


def _handoff_stack() -> list:
    st = getattr(_STATIC_HANDOFF, "stack", None)
    if st is None:
        st = _STATIC_HANDOFF.stack = []
    return st


class static_handoff:
    def __init__(self, input_grads: bool = True):
        self.input_grads = bool(input_grads)

    def __enter__(self):
        _handoff_stack().append(self.input_grads)

    def __exit__(self, *exc):
        _handoff_stack().pop()


def static_ok() -> bool:
    """Forward results may be aliases of the engines' static buffers (every consumer reads at once)."""
    return len(_handoff_stack()) > 0


def static_grads_ok() -> bool:
    """Input gradients may be aliases too: the innermost context says the differentiated inputs have one consumer each."""
    st = _handoff_stack()
    return len(st) > 0 and st[-1]


_GRAD_FLAGS_ATTR = "_ufr_grad_flags"


def freeze_parameters(module) -> None:
    """(A caller who changes `requires_grad` by hand between two attacks calls `patch_attack.release(net)` FIRST: the record is made
    once per module and `release()` / `restore_parameters()` write it back as it was made.)
    The fused steps compute data gradients only: they freeze the caller's parameters.  The flags the CALLER had are
    recorded ONCE per module (`module.__dict__[_GRAD_FLAGS_ATTR]`, an EngineCache-like dict that copies / pickles empty), at the
    first freeze -- a later step sees parameters that are already frozen and must not overwrite the record, and the record must
    not depend on which step the LRU step cache happens to hold or to iterate first."""
    flags = module.__dict__.setdefault(_GRAD_FLAGS_ATTR, _GradFlags())
    for p in module.parameters():
        flags.setdefault(id(p), (p, p.requires_grad))
        p.requires_grad_(False)


def restore_parameters(module) -> None:
    """Give the parameters back the `requires_grad` flags they had before the first `freeze_parameters(module)`."""
    for p, flag in module.__dict__.pop(_GRAD_FLAGS_ATTR, {}).values():
        p.requires_grad_(flag)


class _GradFlags(dict):
    def __deepcopy__(self, memo):
        """A deep copy of a frozen module is frozen too, so it carries the CALLER's flags for its own parameters: the record is
        re-keyed onto the copies (`memo` maps every original parameter onto its copy), and `restore_parameters(copy)` gives the
        copy's parameters back what the caller had set (round 6, ADVICE r5: the copy used to get an empty record and stayed frozen)."""
        import copy as _copy
        out = _GradFlags()
        for p, flag in self.values():
            q = _copy.deepcopy(p, memo)
            out[id(q)] = (q, flag)
        return out

    def __reduce__(self):
        return (_GradFlags, ())


class LruDict(dict):
    """A small module-level workspace cache: `get` makes an entry the most recently used one, an insert beyond `capacity` drops the
    LEAST recently used one (round 6, ADVICE r5: these caches used to be cleared whole at a fixed count)."""

    def __init__(self, capacity: int):
        super().__init__()
        self.capacity = int(capacity)

    def get(self, key, default=None):
        if key in self:
            value = super().pop(key)
            super().__setitem__(key, value)
            return value
        return default

    def __setitem__(self, key, value):
        if key in self:
            super().pop(key)
        elif len(self) >= self.capacity:
            del self[next(iter(self))]
        super().__setitem__(key, value)


def engine_cache(module, attr: str) -> EngineCache:
    return module.__dict__.setdefault(attr, EngineCache())
