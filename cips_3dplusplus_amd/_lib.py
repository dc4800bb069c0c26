"""ctypes binding of libcips3d_hip.so (the C ABI declared in include/cips3d_hip.h).

There is no CPU fallback: if the shared library is absent and cannot be built, importing the
symbols raises; every wrapper refuses non-HIP tensors the way the reference's bindings refuse
non-CUDA tensors (`TORCH_CHECK(x.is_cuda())`, exp/op/fused_bias_act.cpp:7-16).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcips3d_hip.so")

c_f32p = C.c_void_p
c_i64 = C.c_int64
c_int = C.c_int
c_f32 = C.c_float


class LinearDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("x", C.c_void_p), ("out", C.c_void_p),
                ("x_stride", C.c_int64), ("out_stride", C.c_int64), ("in_dim", C.c_int32),
                ("out_dim", C.c_int32), ("w_scale", C.c_float), ("b_scale", C.c_float),
                ("out_scale", C.c_float), ("out_shift", C.c_float), ("row_begin", C.c_int32),
                ("pad_", C.c_int32)]


class ModulateDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("s", C.c_void_p), ("out", C.c_void_p), ("s_stride", C.c_int64),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("ksq", C.c_int32), ("flags", C.c_int32),
                ("scale", C.c_float), ("row_begin", C.c_int32),
                ("lconst", C.c_void_p), ("bias", C.c_void_p), ("noise_w", C.c_void_p), ("fir", C.c_void_p),
                ("n_bias", C.c_int32), ("pad_", C.c_int32)]


class Range(C.Structure):
    """cips3d_range: range tracking of the split-fp16 modes (include/cips3d_hip.h)."""
    _fields_ = [("x_amax", C.c_void_p), ("x_exp", C.c_void_p), ("x_exp_const", C.c_int32), ("x_max_const", C.c_float),
                ("x_pmax", C.c_void_p), ("lconst", C.c_void_p), ("lconst2", C.c_void_p),
                ("out_amax", C.c_void_p), ("out_exp", C.c_void_p), ("next_amax", C.c_void_p), ("out_pmax", C.c_void_p),
                ("next_gain", C.c_float), ("half_chip", C.c_int32), ("ride", C.c_void_p)]


class ReduceJob(C.Structure):
    """cips3d_reduce_job: a ToRGB fold that rides on a split-planes GEMM launch (include/cips3d_hip.h)."""
    _fields_ = [("part", C.c_void_p), ("bias", C.c_void_p * 8), ("skip", C.c_void_p), ("out", C.c_void_p),
                ("n4", C.c_int64), ("HW4", C.c_int64), ("slot_stride", C.c_int64), ("n_slots", C.c_int32), ("n_bias", C.c_int32)]


class AdamEntry(C.Structure):
    """cips3d_adam_entry (include/cips3d_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


class AdamHyper(C.Structure):
    """cips3d_adam_hyper (include/cips3d_hip.h)."""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int32)]


class ActBwd(C.Structure):
    """cips3d_actbwd: operands of the activation-backward epilogue (include/cips3d_hip.h)."""
    _fields_ = [("y", C.c_void_p), ("rgb_w", C.c_void_p), ("drgb", C.c_void_p), ("d_bias", C.c_void_p),
                ("d_noise_w", C.c_void_p), ("d_rgb_w", C.c_void_p),
                ("slots", C.c_int32), ("slot_stride", C.c_int32), ("rgb_slot_stride", C.c_int32), ("pad_", C.c_int32)]


AMAX_SLOTS, AMAX_STRIDE = 8, 64                 # CIPS3D_AMAX_SLOTS / CIPS3D_AMAX_STRIDE (re-read from the library by load())
AMAX_FLOATS = AMAX_SLOTS * AMAX_STRIDE          # floats per (tensor, sample) of an amax array
FEATURES_EXP = -14
PLANES_EXP_BLOCK = 128


class NerfParams(C.Structure):
    _fields_ = [("cam_poses", C.c_void_p), ("focals", C.c_void_p), ("near_", C.c_void_p), ("far_", C.c_void_p),
                ("perturb_u", C.c_void_p),
                ("w_first", C.c_void_p), ("packed", C.c_void_p), ("w_view", C.c_void_p), ("film", C.c_void_p),
                ("layer_bias", C.c_void_p), ("w_sigma", C.c_void_p), ("w_rgb", C.c_void_p),
                ("b_sigma", C.c_void_p), ("b_rgb", C.c_void_p), ("sigmoid_beta", C.c_void_p),
                ("B", C.c_int32), ("img_size", C.c_int32), ("n_samples", C.c_int32), ("hidden", C.c_int32),
                ("depth", C.c_int32), ("static_viewdirs", C.c_int32), ("n_chunks", C.c_int32), ("n_rays", C.c_int32),
                ("part", C.c_void_p), ("sdf", C.c_void_p),
                ("x_pts", C.c_void_p), ("x_rays_d", C.c_void_p), ("x_viewdirs", C.c_void_p), ("x_z_vals", C.c_void_p),
                ("o_features", C.c_void_p), ("o_thumb", C.c_void_p), ("o_xyz", C.c_void_p), ("o_mask", C.c_void_p),
                ("features_planes", C.c_int32), ("raw_density", C.c_int32),
                ("stash", C.c_void_p), ("bwd_sdf", C.c_void_p), ("bwd_crgb", C.c_void_p), ("packed32", C.c_void_p),
                ("zero_words", C.c_void_p), ("n_zero_words", C.c_int64), ("mask_planar", C.c_int32), ("pad3_", C.c_int32)]


class NerfBwdGeom(C.Structure):
    _fields_ = [("cam_poses", C.c_void_p), ("focals", C.c_void_p), ("near_", C.c_void_p), ("far_", C.c_void_p),
                ("perturb_u", C.c_void_p), ("B", C.c_int32), ("img_size", C.c_int32), ("n_samples", C.c_int32),
                ("static_viewdirs", C.c_int32)]


class NerfBwdFusedParams(C.Structure):
    _fields_ = [("geom", NerfBwdGeom)] + [(n, C.c_void_p) for n in (
        "w_first", "packed", "packed_t", "w_view", "film", "layer_bias", "w_sigma", "b_sigma", "w_rgb", "b_rgb",
        "sigmoid_beta", "d_features", "d_thumb", "stash", "scratch", "dfilm", "dcam")] + [
        ("hidden", C.c_int32), ("depth", C.c_int32), ("n_chunks", C.c_int32), ("pad_", C.c_int32),
        ("fwd_sdf", C.c_void_p), ("fwd_crgb", C.c_void_p)]


_SIGS = {
    "cips3d_abi_version": (c_int, []),
    "cips3d_strerror": (C.c_char_p, [c_int]),
    "cips3d_fused_bias_act": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int, c_int, c_f32, c_f32,
                                      C.c_void_p]),
    "cips3d_upfirdn2d": (c_int, [c_f32p, c_f32p, c_f32p, c_i64] + [c_int] * 13 + [C.c_void_p]),
    "cips3d_rgb_to_uint8": (c_int, [c_f32p, C.c_void_p, c_i64, C.c_void_p]),
    "cips3d_linear": (c_int, [c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_int,
                              c_int, c_f32, c_f32, c_f32, c_f32p, c_f32, c_int, c_i64, C.c_void_p]),
    "cips3d_pixel_norm": (c_int, [c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    "cips3d_linear_table": (c_int, [C.c_void_p, c_int, c_int, c_int, C.c_void_p]),
    "cips3d_linear_table_bwd": (c_int, [C.c_void_p, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p,
                                        c_f32p, C.c_void_p]),
    "cips3d_camera_params": (c_int, [c_f32p, c_f32p, c_f32, c_f32p, c_f32, c_int, c_int, c_f32p, c_f32p, c_f32p,
                                     c_f32p, C.c_void_p]),
    "cips3d_nerf_pack_weights": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    "cips3d_nerf_pack_weights32": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    "cips3d_nerf_packed_floats": (c_i64, [c_int, c_int]),
    "cips3d_nerf_suggest_chunks": (c_int, [c_int, c_int, c_int]),
    "cips3d_nerf_part_floats": (c_i64, [c_int, c_int, c_int, c_int]),
    "cips3d_nerf_render": (c_int, [C.POINTER(NerfParams), C.c_void_p]),
    "cips3d_nerf_fuses_finish": (c_int, [C.POINTER(NerfParams)]),
    "cips3d_nerf_finish": (c_int, [c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_nerf_finish_rays": (c_int, [c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_modulate_weights": (c_int, [c_f32p, c_f32p, c_i64, c_f32p, c_int, c_int, c_int, c_int, c_f32, c_int,
                                        C.c_void_p]),
    "cips3d_modulate_table": (c_int, [C.c_void_p, c_int, c_int, c_int, c_f32, C.c_void_p]),
    "cips3d_absmax": (c_int, [c_f32p, c_int, c_i64, c_f32p, C.c_void_p]),
    "cips3d_absmax_raise": (c_int, [c_f32p, c_int, c_i64, c_f32p, C.c_void_p]),
    "cips3d_amax_layout": (c_int, [C.POINTER(c_int), C.POINTER(c_int)]),
    "cips3d_split_words": (c_int, [c_f32p, c_f32, C.c_void_p, C.c_void_p, c_i64, C.c_void_p]),
    "cips3d_range_consts": (c_int, [c_f32p, c_int, c_f32p, c_f32, c_f32p, c_f32, c_f32p, c_f32p, c_int, C.c_void_p]),
    "cips3d_modconv1x1_supported": (c_int, [c_int, c_int, c_i64]),
    "cips3d_modconv1x1": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_i64, c_int, c_f32p, c_i64, c_f32p,
                                  c_f32p, C.c_void_p, C.c_void_p]),
    "cips3d_modconv1x1_torgb": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_i64, c_int, c_f32p, c_i64, c_f32p,
                                        c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_torgb_reduce": (c_int, [c_f32p, c_int, C.c_void_p, c_int, c_f32p, c_f32p, c_int, c_i64, C.c_void_p]),
    "cips3d_up2_fir_act": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_i64, c_f32p, c_f32p,
                                   c_f32p, C.c_void_p]),
    "cips3d_noise_bias_act": (c_int, [c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_fused_up_conv_supported": (c_int, [c_int, c_int, c_int]),
    "cips3d_fused_flat_conv_supported": (c_int, [c_int, c_int, c_int]),
    "cips3d_fused_up_conv": (c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p,
                                     c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_int, c_int, c_int,
                                     C.c_void_p, C.c_void_p]),
    "cips3d_fused_up_conv_chains": (c_int, [c_int]),
    "cips3d_fused_up_conv_next": (c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p,
                                          c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int,
                                          c_int, C.c_void_p, C.c_void_p]),
    "cips3d_torgb": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_int, c_int, c_int, c_int,
                             C.c_void_p]),
    "cips3d_modconv_kxk": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   C.c_void_p]),
    "cips3d_planes_supported": (c_int, [c_int, c_int, c_i64]),
    "cips3d_to_planes": (c_int, [c_f32p, C.c_void_p, c_int, c_int, c_i64, c_f32p, C.c_void_p, c_f32p, C.c_void_p]),
    "cips3d_from_planes": (c_int, [C.c_void_p, c_f32p, c_int, c_int, c_i64, C.c_void_p, C.c_void_p]),
    "cips3d_modconv1x1_planes": (c_int, [C.c_void_p, c_f32p, C.c_void_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_f32p, c_i64,
                                         c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_to_planes16": (c_int, [c_f32p, C.c_void_p, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_from_planes16": (c_int, [C.c_void_p, c_f32p, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_modconv1x1_planes16": (c_int, [C.c_void_p, c_f32p, C.c_void_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_f32p,
                                           c_i64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_rng_fill": (c_int, [C.c_uint64, C.c_uint64, c_f32p, c_i64, c_f32p, c_i64, C.c_void_p]),
    "cips3d_rng_fill_threads": (c_i64, [c_i64, c_i64]),
    "cips3d_rng_words": (c_int, [C.c_uint64, C.c_uint64, C.c_void_p, c_i64, C.c_void_p]),
    "cips3d_modconv3x3_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cips3d_modconv3x3": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_f32p, c_int, c_f32p,
                                  c_i64, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "cips3d_rays_in_world": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_z_vals": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, C.c_void_p]),
    "cips3d_z_vals_stratified": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, C.c_void_p]),
    "cips3d_ray_points": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_volume_integration": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int,
                                          c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_points_linear": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_int, c_int, c_int, c_f32, c_f32, c_f32p,
                                     C.c_void_p]),
    "cips3d_linear_bwd": (c_int, [c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_f32p, c_i64, c_int, c_int, c_int, c_f32, c_f32,
                                  c_int, c_f32, c_f32, c_f32p, c_i64, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_modulate_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_int, c_f32, c_int, c_f32p,
                                    c_f32p, c_i64, C.c_void_p]),
    "cips3d_pack_weights": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    "cips3d_gemm_wgrad": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_modconv1x1_actbwd": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_i64, c_int, C.c_void_p, c_f32p, c_i64,
                                         C.c_void_p, C.c_void_p]),
    "cips3d_act_tail_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int,
                                    c_int, c_i64, c_int, c_int, c_int, C.c_void_p]),
    "cips3d_slot_reduce": (c_int, [C.c_void_p, c_int, c_int, C.c_void_p]),
    "cips3d_up2_fir_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, C.c_void_p]),
    "cips3d_modulate_table_bwd": (c_int, [C.c_void_p, c_int, c_int, c_int, C.c_void_p]),
    "cips3d_decoder_grad_forward": (c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_decoder_grad_backward": (c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_sizeof_grad_plan": (c_int, []),
    "cips3d_sizeof_grad_io": (c_int, []),
    "cips3d_adam_step": (c_int, [C.c_void_p, c_int, C.c_float, C.c_float, C.c_float, C.c_float, c_int, C.c_void_p]),
    "cips3d_adam_step_groups": (c_int, [C.c_void_p, C.c_void_p, c_int, C.c_void_p, c_int, C.c_void_p]),
    "cips3d_gemm_wgrad_split": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_i64, c_f32p, c_f32p, c_int, C.c_void_p]),
    "cips3d_noise_bias_act_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_torgb_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_nerf_bwd_points": (c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                       C.c_void_p]),
    "cips3d_nerf_bwd_film": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int, c_int, c_int,
                                     c_i64, C.c_void_p]),
    "cips3d_nerf_bwd_heads": (c_int, [c_f32p, c_f32p, c_int, c_int, c_f32p, c_int, c_int, c_int, c_i64, c_f32p, C.c_void_p]),
    "cips3d_nerf_bwd_dot": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_i64, c_f32p, C.c_void_p]),
    "cips3d_nerf_bwd_composite": (c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_f32p, c_f32p, C.c_void_p]),
    "cips3d_nerf_bwd_row_dots": (c_int, [c_f32p, c_f32p, c_int, c_i64, c_f32p, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_nerf_bwd_film_grad": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          c_f32p, c_f32p, c_int, c_int, c_int, c_i64, C.c_void_p]),
    "cips3d_nerf_bwd_camera": (c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_nerf_bwd_camera_acc": (c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "cips3d_camera_params_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_f32p, C.c_void_p]),
    "cips3d_nerf_bwd_fused_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "cips3d_nerf_bwd_fused_stash_floats": (c_i64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cips3d_nerf_bwd_fused_scratch_floats": (c_i64, [c_int, c_int, c_int, c_int, c_int]),
    "cips3d_nerf_pack_weights_t": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, C.c_void_p]),
    "cips3d_nerf_bwd_fused": (c_int, [C.c_void_p, C.c_void_p]),
    "cips3d_generator_forward": (c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_style_phase": (c_int, [C.c_void_p, C.c_void_p, c_int, C.c_void_p]),
    "cips3d_sqdiff_pair_partials": (c_int, [C.c_int64, C.c_int64]),
    "cips3d_sqdiff_pair": (c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_sqdiff_pair_bwd": (c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cips3d_sizeof_plan": (c_i64, []),
    "cips3d_sizeof_io": (c_i64, []),
    "cips3d_sizeof_struct": (c_i64, [c_int]),
}

EXPORTED = tuple(_SIGS)
ABI_VERSION = 30           # == CIPS3D_ABI_VERSION of include/cips3d_hip.h
_lib = None


def _struct_table():
    """index of cips3d_sizeof_struct -> the ctypes mirror of that struct (plan.py holds the two big ones)."""
    from . import plan
    return {0: plan.GeneratorPlan, 1: plan.ForwardIO, 2: NerfParams, 3: LinearDesc, 4: ModulateDesc, 5: plan.DecLayer,
            6: NerfBwdGeom, 7: NerfBwdFusedParams, 8: Range, 9: ReduceJob}


def load(build_if_missing=True):
    """dlopen the HIP library.  A missing OR stale library (older than any csrc/*.hip, *.h or the public header) is rebuilt
    in-tree with hipcc first; with `build_if_missing=False` either case raises instead."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build
    if not build.up_to_date():
        state = "stale (older than its sources)" if os.path.exists(LIB_PATH) else "missing"
        if not build_if_missing:
            raise RuntimeError(f"{LIB_PATH} is {state}: run `python -m cips_3dplusplus_amd.build`")
        build.build_library(force=True)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError here = ABI mismatch between header and library
        fn.restype = res
        fn.argtypes = args
    if lib.cips3d_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libcips3d_hip.so has ABI version {lib.cips3d_abi_version()}, this binding expects {ABI_VERSION}; "
                           "rebuild with cips_3dplusplus_amd.build")
    for which, st in _struct_table().items():
        if lib.cips3d_sizeof_struct(which) != C.sizeof(st):
            raise RuntimeError(f"layout mismatch for {st.__name__}: library {lib.cips3d_sizeof_struct(which)} bytes, "
                               f"binding {C.sizeof(st)} bytes")
    global AMAX_SLOTS, AMAX_STRIDE, AMAX_FLOATS
    sl, st = C.c_int(0), C.c_int(0)
    lib.cips3d_amax_layout(C.byref(sl), C.byref(st))
    AMAX_SLOTS, AMAX_STRIDE, AMAX_FLOATS = sl.value, st.value, sl.value * st.value
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().cips3d_strerror(code).decode()
        raise RuntimeError(f"{what} failed ({code}): {msg}")


def dev_ptr(t, name="tensor", allow_none=False, dtype=torch.float32):
    """Device pointer of a contiguous HIP tensor of `dtype` (fp32 unless stated); loud failure for anything else."""
    if t is None:
        if allow_none:
            return None
        raise RuntimeError(f"{name} is required")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (the cips3d HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def stream_ptr():
    """The current HIP stream of the current device as an integer (torch._C._cuda_getCurrentRawStream: a plain C call -- the
    torch.cuda.current_stream() object costs ~10 us to build, twice per forward that was a tenth of its host time)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:          # (a torch build without the private accessor)
        return torch.cuda.current_stream().cuda_stream
