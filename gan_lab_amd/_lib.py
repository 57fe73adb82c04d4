"""ctypes binding of the C-ABI kernel library (include/ganlab_hip.h -> csrc/libganlab_hip.so).

There is NO fallback: if the shared library is missing or a kernel returns an error code the call
raises.  PyTorch is used only for device memory, streams and autograd bookkeeping.
"""
import ctypes
import os
import subprocess

# PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It MUST be loaded before our
# library so that `libamdhip64.so.7` resolves to that same runtime: otherwise two runtimes coexist,
# torch's stream handles are foreign to ours and every launch fails.
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# GANLAB_HIP_LIB: file name of an A/B build of the same library under csrc/ (make VARIANT=...); never a fallback
SO_PATH = os.path.join(CSRC, os.path.basename(os.environ.get('GANLAB_HIP_LIB', '') or 'libganlab_hip.so'))

_c_int, _c_ll, _c_f, _c_p, _c_sz, _c_u64 = (ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_uint64)

ACT_NONE, ACT_LRELU = 0, 1
PACK_FWD, PACK_DGRAD = 0, 1


class ConvGeom(ctypes.Structure):
    """Mirror of `ganlab_conv_geom` (include/ganlab_hip.h)."""
    _fields_ = [('N', _c_int), ('Cin', _c_int), ('Hin', _c_int), ('Win', _c_int),
                ('Cout', _c_int), ('ks', _c_int), ('pad', _c_int), ('up', _c_int), ('pool', _c_int)]


_GP = ctypes.POINTER(ConvGeom)


class PackDesc(ctypes.Structure):
    """Mirror of `ganlab_pack_desc` (include/ganlab_hip.h): one weight re-layout of ganlab_pack_many's table."""
    _fields_ = [('src', _c_p), ('dst', _c_p), ('kind', _c_int), ('Cout', _c_int), ('Cin', _c_int), ('ks', _c_int),
                ('mode', _c_int), ('up', _c_int), ('scale', _c_f), ('reserved', _c_int), ('total', _c_ll),
                ('block0', _c_ll)]

# name -> (restype, argtypes): must list every function declared in include/ganlab_hip.h
SIGNATURES = {
    'ganlab_abi_version': (_c_int, []),
    'ganlab_last_launch': (_c_int, [ctypes.c_char_p, _c_int, ctypes.POINTER(ctypes.c_uint)]),
    'ganlab_launch_count': (ctypes.c_ulonglong, []),
    'ganlab_launch_history': (_c_int, [_c_int, ctypes.c_char_p, _c_int, ctypes.POINTER(ctypes.c_uint)]),
    'ganlab_conv_geom_size': (_c_int, []),
    'ganlab_conv_out_hw': (_c_int, [_GP, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    'ganlab_conv_pack_f32': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_dgrad_f32': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_splitk_plan': (_c_int, [_GP, _c_int]),
    'ganlab_conv_fwd_splitk_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_dgrad_splitk_f32': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p, _c_sz, _c_p]),
    'ganlab_conv_dgrad_mask_supported': (_c_int, [_GP]),
    'ganlab_conv_dgrad_mask_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_act_bwd_fused_supported': (_c_int, [_GP]),
    'ganlab_conv_dgrad_act_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_fwd_mask_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_wgrad_act_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_wgrad_workspace': (_c_sz, [_GP]),
    'ganlab_conv_wgrad_f32': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_s2_supported': (_c_int, [_GP]),
    'ganlab_conv_s2_pack_f32': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_dgrad_f32': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_s2_wgrad_workspace': (_c_sz, [_GP]),
    'ganlab_conv_s2_wgrad_f32': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_bf16_supported': (_c_int, [_GP]),
    'ganlab_conv_pack_bf16': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_fwd_bf16': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_dgrad_bf16': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_bf16_splitk_plan': (_c_int, [_GP, _c_int]),
    'ganlab_conv_fwd_bf16_splitk': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_dgrad_bf16_splitk': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p, _c_sz, _c_p]),
    'ganlab_conv_wgrad_bf16_workspace': (_c_sz, [_GP]),
    'ganlab_conv_wgrad_bf16': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_x3_supported': (_c_int, [_GP, _c_int]),
    'ganlab_conv_x3_pack': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_fwd_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_dgrad_x3': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_dgrad_mask_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_fwd_aff_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_fwd_aff_tail_x3_chunks': (_c_int, [_GP]),
    'ganlab_conv_s2_x3_supported': (_c_int, [_GP, _c_int]),
    'ganlab_conv_s2_x3_pack': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_fwd_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_fwd_aff_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_dgrad_x3': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_s2_down_x3_supported': (_c_int, [_GP, _c_int]),
    'ganlab_conv_s2_down_x3_pack': (_c_ll, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_down_fwd_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_down_dgrad_x3': (_c_int, [_c_p, _c_p, _c_p, _GP, _c_p]),
    'ganlab_conv_wgrad_x3_supported': (_c_int, [_GP]),
    'ganlab_conv_wgrad_x3_workspace': (_c_sz, [_GP]),
    'ganlab_conv_wgrad_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_s2_wgrad_x3_supported': (_c_int, [_GP]),
    'ganlab_conv_s2_wgrad_x3_workspace': (_c_sz, [_GP]),
    'ganlab_conv_s2_wgrad_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_fwd_aff_tail_x3': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f,
                                             _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_mask_bits_supported': (_c_int, [_c_int, _c_int]),
    'ganlab_blur3x3_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_p]),
    'ganlab_blur_act_bwd_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_f, _c_f,
                                              _c_p, _c_sz, _c_p]),
    'ganlab_act_bwd_blur_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                              _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_fwd_blur_supported': (_c_int, [_GP, _c_p, _c_p]),
    'ganlab_conv_fwd_blur_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_f, _c_p]),
    'ganlab_conv_s2_blur_supported': (_c_int, [_GP]),
    'ganlab_conv_s2_blur_workspace': (_c_sz, [_GP]),
    'ganlab_conv_s2_fwd_blur_tail_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int,
                                                  _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_s2_dgrad_blur_act_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_dgrad_rgb_sums_supported': (_c_int, [_GP, _c_int]),
    'ganlab_conv_dgrad_rgb_sums_workspace': (_c_sz, [_GP]),
    'ganlab_conv_dgrad_rgb_sums_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_int, _c_f, _c_f, _c_f, _c_int, _c_int,
                                                _c_p, _c_sz, _c_p]),
    'ganlab_conv_fwd_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_dgrad_act_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_fwd_mask_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p]),
    'ganlab_conv_wgrad_act_bits_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_pack_desc_size': (_c_int, []),
    'ganlab_pack_many': (_c_int, [_c_p, _c_int, _c_ll, _c_p]),
    'ganlab_in_affine_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_p]),
    'ganlab_conv_aff_supported': (_c_int, [_GP]),
    'ganlab_conv_fwd_aff_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_fwd_aff_tail_chunks': (_c_int, [_GP]),
    'ganlab_conv_fwd_aff_tail_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f,
                                              _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_wgrad_aff_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_conv_s2_aff_supported': (_c_int, [_GP]),
    'ganlab_conv_s2_fwd_aff_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_conv_s2_wgrad_aff_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_mod_conv_supported': (_c_int, [_GP]),
    'ganlab_mod_conv_stat_chunks': (_c_int, [_GP]),
    'ganlab_mod_conv_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _GP, _c_f, _c_int,
                                         _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_mod_torgb_prep_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_f, _c_p]),
    'ganlab_mod_torgb_wgrad_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_f, _c_p]),
    'ganlab_mod_torgb_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_ll, _c_p]),
    'ganlab_mod_torgb_cross_workspace': (_c_sz, [_c_int]),
    'ganlab_mod_torgb_cross_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_ll, _c_p, _c_sz, _c_p]),
    'ganlab_blur3x3_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_int, _c_int, _c_p]),
    'ganlab_up2_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_pool2_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_int, _c_int, _c_f, _c_p]),
    'ganlab_resample2d_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_int, _c_p]),
    'ganlab_bias_act_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_int, _c_f, _c_p]),
    'ganlab_act_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_f, _c_p]),
    'ganlab_act_bwd_bias_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_channel_sum_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_channel_sum_workspace': (_c_sz, [_c_int, _c_int, _c_ll]),
    'ganlab_instnorm_stats_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_ll, _c_f, _c_p]),
    'ganlab_row_stats_workspace': (_c_sz, [_c_ll, _c_ll]),
    'ganlab_row_stats_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_ll, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_instnorm_style_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'ganlab_instnorm_style_bwd_reduce_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_ll, _c_p]),
    'ganlab_instnorm_style_bwd_apply_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int,
                                                     _c_ll, _c_p]),
    'ganlab_pixelnorm_fwd_f32': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'ganlab_pixelnorm_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'ganlab_blur_fused_supported': (_c_int, [_c_int, _c_int]),
    'ganlab_blur_fused_workspace': (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    'ganlab_blur_bias_act_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_f,
                                          _c_int, _c_f, _c_p]),
    'ganlab_blur_act_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_f, _c_f,
                                         _c_p, _c_sz, _c_p]),
    'ganlab_act_bwd_blur_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                         _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_act_stats_workspace': (_c_sz, [_c_int, _c_int, _c_ll]),
    'ganlab_bias_act_stats_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_int,
                                           _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_blur_bias_act_stats_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                                _c_f, _c_int, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_instnorm_bwd_act_workspace': (_c_sz, [_c_int, _c_int, _c_ll]),
    'ganlab_instnorm_style_bwd_act_f32': (_c_int, [_c_p] * 11 + [_c_int, _c_int, _c_ll, _c_int, _c_f, _c_f, _c_p, _c_sz,
                                                                 _c_p]),
    'ganlab_instnorm_bwd_rgb_supported': (_c_int, [_c_int, _c_int, _c_int, _c_ll]),
    'ganlab_instnorm_bwd_reduce_rgb_workspace': (_c_sz, [_c_int, _c_int, _c_ll]),
    'ganlab_instnorm_style_bwd_reduce_rgb_f32': (_c_int, [_c_p, _c_p, _c_int] + [_c_p] * 5 + [_c_int, _c_int, _c_ll, _c_p,
                                                                                            _c_sz, _c_p]),
    'ganlab_instnorm_style_bwd_act_rgb_f32': (_c_int, [_c_p, _c_p, _c_int] + [_c_p] * 10 + [_c_int, _c_int, _c_ll, _c_int,
                                                                                         _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_instnorm_style_bwd_act_blur_f32': (_c_int, [_c_p] * 11 + [_c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_f,
                                                                      _c_p, _c_sz, _c_p]),
    'ganlab_u8_box_decode_f32': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p]),
    'ganlab_chan_affine_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'ganlab_mul_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_p]),
    'ganlab_ln_affine_fwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_int, _c_f, _c_p]),
    'ganlab_colscale_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_p]),
    'ganlab_coldot_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_p]),
    'ganlab_ln_rowsums_workspace': (_c_sz, [_c_int, _c_ll]),
    'ganlab_ln_rowsums_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_p, _c_sz, _c_p, _c_p,
                                       _c_f, _c_p]),
    'ganlab_ln_project_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_p]),
    'ganlab_ln_bwd_cols_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_p]),
    'ganlab_ln_bwdbwd_apply_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_ll, _c_p]),
    'ganlab_bn_stats_f32': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_p, _c_sz, _c_p]),
    'ganlab_bn_finalize_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_f, _c_f, _c_p]),
    'ganlab_bn_apply_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_int, _c_f, _c_p]),
    'ganlab_bn_bwd_sums_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_p, _c_sz, _c_p, _c_p, _c_f,
                                        _c_p]),
    'ganlab_bn_bwd_apply_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_p]),
    'ganlab_tanh_fwd_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_p]),
    'ganlab_tanh_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_p]),
    'ganlab_mbstd_fwd_f32': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'ganlab_mbstd_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'ganlab_mbstd_bwdbwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_p]),
    'ganlab_axpby_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_f, _c_f, _c_p]),
    'ganlab_scale_dev_f32': (_c_int, [_c_p, _c_p, _c_p, _c_ll, _c_f, _c_p]),
    'ganlab_lerp_rows_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_ll, _c_p]),
    'ganlab_sum_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_f, _c_int, _c_p, _c_sz, _c_p]),
    'ganlab_sum_workspace': (_c_sz, [_c_ll]),
    'ganlab_bce_logits_fwd_f32': (_c_int, [_c_p, _c_p, _c_int, _c_f, _c_p]),
    'ganlab_bce_logits_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_f, _c_p]),
    'ganlab_chnorm_penalty_fwd_f32': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_f, _c_p, _c_sz, _c_p]),
    'ganlab_chnorm_penalty_bwd_f32': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_ll, _c_f, _c_f, _c_p]),
    'ganlab_adam_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_p]),
    'ganlab_ewma_f32': (_c_int, [_c_p, _c_p, _c_ll, _c_f, _c_p]),
    'ganlab_randn_f32': (_c_int, [_c_p, _c_ll, _c_u64, _c_u64, _c_p]),
    'ganlab_randn_dev_f32': (_c_int, [_c_p, _c_ll, _c_u64, _c_p, _c_u64, _c_p]),
    'ganlab_adam_dev_f32': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_f, _c_f, _c_f, _c_f, _c_p]),
    'ganlab_step_scalars_size': (_c_int, []),
    'ganlab_set_step_scalars': (_c_int, [_c_p, _c_u64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_p]),
}

_LIB = None


class GanlabLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise GanlabLibraryError('building libganlab_hip.so failed:\n' + res.stderr[-4000:])
    return SO_PATH


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise GanlabLibraryError(
                f'{SO_PATH} not found: the HIP kernel library is required (there is no CPU/PyTorch '
                f'fallback). Build it with `python -c "import __graft_entry__ as g; g.build()"` or '
                f'`make -C {CSRC}`.')
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if handle.ganlab_conv_geom_size() != ctypes.sizeof(ConvGeom):
            raise GanlabLibraryError(f'ConvGeom mirror is {ctypes.sizeof(ConvGeom)} bytes, the library\'s '
                                     f'ganlab_conv_geom {handle.ganlab_conv_geom_size()}: header and binding disagree')
        if handle.ganlab_pack_desc_size() != ctypes.sizeof(PackDesc):
            raise GanlabLibraryError(f'PackDesc mirror is {ctypes.sizeof(PackDesc)} bytes, the library\'s '
                                     f'ganlab_pack_desc {handle.ganlab_pack_desc_size()}: header and binding disagree')
        _LIB = handle
    return _LIB


def last_launch():
    """(demangled kernel symbol, workgroups) of this thread's most recent launch through the library."""
    buf = ctypes.create_string_buffer(1024)
    grid = ctypes.c_uint(0)
    n = lib().ganlab_last_launch(buf, 1024, ctypes.byref(grid))
    return (buf.value.decode() if n > 0 else None), int(grid.value)


def launch_count():
    """Kernels this thread has launched through the library so far."""
    return int(lib().ganlab_launch_count())


def launches_since(count):
    """[(symbol, workgroups)] of this thread's launches after ``launch_count()`` returned ``count`` (oldest first; the
    library keeps the last 16)."""
    n = min(launch_count() - count, 16)
    out = []
    buf = ctypes.create_string_buffer(1024)
    grid = ctypes.c_uint(0)
    for back in range(n - 1, -1, -1):
        if lib().ganlab_launch_history(back, buf, 1024, ctypes.byref(grid)) > 0:
            out.append((buf.value.decode(), int(grid.value)))
    return out


_ERR = {-1: 'GANLAB_EINVAL (bad argument)', -2: 'GANLAB_EWORKSPACE (workspace too small)',
        -3: 'GANLAB_ELAUNCH (kernel launch failed)', -4: 'GANLAB_EUNSUPPORTED (geometry not handled)'}


def check(rc, what):
    if rc != 0:
        raise GanlabLibraryError(f'{what} failed: {_ERR.get(rc, rc)}')
