"""ctypes binding of libssv_hip.so (include/ssv_hip.h) - the only door into the HIP hot path.

There is no CPU fallback: if the library is missing or a call fails this module raises.
torch is used for device memory and the current HIP stream only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSV_HIP_LIB") or os.path.join(_HERE, "csrc", "libssv_hip.so")   # override: diagnostic builds only

ABI_VERSION = 120        # ssv_version() of the library this binding was written against (include/ssv_hip.h)
PROF_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn_fwd", "bn_bwd", "pool", "loss", "optim", "aug", "misc", "attn", "norm")


class SsvError(RuntimeError):
    pass


ARITH_F32_MFMA, ARITH_BF16X3 = 0, 6      # ssv_conv_desc.arithmetic (include/ssv_hip.h)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo", "arithmetic", "reserved")] + [("w_planes", C.c_void_p)]


class BnGate(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "scale", "shift", "mask", "mean", "invstd", "psum_g", "psum_gx", "x2", "mean2", "invstd2", "psum_gx2")]


class BnDyin(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "coef")]


class AugCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("brightness", "contrast", "saturation", "hue", "p_jitter", "p_gray", "p_flip",
                                          "scale_min", "scale_max", "ratio_min", "ratio_max")]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_u64 = C.c_uint64
_f3 = C.POINTER(C.c_float)
_cd = C.POINTER(ConvDesc)

# name -> (restype, argtypes); mirrors include/ssv_hip.h one to one
SIGNATURES = {
    "ssv_version": (C.c_int, []),
    "ssv_last_error": (C.c_char_p, []),
    "ssv_source_sha16": (C.c_char_p, []),
    "ssv_device_cus": (C.c_int, []),
    "ssv_conv2d_fwd": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_pad_channels": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ssv_group_expand": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_group_extract": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ssv_conv2d_fwd_stats_groups": (_i64, [_cd]),
    "ssv_conv2d_fwd_stats": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_fwd_bnrelu_in_stats": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_fwd_sumin_stats": (C.c_int, [_cd] + [_vp] * 13),
    "ssv_conv2d_wgrad_bnrelu_in": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_bn_stats_finalize": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_bn_apply": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ssv_bn_relu_bwd_affine": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_bn_train_fwd_partials": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, C.c_int, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_filter_transpose": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_conv2d_dgrad": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_fwd_gate_groups": (_i64, [_cd]),
    "ssv_conv2d_fwd_gated": (C.c_int, [_cd, _vp, _vp, _vp, _vp, C.POINTER(BnGate), _vp]),
    "ssv_conv2d_fwd_gated_s2add": (C.c_int, [_cd, _vp, _vp, _vp, _i32, _i32, _vp, C.POINTER(BnGate), _vp]),
    "ssv_conv2d_fwd_dyin_s2add": (C.c_int, [_cd, _vp, C.POINTER(BnDyin), _vp, _vp, _i32, _i32, _vp, C.POINTER(BnGate), _vp]),
    "ssv_conv2d_dgrad_gate_groups": (_i64, [_cd]),
    "ssv_conv2d_dgrad_gated": (C.c_int, [_cd, _vp, _vp, _vp, _vp, C.POINTER(BnGate), _vp]),
    "ssv_bn_bwd_from_partials": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_bn_bwd_coef": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_conv2d_fwd_dyin": (C.c_int, [_cd, _vp, C.POINTER(BnDyin), _vp, _vp, _vp, C.POINTER(BnGate), _vp]),
    "ssv_conv2d_wgrad_dyin": (C.c_int, [_cd, _vp, _vp, _vp, _vp, C.POINTER(BnDyin), _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_conv2d_wgrad_workspace_bytes": (_sz, [_cd]),
    "ssv_conv2d_fwd_grouped": (C.c_int, [_cd, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_dgrad_grouped": (C.c_int, [_cd, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_wgrad_grouped_workspace_bytes": (_sz, [_cd, _i32]),
    "ssv_conv2d_wgrad_grouped": (C.c_int, [_cd, _i32, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_conv2d_wgrad": (C.c_int, [_cd, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_conv2d_wgrad_bias_workspace_bytes": (_sz, [_cd]),
    "ssv_conv2d_wgrad_bias": (C.c_int, [_cd, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_stem_conv_fwd": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_stem_conv_fwd_stats_rows_per_group": (_i64, [_cd]),
    "ssv_stem_conv_wgrad_rows_per_group": (_i64, [_cd]),
    "ssv_stem_conv_wgrad_workspace_bytes": (_sz, [_cd]),
    "ssv_stem_conv_wgrad": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_wino_tiles": (_i64, [_i32, _i32, _i32]),
    "ssv_wino_groups": (_i64, [_i32, _i32, _i32]),
    "ssv_wino_stats_rows_per_group": (_i32, [_i32, _i32, _i32]),
    "ssv_wino_filter_transform": (C.c_int, [_i32, _i32, _vp, _vp, _vp]),
    "ssv_wino_filter_grad": (C.c_int, [_i32, _i32, _vp, _vp, C.c_int, _vp]),
    "ssv_wino_input_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ssv_wino_dy_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_wino_output_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(BnGate), _vp]),
    "ssv_wino44_tiles": (_i64, [_i32, _i32, _i32]),
    "ssv_wino44_groups": (_i64, [_i32, _i32, _i32, _i32]),
    "ssv_wino44_stats_rows_per_group": (_i64, [_i32, _i32, _i32]),
    "ssv_wino44_filter_transform": (C.c_int, [_i32, _i32, _vp, _vp, _vp]),
    "ssv_wino44_input_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_wino44_output_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(BnGate), _vp]),
    "ssv_wino44_dy_transform": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_wino44_filter_grad": (C.c_int, [_i32, _i32, _vp, _vp, C.c_int, _vp]),
    "ssv_wino44_dy_transform_both": (C.c_int, [_i32, _i32, _i32, _i32, _vp, C.POINTER(BnDyin), _vp, _vp, _vp]),
    "ssv_gemm_batched": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ssv_gemm_batched_split": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_gemm_batched_wgrad_split": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "ssv_split_planes": (C.c_int, [_i64, _vp, _vp, _vp]),
    "ssv_conv_arithmetic": (C.c_int, [_cd, _i32]),
    "ssv_gemm_batched_wgrad_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "ssv_gemm_batched_wgrad": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_gemm_batched_wgrad_blocked_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32, _i32]),
    "ssv_gemm_batched_wgrad_blocked": (C.c_int, [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "ssv_bn_workspace_bytes": (_sz, [_i64, _i32]),
    "ssv_bn_train_fwd": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, C.c_int, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_bn_train_bwd": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_bn_relu_maxpool_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_bn_relu_maxpool_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_colsum": (C.c_int, [_i64, _i32, _vp, _vp, C.c_int, _vp, _sz, _vp]),
    "ssv_maxpool3x3s2_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ssv_maxpool3x3s2_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ssv_gap_fwd": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_gap_bwd": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_nchw_to_nhwc": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_nhwc_to_nchw": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssv_l2norm_fwd": (C.c_int, [_i32, _i32, _vp, _i32, _f32, _vp, _i32, _vp, _vp]),
    "ssv_l2norm_bwd": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ssv_ntxent_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp]),
    "ssv_ntxent_loss": (C.c_int, [_i32, _vp, _vp, _f32, _vp, _vp]),
    "ssv_ntxent_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _vp]),
    "ssv_ntxent_default_splits": (_i64, [_i32, _i32]),
    "ssv_ntxent_split_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "ssv_ntxent_fwd_split": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "ssv_ntxent_bwd_split": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _i32, _vp, _sz, _vp]),
    "ssv_ntxent_gram_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp]),
    "ssv_ntxent_gram_weights": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _f32, _f32, _vp]),
    "ssv_reduce_workspace_bytes": (_sz, [_i64]),
    "ssv_mse_pair_fwd_bwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_scale": (C.c_int, [_i64, _vp, _vp, _vp]),
    "ssv_barlow_cgrad": (C.c_int, [_i32, _vp, _f32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "ssv_sgd_nesterov": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, C.c_int, _vp]),
    "ssv_sgd_nesterov_dev": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_ema": (C.c_int, [_i64, _vp, _vp, _f32, _vp]),
    "ssv_fill": (C.c_int, [_i64, _vp, _f32, _vp]),
    "ssv_add": (C.c_int, [_i64, _vp, _vp, _vp]),
    "ssv_augment_params": (C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(AugCfg), _u64, _u64, _vp, _i64, _vp, _vp]),
    "ssv_augment_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "ssv_augment_views": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f3, _f3, _vp, _vp, _sz, _vp]),
    "ssv_center_view": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _f3, _f3, _vp, _vp]),
    "ssv_knn_workspace_bytes": (_sz, [_i64]),
    "ssv_knn_label_agreement": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "ssv_knn_workspace_bytes_arith": (_sz, [_i64, _i32, _i32]),
    "ssv_knn_label_agreement_arith": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _sz, _vp]),
    "ssv_vit_embed_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ssv_vit_embed_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "ssv_layernorm_fwd": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "ssv_layernorm_workspace_bytes": (_sz, [_i64, _i32]),
    "ssv_layernorm_bwd": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "ssv_linear_gelu_fwd": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_conv2d_dgrad_gelu": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_linear_fwd_gelugrad": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_linear_gelu_fwd_dact": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_linear_fwd_mulgrad": (C.c_int, [_cd, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssv_gelu_fwd": (C.c_int, [_i64, _vp, _vp, _vp]),
    "ssv_gelu_bwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "ssv_attention_fwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _vp]),
    "ssv_attention_fwd_arith": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _i32, _vp]),
    "ssv_attention_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ssv_weightnorm_fwd": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ssv_weightnorm_bwd": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ssv_dino_loss_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "ssv_dino_loss": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "ssv_dino_center_update": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _f32, _vp, _vp]),
    "ssv_adamw": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "ssv_adamw_counted": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _f32, _vp]),
    "ssv_adamw_counted_dev": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _f32, _vp]),
    "ssv_multicrop_params": (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.c_double, C.c_double, C.c_uint64, C.c_uint64, _vp, _i64, _vp, _vp]),
    "ssv_multicrop": (C.c_int, [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ssv_negdot_pair_fwd_bwd": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_relic_kl_workspace_bytes": (_sz, [_i32]),
    "ssv_relic_kl_fwd_bwd": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_moco_loss_fwd_bwd": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _sz, _vp]),
    "ssv_queue_push": (C.c_int, [_i32, _i32, _vp, _i32, _i32, _vp, _f32, _vp]),
    "ssv_queue_push_counted": (C.c_int, [_i32, _i32, _vp, _vp, _i32, _vp, _f32, _vp]),
    "ssv_softmax_ce_workspace_bytes": (_sz, [_i32]),
    "ssv_softmax_ce_fwd_bwd": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssv_sgd": (C.c_int, [_i64, _vp, _vp, _vp, _f32, _f32, _f32, C.c_int, C.c_int, _vp]),
    "ssv_prof_enable": (C.c_int, [C.c_int]),
    "ssv_prof_reset": (C.c_int, []),
    "ssv_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """dlopen the library (once) and attach the signatures.  Raises SsvError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsvError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or `make -C {os.path.dirname(LIB_PATH)}`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.ssv_version() < ABI_VERSION:
        raise SsvError(f"{LIB_PATH} is ABI version {lib.ssv_version()}, this package binds version {ABI_VERSION}: rebuild it (make -C {os.path.dirname(LIB_PATH)})")
    _lib = lib
    return lib


def source_sha16():
    """The loaded library's build identity: sha256[:16] of the sources it was compiled from (include/ssv_hip.h: ssv_source_sha16)."""
    return load().ssv_source_sha16().decode()


def lib_sha16(path=None):
    """First 16 hex digits of the sha256 of the library file: profiles record it, bench.py compares it with the library it has loaded."""
    import hashlib
    h = hashlib.sha256()
    with open(path or LIB_PATH, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def _check(rc, what):
    if rc != 0:
        raise SsvError(f"{what} failed ({rc}): {load().ssv_last_error().decode()}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream's handle (what every ssv_* call is enqueued on).  torch.cuda.current_stream() builds a Python Stream object per call (~7 us, and a
    training step asks ~600 times: 2 ms of a 10 ms launch-bound step, tools/exp/r05_host_profile.py); the raw accessor returns the same handle in ~0.2 us."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise SsvError("libssv_hip operates on device tensors only (got a CPU tensor); there is no CPU fallback")


def call(name, *args):
    _check(getattr(load(), name)(*args), name)


# ------------------------------------------------------------------------------------------- workspace
class _Workspace:
    """One grow-only scratch buffer per (device, HIP stream): kernels of two streams may run concurrently, so they
    must never share scratch.  The torch caching allocator owns the memory."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device):
        key = (device, (_raw_stream(device.index if device.index is not None else _raw_device()) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
               if device.type == "cuda" else 0)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


workspace = _Workspace()


# ------------------------------------------------------------------------------------------- profiling
def prof_enable(on=True):
    call("ssv_prof_enable", int(bool(on)))


def prof_reset():
    call("ssv_prof_reset")


def prof_collect():
    ms = (C.c_double * len(PROF_CLASSES))()
    n = (C.c_int64 * len(PROF_CLASSES))()
    call("ssv_prof_collect", ms, n)
    return {k: (ms[i], n[i]) for i, k in enumerate(PROF_CLASSES)}
