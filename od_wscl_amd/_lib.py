"""ctypes binding of libodwscl.so (the C-ABI in include/odwscl.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
operator below hands raw device pointers + the current stream to the hand
written gfx950 kernels.  There is NO fallback: if the shared object is missing
or a tensor is not on the GPU the call raises (the reference raises
"Not implemented on the CPU" the same way, csrc/ROIPool.h:23).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libodwscl.so")
_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_p = ctypes.c_void_p
c_u = ctypes.c_uint32

# name -> (restype, argtypes); must list every symbol include/odwscl.h declares
SIGNATURES = {
    "odw_last_error": (ctypes.c_char_p, []),
    "odw_version": (c_i, []),
    "odw_roi_pool_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_roi_pool_forward_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "odw_roi_pool_forward": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_l, c_p]),
    "odw_roi_pool_backward": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_roi_pool_backward_det": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_roi_align_forward": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_roi_align_forward_ws": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_roi_align_backward": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_roi_align_backward_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_roi_align_forward_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "odw_roi_align_backward_ws": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_nms_workspace": (c_l, [c_i]),
    "odw_nms": (c_i, [c_p, c_p, c_i, c_f, c_i, c_p, c_p, c_p, c_l, c_p]),
    "odw_box_iou": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p]),
    "odw_pairwise_sim": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "odw_pairwise_sim_ld": (c_i, [c_p, c_i, c_i, c_p, c_l, c_p]),
    "odw_pairwise_sim_planes_min": (c_i, []),
    "odw_pairwise_sim_workspace": (c_l, [c_i, c_i]),
    "odw_pairwise_sim_ws": (c_i, [c_p, c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_pairwise_split_planes": (c_i, [c_p, c_i, c_p, c_p]),
    "odw_pairwise_sim_planes": (c_i, [c_p, c_i, c_p, c_p]),
    "odw_supcon_workspace": (c_l, [c_i]),
    "odw_supcon_v2": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_l, c_p]),
    "odw_rng_uniform": (c_i, [c_p, c_l, c_u, c_u, c_u, c_p]),
    "odw_rng_normal": (c_i, [c_p, c_l, c_u, c_u, c_p]),
    "odw_dropblock_keep_mask": (c_i, [c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_p, c_p, c_p]),
    "odw_dropout": (c_i, [c_p, c_p, c_l, c_u, c_u, c_f, c_p]),
    "odw_noise_mul": (c_i, [c_p, c_p, c_l, c_u, c_u, c_p]),
    "odw_stack_clean_aug": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "odw_unstack_clean_aug_bwd": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "odw_rows_drop_noise": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_u, c_u, c_p, c_p, c_i, c_i, c_p]),
    "odw_roi_pool_stack_forward": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_p]),
    "odw_roi_pool_stack_nhwc_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "odw_roi_pool_stack_forward_nhwc": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_p]),
    "odw_roi_pool_stack_nhwc_f32_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "odw_roi_pool_stack_forward_nhwc_f32": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p,
                                                  c_p, c_p, c_l, c_p]),
    "odw_roi_pool_stack_forward_nhwc_f32_cm": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p,
                                                     c_p, c_p, c_l, c_l, c_p, c_l, c_p]),
    "odw_roi_pool_stack_backward": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                          c_i, c_p, c_p]),
    "odw_roi_pool_stack_backward_ws": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                             c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_rows_drop_noise_bwd": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_u, c_u, c_p, c_p, c_p]),
    "odw_rows_drop_noise_bwd_store": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_u, c_u, c_p, c_p, c_p]),
    "odw_l2norm_rows": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "odw_l2norm_rows_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p]),
    "odw_gemm_nt_bf16_variant": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i]),
    "odw_gemm_nt_cm_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_gemm_nt_cm_pair_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_gemm_nt_cm": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_f, c_i, c_p,
                             c_p, c_p, c_p, c_l, c_p]),
    "odw_split_rows_cm": (c_i, [c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_l, c_p]),
    "odw_gemm_nt_bf16_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "odw_gemm_nt_bf16_ws": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_f, c_f, c_i, c_p, c_p, c_p,
                                  c_i, c_p, c_l, c_p]),
    "odw_gemm_nt_bf16": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_f, c_f, c_i, c_p, c_p, c_i, c_p]),
    "odw_linear_bwd_prep": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_f, c_p, c_i, c_p, c_i, c_p, c_p]),
    "odw_linear_bwd_prep_part": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_f, c_p, c_i, c_p, c_i, c_i, c_p, c_p]),
    "odw_transpose_to_bf16_part": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "odw_transpose_to_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "odw_f32_to_bf16": (c_i, [c_p, c_p, c_l, c_p]),
    "odw_sgd_momentum": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_p]),
    "odw_sgd_momentum_paced": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_i, c_p]),
    "odw_discover_iou": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_i, c_p, c_p]),
    "odw_discover_sim": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_i,
                               c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "odw_conv3x3_planes2_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "odw_conv3x3_planes2_ws": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_l, c_p]),
    "odw_maxpool2x2_nhwc_f32_planes2": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "odw_conv3x3_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_conv3x3_workspace_hw": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "odw_conv3x3_nhwc_bf16_ws": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p,
                                       c_p, c_l, c_p]),
    "odw_conv3x3_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_p]),
    "odw_conv_weight_prep": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p]),
    "odw_conv_wgrad_unpack": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_conv_weight_prep_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "odw_conv_weight_prep_planes_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "odw_conv_wgrad_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "odw_conv_wgrad_nt": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_l, c_p]),
    "odw_colsum_bf16": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "odw_colsum_workspace": (c_l, [c_i, c_i]),
    "odw_colsum_bf16_ws": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_conv_wgrad_tn_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_conv_wgrad_tn": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_l, c_p]),
    "odw_conv_wgrad_tn_bias_workspace": (c_l, [c_i, c_i, c_i]),
    "odw_conv_wgrad_tn_bias": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_l, c_p]),
    "odw_im2col_t_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "odw_maxpool2x2_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_maxpool2x2_nhwc_bf16_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_nhwc_bf16_to_nchw_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "odw_nchw_f32_to_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_refine_workspace": (c_l, [c_i]),
    "odw_wsddn_scores": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_p]),
    "odw_refine_losses": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p,
                                c_p, c_l, c_p]),
    "odw_detect_postprocess": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i,
                                     c_p, c_p, c_p, c_p, c_p]),
    "odw_add_relu_bf16": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "odw_relu_bwd_bf16": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "odw_stem_conv7x7_bn_relu": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_stem_conv3x3_bias_relu": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_stem_conv3x3_bias_relu_planes": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_l, c_i, c_p]),
    "odw_maxpool3x3s2_nhwc_bf16": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_detect_decode": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "odw_detect_filter": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p]),
    "odw_image_preprocess_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "odw_image_preprocess": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_l, c_p]),
    "odw_od_assign_indexed": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p]),
    "odw_od_assign_indexed_dev": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p]),
    "odw_im2col_t_bf16_part": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "odw_maxpool2x2_nhwc_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_maxpool2x2_nhwc_f32_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_maxpool2x2_nhwc_f32x_bf16_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_nchw_f32_to_nhwc_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_nhwc_f32_to_nchw_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_split_rows_bf16": (c_i, [c_p, c_l, c_i, c_i, c_p, c_i, c_p, c_l, c_i, c_p]),
    "odw_split_cols_bf16": (c_i, [c_p, c_l, c_i, c_i, c_p, c_i, c_p, c_l, c_i, c_p]),
    "odw_linear_bwd_mask_workspace": (c_l, [c_i, c_i]),
    "odw_linear_bwd_mask_f32": (c_i, [c_p, c_l, c_p, c_i, c_l, c_i, c_i, c_f, c_p, c_l, c_p, c_p, c_l, c_p]),
    "odw_stack_clean_aug_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "odw_rows_drop_noise_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_u, c_u, c_p, c_p, c_i, c_i, c_p]),
    "odw_rows_views_cm": (c_i, [c_p, c_l, c_l, c_p, c_i, c_i, c_i, c_i, c_f, c_u, c_u, c_u, c_u, c_p, c_p, c_l, c_l, c_p, c_l, c_i, c_p]),
    "odw_add_relu_f32": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "odw_relu_bwd_f32": (c_i, [c_p, c_p, c_p, c_l, c_p]),
    "odw_stem_conv7x7_bn_relu_f32": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_maxpool3x3s2_nhwc_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "odw_od_assign": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p]),
    # ---- device-resident control flow of the loss (round 6)
    "odw_loss_lists_a": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "odw_loss_lists_b": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i,
                               c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "odw_gather_rows2_dyn": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p, c_p]),
    "odw_scatter_rows2_dyn": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_f, c_p, c_p, c_p]),
    "odw_gather_rows_dyn": (c_i, [c_p, c_l, c_p, c_p, c_i, c_l, c_p, c_l, c_p]),
    "odw_zero_rows_dyn": (c_i, [c_p, c_l, c_l, c_p, c_i, c_p]),
    "odw_gemm_nt_bf16_dyn_workspace": (c_l, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p]),
    "odw_gemm_nt_bf16_dyn": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_f, c_f, c_p, c_p, c_i, c_p, c_i,
                                   c_i, c_p, c_l, c_p]),
    "odw_gemm_nt_cm_dyn_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "odw_gemm_nt_cm_dyn": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_f, c_p, c_p, c_i, c_p,
                                 c_l, c_p]),
    "odw_transpose_to_bf16_dyn": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "odw_linear_bwd_prep_dyn": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_f, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "odw_split_rows_bf16_dyn": (c_i, [c_p, c_l, c_i, c_i, c_p, c_i, c_p, c_l, c_i, c_p, c_p]),
    "odw_l2norm_rows_dyn": (c_i, [c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    "odw_l2norm_rows_bwd_dyn": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "odw_rows_views_cm_grouped": (c_i, [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_l, c_l, c_p, c_l,
                                        c_p]),
    "odw_rows_views_bwd_store_grouped": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "odw_supcon_dyn_workspace": (c_l, [c_i]),
    "odw_supcon_v2_dyn": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_l, c_p]),
    "odw_roi_pool_stack_backward_dyn": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                              c_i, c_i, c_p, c_p, c_l, c_p]),
    "odw_stream_create_cu_mask": (c_i, [c_i, c_p, c_p]),
    "odw_stream_destroy": (c_i, [c_p]),
    "odw_gemm_nt_bf16_absmax": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_f, c_p, c_p, c_l, c_p]),
    "odw_roi_pool_stack_backward_scaled": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                                 c_i, c_i, c_p, c_p, c_l, c_p]),
}


def lib():
    """Load (once) and return the shared object; raises if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libodwscl.so is missing (%s): build it with `python -m od_wscl_amd._build` -- "
                "there is no CPU/PyTorch fallback for the hot path" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, code, lib().odw_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """torch's CURRENT stream on the current device as a raw hipStream_t (every launch goes there).
    torch.cuda.current_stream() builds a Stream object per call (~9 us, 100 calls per step): the raw getter
    is the same query without the object."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("od_wscl_amd: tensor is not on the GPU -- the hot path has no CPU implementation")
