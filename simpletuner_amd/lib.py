"""ctypes binding of libst355.so (include/st355.h).

This is the ONLY way the Python host reaches compute: there is no eager/PyTorch fallback.  If the shared
library is missing (or was not built for gfx950) every op raises `St355Unavailable` — loudly — instead of
silently running something else (task rule: the product path must fail when the HIP extension is missing).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "csrc" / "libst355.so"

# every symbol include/st355.h declares; tests check the .so exports exactly these
SYMBOLS = [
    "st355_version", "st355_arch", "st355_last_error",
    "st355_prof_enable", "st355_prof_reset", "st355_prof_collect", "st355_prof_dump",
    "st355_flow_noise_mix", "st355_ddpm_noise_mix", "st355_mse_loss", "st355_cond_loss", "st355_cond_loss_masked",
    "st355_flux_pack", "st355_flux_unpack", "st355_patchify", "st355_unpatchify",
    "st355_timestep_proj", "st355_silu", "st355_gelu_tanh", "st355_silu_bwd", "st355_add", "st355_gather_rows", "st355_scatter_rows", "st355_scale_cols",
    "st355_gemm_bf16", "st355_gemm_bf16_grouped", "st355_gemm_set_persistent", "st355_gemm_set_tail_split", "st355_gemm_tail_placement", "st355_gemm_tn_bf16", "st355_gemm_tn_seg_bf16", "st355_fp8_quantize_weight", "st355_fp8_quantize_act", "st355_linear_fp8", "st355_colsum_workspace", "st355_colsum_prod", "st355_stats_workspace", "st355_ln_modulate_bwd_stats", "st355_scale_cols_stats", "st355_colsum_rows", "st355_transpose_bf16", "st355_sum_chunks_bf16", "st355_skinny_tn_workspace", "st355_skinny_tn", "st355_skinny_tn_seg", "st355_skinny_tn_multi",
    "st355_ln_modulate_fwd", "st355_ln_modulate_bwd",
    "st355_qk_norm_rope_fwd", "st355_qk_norm_rope_bwd", "st355_qk_norm_wgrad_workspace", "st355_qk_norm_rope_bwd_wgrad", "st355_qk_rope_norm_bwd",
    "st355_attn_set_impl", "st355_attn_fwd", "st355_attn_fwd_vrows", "st355_attn_bwd_workspace", "st355_attn_bwd", "st355_attn_bwd_rope",
    "st355_adamw_ema_step", "st355_adamw_ema_step_bf16", "st355_adamw_bf16_sr_step", "st355_ema_update", "st355_grad_norm", "st355_grad_norm_ws", "st355_grad_clamp", "st355_grad_clip_norm",
    "st355_lora_pack",
    "st355_workspace_bytes",
    "st355_comm_unique_id", "st355_comm_init", "st355_comm_destroy", "st355_comm_all_reduce", "st355_comm_reduce_scatter", "st355_comm_all_gather",
    # UNet path (SDXL / SD1.5)
    "st355_conv_grid_rows", "st355_conv_bf16", "st355_conv_wgrad_bf16", "st355_grid_from_nchw", "st355_grid_to_nchw", "st355_im2col3x3", "st355_col2im3x3", "st355_softmax_rows", "st355_vae_encode_workspace", "st355_vae_encode", "st355_block_flux_single_fwd", "st355_block_flux_single_bwd", "st355_block_flux_double_fwd", "st355_block_flux_double_bwd", "st355_block_pixart_fwd", "st355_block_pixart_bwd", "st355_block_sd3_joint_fwd", "st355_block_sd3_joint_bwd",
    "st355_upsample2x", "st355_upsample2x_bwd", "st355_tokens_to_grid", "st355_grid_to_tokens",
    "st355_groupnorm_workspace", "st355_groupnorm_fwd", "st355_groupnorm_bwd",
    "st355_layernorm_fwd", "st355_layernorm_bwd", "st355_layernorm_param_grads_workspace", "st355_layernorm_param_grads",
    "st355_geglu_fwd", "st355_geglu_bwd", "st355_head_split", "st355_head_merge", "st355_head_split_pad", "st355_head_merge_pad", "st355_softmax_rows_bwd", "st355_attn_cross_fwd", "st355_attn_cross_bwd", "st355_attn_fwd_res", "st355_attn_bwd_res",
]

KERNEL_CLASSES = [
    "gemm", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_prep",
    "ln_mod", "qk_rope", "skinny", "elementwise", "optim",
]

EPI_NONE, EPI_GELU, EPI_GATE_RESIDUAL, EPI_MUL_GELU_GRAD, EPI_ADD, EPI_QK_NORM_ROPE, EPI_GEGLU, EPI_GEGLU_GRAD, EPI_HEADS = 0, 1, 2, 3, 4, 5, 6, 7, 8


class St355Unavailable(RuntimeError):
    pass


class St355Error(RuntimeError):
    pass


class QkRope(C.Structure):
    """st355_qk_rope (include/st355.h): operands of the fused QKV projection epilogue ST355_EPI_QK_NORM_ROPE"""
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("rrms", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p),
        ("H", C.c_int32), ("S", C.c_int32), ("pos0", C.c_int32), ("eps", C.c_float),
        ("Vt", C.c_void_p), ("Sp", C.c_int32),
    ]


class FluxSingleFwdArgs(C.Structure):
    """st355_flux_single_fwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("K2", C.c_int32), ("k2_real", C.c_int32),
        ("scale", C.c_float),
        ("x", C.c_void_p),
        ("mod_shift", C.c_void_p), ("mod_scale", C.c_void_p), ("mod_gate", C.c_void_p), ("mod_stride", C.c_int64),
        ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p),
        ("A_cat", C.c_void_p), ("B_blk", C.c_void_p),
        ("norm_q", C.c_void_p), ("norm_k", C.c_void_p),
        ("w_mlp", C.c_void_p), ("b_mlp", C.c_void_p),
        ("w_out", C.c_void_p), ("ld_w_out", C.c_int64), ("b_out", C.c_void_p),
        ("cos_p", C.c_void_p), ("sin_p", C.c_void_p),
        ("key_bias", C.c_void_p),
        ("n", C.c_void_p), ("V", C.c_void_p), ("rrms", C.c_void_p), ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p), ("lse2", C.c_void_p),
        ("hpre", C.c_void_p), ("T", C.c_void_p),
        ("Vt", C.c_void_p), ("hact", C.c_void_p),
        ("gemm_ws", C.c_void_p), ("gemm_ws_bytes", C.c_int64),
        ("x_out", C.c_void_p),
    ]


class FluxSingleBwdArgs(C.Structure):
    """st355_flux_single_bwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("K2", C.c_int32), ("k2_real", C.c_int32), ("n_targets", C.c_int32),
        ("rank", C.c_int32), ("r_pad", C.c_int32), ("accumulate", C.c_int32),
        ("scale", C.c_float), ("lora_scale", C.c_float),
        ("x", C.c_void_p), ("n", C.c_void_p), ("V", C.c_void_p), ("rrms", C.c_void_p), ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p),
        ("lse2", C.c_void_p), ("hpre", C.c_void_p), ("T", C.c_void_p),
        ("mod_scale", C.c_void_p), ("mod_gate", C.c_void_p), ("mod_stride", C.c_int64),
        ("gate_prev", C.c_void_p),
        ("wT_qkv", C.c_void_p), ("wT_mlp", C.c_void_p), ("wT_out", C.c_void_p),
        ("A_cat_T", C.c_void_p), ("B_blk_T", C.c_void_p),
        ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("cos_p", C.c_void_p), ("sin_p", C.c_void_p), ("key_bias", C.c_void_p),
        ("dx", C.c_void_p), ("dxg", C.c_void_p),
        ("gA", C.c_void_p * 4), ("gB", C.c_void_p * 4),
        ("g", C.c_void_p), ("dO", C.c_void_p), ("dhpre", C.c_void_p), ("dn_mlp", C.c_void_p), ("dqkv", C.c_void_p), ("U", C.c_void_p), ("dn", C.c_void_p),
        ("gemm_ws", C.c_void_p), ("gemm_ws_bytes", C.c_int64), ("attn_ws", C.c_void_p), ("skinny_ws", C.c_void_p),
        ("dx_out", C.c_void_p), ("dxg_out", C.c_void_p),
    ]


class FluxDoubleFwdArgs(C.Structure):
    """st355_flux_double_fwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("Si", C.c_int32), ("St", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("K2_qkv", C.c_int32),
        ("k2r_qkv", C.c_int32), ("K2_out", C.c_int32), ("k2r_out", C.c_int32),
        ("scale", C.c_float),
        ("img", C.c_void_p), ("txt", C.c_void_p), ("mod_img", C.c_void_p), ("mod_txt", C.c_void_p),
        ("mod_stride", C.c_int64),
        ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_add_qkv", C.c_void_p), ("b_add_qkv", C.c_void_p), ("A_qkv", C.c_void_p), ("Bb_qkv", C.c_void_p),
        ("A_out", C.c_void_p), ("Bb_out", C.c_void_p), ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("norm_added_q", C.c_void_p), ("norm_added_k", C.c_void_p),
        ("w_out", C.c_void_p), ("b_out", C.c_void_p), ("w_add_out", C.c_void_p), ("b_add_out", C.c_void_p), ("w_ff1", C.c_void_p), ("b_ff1", C.c_void_p),
        ("w_ff2", C.c_void_p), ("b_ff2", C.c_void_p), ("w_ffc1", C.c_void_p), ("b_ffc1", C.c_void_p), ("w_ffc2", C.c_void_p), ("b_ffc2", C.c_void_p),
        ("cos_p", C.c_void_p), ("sin_p", C.c_void_p), ("key_bias", C.c_void_p), ("n_img", C.c_void_p), ("n_txt", C.c_void_p), ("V", C.c_void_p),
        ("rrms", C.c_void_p), ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p), ("lse2", C.c_void_p), ("x1_img", C.c_void_p),
        ("x1_txt", C.c_void_p), ("hpre_img", C.c_void_p), ("hpre_txt", C.c_void_p), ("T_img", C.c_void_p), ("T_o", C.c_void_p), ("Vt", C.c_void_p),
        ("n2_img", C.c_void_p), ("n2_txt", C.c_void_p), ("h_img", C.c_void_p), ("h_txt", C.c_void_p), ("gemm_ws", C.c_void_p),
        ("gemm_ws_bytes", C.c_int64),
        ("out_img", C.c_void_p), ("out_txt", C.c_void_p), ("out_joint", C.c_void_p),
    ]


class FluxDoubleBwdArgs(C.Structure):
    """st355_flux_double_bwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("Si", C.c_int32), ("St", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("K2_qkv", C.c_int32),
        ("k2r_qkv", C.c_int32), ("K2_out", C.c_int32), ("k2r_out", C.c_int32), ("rank_qkv", C.c_int32), ("rpad_qkv", C.c_int32), ("rank_out", C.c_int32),
        ("rpad_out", C.c_int32), ("accumulate", C.c_int32),
        ("scale", C.c_float), ("scale_qkv", C.c_float), ("scale_out", C.c_float),
        ("img", C.c_void_p), ("txt", C.c_void_p), ("n_img", C.c_void_p), ("V", C.c_void_p), ("rrms", C.c_void_p), ("Q", C.c_void_p),
        ("K", C.c_void_p), ("O", C.c_void_p), ("lse2", C.c_void_p), ("x1_img", C.c_void_p), ("x1_txt", C.c_void_p), ("hpre_img", C.c_void_p),
        ("hpre_txt", C.c_void_p), ("T_img", C.c_void_p), ("T_o", C.c_void_p), ("mod_img", C.c_void_p), ("mod_txt", C.c_void_p),
        ("mod_stride", C.c_int64),
        ("wT_qkv", C.c_void_p), ("wT_add_qkv", C.c_void_p), ("wT_out", C.c_void_p), ("wT_add_out", C.c_void_p), ("wT_ff1", C.c_void_p), ("wT_ff2", C.c_void_p),
        ("wT_ffc1", C.c_void_p), ("wT_ffc2", C.c_void_p), ("At_qkv", C.c_void_p), ("Bbt_qkv", C.c_void_p), ("At_out", C.c_void_p), ("Bbt_out", C.c_void_p),
        ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("norm_added_q", C.c_void_p), ("norm_added_k", C.c_void_p), ("cos_p", C.c_void_p), ("sin_p", C.c_void_p),
        ("key_bias", C.c_void_p), ("d_img", C.c_void_p), ("d_txt", C.c_void_p),
        ("gA_qkv", C.c_void_p * 4), ("gB_qkv", C.c_void_p * 4), ("gA_out", C.c_void_p * 4), ("gB_out", C.c_void_p * 4),
        ("g_img", C.c_void_p), ("g_txt", C.c_void_p), ("dh_img", C.c_void_p), ("dh_txt", C.c_void_p), ("dn2_img", C.c_void_p), ("dn2_txt", C.c_void_p),
        ("dx1_img", C.c_void_p), ("dx1g_img", C.c_void_p), ("dx1_txt", C.c_void_p), ("dx1g_txt", C.c_void_p), ("dO", C.c_void_p), ("dqkv", C.c_void_p),
        ("U_qkv", C.c_void_p), ("U_out", C.c_void_p), ("dn_img", C.c_void_p), ("dn_txt", C.c_void_p), ("gemm_ws", C.c_void_p),
        ("gemm_ws_bytes", C.c_int64),
        ("attn_ws", C.c_void_p), ("skinny_ws", C.c_void_p), ("d_img_out", C.c_void_p), ("d_txt_out", C.c_void_p),
    ]


class PixartBlockFwdArgs(C.Structure):
    """st355_pixart_block_fwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("Sk", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("d_pad", C.c_int32), ("scale", C.c_float),
        ("h", C.c_void_p), ("ctx", C.c_void_p), ("mod", C.c_void_p), ("mod_stride", C.c_int64), ("key_bias", C.c_void_p), ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p),
        ("w_out1", C.c_void_p), ("b_out1", C.c_void_p), ("w_q2", C.c_void_p), ("b_q2", C.c_void_p), ("w_kv2", C.c_void_p), ("b_kv2", C.c_void_p), ("w_out2", C.c_void_p),
        ("b_out2", C.c_void_p), ("w_ff1", C.c_void_p), ("b_ff1", C.c_void_p), ("w_ff2", C.c_void_p), ("b_ff2", C.c_void_p), ("n1", C.c_void_p), ("qkv", C.c_void_p),
        ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p), ("lse", C.c_void_p), ("ya", C.c_void_p), ("h1", C.c_void_p), ("q2", C.c_void_p),
        ("kv", C.c_void_p), ("Q2", C.c_void_p), ("K2", C.c_void_p), ("O2", C.c_void_p), ("lse_x", C.c_void_p), ("h2", C.c_void_p), ("n2", C.c_void_p),
        ("pre", C.c_void_p), ("act", C.c_void_p), ("yf", C.c_void_p), ("Vt", C.c_void_p), ("V2t", C.c_void_p), ("out", C.c_void_p),
    ]


class PixartBlockBwdArgs(C.Structure):
    """st355_pixart_block_bwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("Sk", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("d_pad", C.c_int32), ("scale", C.c_float),
        ("h", C.c_void_p), ("mod", C.c_void_p), ("mod_stride", C.c_int64), ("key_bias", C.c_void_p), ("wT_qkv", C.c_void_p), ("wT_out1", C.c_void_p), ("wT_q2", C.c_void_p),
        ("wT_out2", C.c_void_p), ("wT_ff1", C.c_void_p), ("wT_ff2", C.c_void_p), ("qkv", C.c_void_p), ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p),
        ("lse", C.c_void_p), ("q2", C.c_void_p), ("kv", C.c_void_p), ("Q2", C.c_void_p), ("K2", C.c_void_p), ("O2", C.c_void_p), ("lse_x", C.c_void_p),
        ("h2", C.c_void_p), ("pre", C.c_void_p), ("d_out", C.c_void_p), ("dyf", C.c_void_p), ("dpre", C.c_void_p), ("dn2", C.c_void_p), ("d2", C.c_void_p),
        ("dO2", C.c_void_p), ("dq2", C.c_void_p), ("dkv", C.c_void_p), ("d1", C.c_void_p), ("dya", C.c_void_p), ("dO", C.c_void_p), ("dqkv", C.c_void_p),
        ("dn1", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("attn_ws", C.c_void_p), ("d_in", C.c_void_p),
    ]


class Sd3JointFwdArgs(C.Structure):
    """st355_sd3_joint_fwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("Si", C.c_int32), ("St", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("hd", C.c_int32),
        ("last", C.c_int32), ("K2_qkv", C.c_int32), ("k2r_qkv", C.c_int32), ("K2_aqkv", C.c_int32), ("k2r_aqkv", C.c_int32), ("K2_out", C.c_int32),
        ("k2r_out", C.c_int32), ("K2_aout", C.c_int32), ("k2r_aout", C.c_int32), ("scale", C.c_float), ("img", C.c_void_p), ("txt", C.c_void_p),
        ("mod_img", C.c_void_p), ("mod_txt", C.c_void_p), ("mod_stride", C.c_int64), ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_add_qkv", C.c_void_p),
        ("b_add_qkv", C.c_void_p), ("w_out", C.c_void_p), ("b_out", C.c_void_p), ("w_add_out", C.c_void_p), ("b_add_out", C.c_void_p), ("w_ff1", C.c_void_p),
        ("b_ff1", C.c_void_p), ("w_ff2", C.c_void_p), ("b_ff2", C.c_void_p), ("w_ffc1", C.c_void_p), ("b_ffc1", C.c_void_p), ("w_ffc2", C.c_void_p),
        ("b_ffc2", C.c_void_p), ("A_qkv", C.c_void_p), ("Bb_qkv", C.c_void_p), ("A_aqkv", C.c_void_p), ("Bb_aqkv", C.c_void_p), ("A_out", C.c_void_p),
        ("Bb_out", C.c_void_p), ("A_aout", C.c_void_p), ("Bb_aout", C.c_void_p), ("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("norm_added_q", C.c_void_p),
        ("norm_added_k", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p), ("n_img", C.c_void_p), ("n_txt", C.c_void_p), ("qkv", C.c_void_p),
        ("Q", C.c_void_p), ("K", C.c_void_p), ("O", C.c_void_p), ("lse2", C.c_void_p), ("x1_img", C.c_void_p), ("x1_txt", C.c_void_p),
        ("hpre_img", C.c_void_p), ("hpre_txt", C.c_void_p), ("T_img", C.c_void_p), ("T_txt", C.c_void_p), ("T_o", C.c_void_p), ("T_ao", C.c_void_p),
        ("ya_img", C.c_void_p), ("ya_txt", C.c_void_p), ("yf_img", C.c_void_p), ("yf_txt", C.c_void_p), ("n2_img", C.c_void_p), ("n2_txt", C.c_void_p),
        ("h_img", C.c_void_p), ("h_txt", C.c_void_p), ("Vt", C.c_void_p), ("c_img", C.c_void_p), ("c_txt", C.c_void_p), ("gemm_ws", C.c_void_p),
        ("gemm_ws_bytes", C.c_int64), ("out_img", C.c_void_p), ("out_txt", C.c_void_p),
    ]


class Sd3JointBwdArgs(C.Structure):
    """st355_sd3_joint_bwd_args (include/st355.h)"""
    _fields_ = [
        ("B", C.c_int32), ("Si", C.c_int32), ("St", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("hd", C.c_int32),
        ("last", C.c_int32), ("need_input_grads", C.c_int32), ("K2_qkv", C.c_int32), ("k2r_qkv", C.c_int32), ("K2_aqkv", C.c_int32), ("k2r_aqkv", C.c_int32),
        ("K2_out", C.c_int32), ("k2r_out", C.c_int32), ("K2_aout", C.c_int32), ("k2r_aout", C.c_int32), ("scale", C.c_float), ("img", C.c_void_p),
        ("txt", C.c_void_p), ("mod_img", C.c_void_p), ("mod_txt", C.c_void_p), ("mod_stride", C.c_int64), ("qkv", C.c_void_p), ("Q", C.c_void_p),
        ("K", C.c_void_p), ("O", C.c_void_p), ("lse2", C.c_void_p), ("x1_img", C.c_void_p), ("x1_txt", C.c_void_p), ("hpre_img", C.c_void_p),
        ("hpre_txt", C.c_void_p), ("wT_qkv", C.c_void_p), ("wT_add_qkv", C.c_void_p), ("wT_out", C.c_void_p), ("wT_add_out", C.c_void_p), ("wT_ff1", C.c_void_p),
        ("wT_ff2", C.c_void_p), ("wT_ffc1", C.c_void_p), ("wT_ffc2", C.c_void_p), ("At_qkv", C.c_void_p), ("Bbt_qkv", C.c_void_p), ("At_aqkv", C.c_void_p),
        ("Bbt_aqkv", C.c_void_p), ("At_out", C.c_void_p), ("Bbt_out", C.c_void_p), ("At_aout", C.c_void_p), ("Bbt_aout", C.c_void_p), ("norm_q", C.c_void_p),
        ("norm_k", C.c_void_p), ("norm_added_q", C.c_void_p), ("norm_added_k", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p), ("d_img", C.c_void_p),
        ("d_txt", C.c_void_p), ("g_img", C.c_void_p), ("g_txt", C.c_void_p), ("dh_img", C.c_void_p), ("dh_txt", C.c_void_p), ("dn2_img", C.c_void_p),
        ("dn2_txt", C.c_void_p), ("dx1_img", C.c_void_p), ("dx1g_img", C.c_void_p), ("dx1_txt", C.c_void_p), ("dx1g_txt", C.c_void_p), ("U_o", C.c_void_p),
        ("U_ao", C.c_void_p), ("dO", C.c_void_p), ("dqkv", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("U_qkv", C.c_void_p),
        ("U_aqkv", C.c_void_p), ("dn_img", C.c_void_p), ("dn_txt", C.c_void_p), ("c_img", C.c_void_p), ("c_txt", C.c_void_p), ("gemm_ws", C.c_void_p),
        ("gemm_ws_bytes", C.c_int64), ("attn_ws", C.c_void_p), ("d_img_out", C.c_void_p), ("d_txt_out", C.c_void_p),
        ("dmod_img", C.c_void_p), ("dmod_txt", C.c_void_p), ("dmod_stride", C.c_int64), ("ya_img", C.c_void_p), ("ya_txt", C.c_void_p), ("yf_img", C.c_void_p),
        ("yf_txt", C.c_void_p), ("gb_ff2", C.c_void_p), ("gb_ff1", C.c_void_p), ("gb_out", C.c_void_p), ("gb_qkv", C.c_void_p), ("gb_ffc2", C.c_void_p),
        ("gb_ffc1", C.c_void_p), ("gb_add_out", C.c_void_p), ("gb_add_qkv", C.c_void_p), ("stats_ws", C.c_void_p),
    ]


class StatOut(C.Structure):
    """st355_stat_out (include/st355.h): one destination of a fused column sum"""
    _fields_ = [("out", C.c_void_p), ("stride", C.c_int64), ("reduce_batches", C.c_int32), ("out_bf16", C.c_int32), ("accumulate", C.c_int32), ("_pad", C.c_int32)]


class VaeEncoder(C.Structure):
    """st355_vae_encoder (include/st355.h): architecture + the device-pointer table of st355_vae_encode"""
    _fields_ = [
        ("in_channels", C.c_int32), ("latent_channels", C.c_int32), ("n_levels", C.c_int32), ("layers_per_block", C.c_int32), ("norm_num_groups", C.c_int32),
        ("block_out_channels", C.c_int32 * 8), ("n_tensors", C.c_int32), ("tensors", C.POINTER(C.c_void_p)),
    ]


class Heads(C.Structure):
    """st355_heads (include/st355.h): destinations of the head-splitting projection epilogue ST355_EPI_HEADS"""
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("H", C.c_int32), ("S", C.c_int32), ("pos0", C.c_int32), ("Sp", C.c_int32),
                ("n_q", C.c_int32), ("n_k", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("A2", C.c_void_p), ("lda2", C.c_int64),
        ("B2", C.c_void_p), ("ldb2", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K2", C.c_int32),
        ("bias", C.c_void_p),
        ("epilogue", C.c_int32),
        ("aux_out", C.c_void_p), ("ld_aux_out", C.c_int64),
        ("aux_in", C.c_void_p), ("ld_aux_in", C.c_int64),
        ("gate", C.c_void_p), ("gate_stride", C.c_int64), ("rows_per_batch", C.c_int64),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("K2_real", C.c_int32),
        ("seg_rows", C.c_int64), ("seg_a", C.c_int64), ("seg_a2", C.c_int64), ("seg_c", C.c_int64), ("seg_in", C.c_int64), ("seg_out", C.c_int64),
        ("rope", C.POINTER(QkRope)),
        ("heads", C.POINTER(Heads)),
        ("tile_flags", C.c_void_p),
    ]


_lib = None


def _declare(lib):
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
    sz = C.c_size_t
    sig = {
        "st355_version": (C.c_int, []),
        "st355_arch": (C.c_char_p, []),
        "st355_last_error": (C.c_char_p, []),
        "st355_prof_enable": (C.c_int, [i32]),
        "st355_prof_reset": (C.c_int, []),
        "st355_prof_collect": (C.c_int, [vp, vp, vp, vp, i32]),
        "st355_prof_dump": (C.c_int, [C.c_char_p]),
        "st355_flow_noise_mix": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, u64, u64]),
        "st355_ddpm_noise_mix": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64]),
        "st355_mse_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, f32]),
        "st355_cond_loss": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, f32]),
        "st355_cond_loss_masked": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, i64, vp, vp, vp, i64, i64, f32]),
        "st355_flux_pack": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
        "st355_flux_unpack": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
        "st355_patchify": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32]),
        "st355_unpatchify": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32]),
        "st355_timestep_proj": (C.c_int, [vp, vp, vp, i32, i32, f32]),
        "st355_silu_bwd": (C.c_int, [vp, vp, vp, vp, i64]),
        "st355_silu": (C.c_int, [vp, vp, vp, i64]),
        "st355_gelu_tanh": (C.c_int, [vp, vp, vp, i64]),
        "st355_add": (C.c_int, [vp, vp, vp, vp, i64]),
        "st355_gather_rows": (C.c_int, [vp, vp, i64, i64, vp, vp, i64, i64, i32, i32, i32]),
        "st355_scatter_rows": (C.c_int, [vp, vp, i64, i64, vp, vp, i64, i64, i32, i32, i32]),
        "st355_scale_cols": (C.c_int, [vp, vp, i64, vp, i64, i64, vp, i64, i64, i64]),
        "st355_gemm_bf16": (C.c_int, [vp, C.POINTER(GemmArgs)]),
        "st355_gemm_bf16_grouped": (C.c_int, [vp, C.POINTER(GemmArgs), i32]),
        "st355_gemm_set_persistent": (C.c_int, [i32]),
        "st355_gemm_set_tail_split": (C.c_int, [i32]),
        "st355_gemm_tail_placement": (C.c_int, []),
        "st355_colsum_workspace": (sz, [i64, i32, i64]),
        "st355_colsum_prod": (C.c_int, [vp, vp, i64, vp, i64, i64, i32, i64, vp, i64, i32, vp, i64, vp, vp, i64, i32, vp]),
        "st355_stats_workspace": (sz, [i64, i32, i64, i32]),
        "st355_ln_modulate_bwd_stats": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, f32, vp, i64,
                                                  C.POINTER(StatOut), C.POINTER(StatOut), C.POINTER(StatOut), C.POINTER(StatOut), vp]),
        "st355_scale_cols_stats": (C.c_int, [vp, vp, i64, vp, i64, i64, vp, i64, i64, i32, vp, i64, C.POINTER(StatOut), C.POINTER(StatOut), vp]),
        "st355_colsum_rows": (C.c_int, [vp, vp, i64, i64, i64, i32, i32, C.POINTER(StatOut), vp]),
        "st355_transpose_bf16": (C.c_int, [vp, vp, i64, vp, i64, i32, i32]),
        "st355_sum_chunks_bf16": (C.c_int, [vp, vp, i32, i64, vp]),
        "st355_fp8_quantize_weight": (C.c_int, [vp, vp, i64, vp, vp, i32, i32]),
        "st355_fp8_quantize_act": (C.c_int, [vp, vp, i64, vp, vp, i64, i32, vp]),
        "st355_linear_fp8": (C.c_int, [vp, vp, i64, vp, vp, i64, vp, vp, vp, i64, i32, i32, i32]),
        "st355_gemm_tn_bf16": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, vp, i64]),
        "st355_gemm_tn_seg_bf16": (C.c_int, [vp, vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i32, i32, i32, vp, i64]),
        "st355_skinny_tn_workspace": (sz, [i64, i64, i32]),
        "st355_skinny_tn": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, i32, f32, i32, vp]),
        "st355_skinny_tn_seg": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, i32, f32, i32, vp, i64, i64, i64]),
        "st355_skinny_tn_multi": (C.c_int, [vp, vp, i64, vp, i64, vp, i32, i64, i64, i64, i64, i32, f32, i32, vp, i64, i64, i64]),
        "st355_ln_modulate_fwd": (C.c_int, [vp, vp, i64, vp, vp, i64, i64, vp, i64, i64, i32, f32]),
        "st355_ln_modulate_bwd": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, f32]),
        "st355_qk_norm_rope_fwd": (C.c_int, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32]),
        "st355_qk_norm_rope_bwd": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32]),
        "st355_qk_norm_wgrad_workspace": (sz, [i32, i32, i32, i32]),
        "st355_qk_norm_rope_bwd_wgrad": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp]),
        "st355_qk_rope_norm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32]),
        "st355_attn_set_impl": (C.c_int, [i32, i32, i32]),
        "st355_attn_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, f32]),
        "st355_attn_fwd_vrows": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, f32]),
        "st355_attn_bwd_rope": (C.c_int, [vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp]),
        "st355_attn_bwd_workspace": (sz, [i32, i32, i32, i32, i32]),
        "st355_attn_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i64,
                                      i32, i32, i32, i32, i32, f32, vp]),
        "st355_adamw_ema_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, f32]),
        "st355_adamw_ema_step_bf16": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, f32]),
        "st355_adamw_bf16_sr_step": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, i32,
                                               vp, u64, u64, f32]),
        "st355_ema_update": (C.c_int, [vp, vp, vp, i64, f32, i32]),
        "st355_grad_norm": (C.c_int, [vp, vp, i64, i32, vp]),
        "st355_grad_norm_ws": (C.c_int, [vp, vp, i64, i32, vp, vp]),
        "st355_grad_clamp": (C.c_int, [vp, vp, i64, i32, f32]),
        "st355_grad_clip_norm": (C.c_int, [vp, vp, i64, i32, vp, f32, f32]),
        "st355_lora_pack": (C.c_int, [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, i32, i32, i32, i32]),
        "st355_workspace_bytes": (i64, [i32, vp, i32]),
        "st355_comm_unique_id": (C.c_int, [vp]),
        "st355_comm_init": (C.c_int, [vp, vp, i32, i32]),
        "st355_comm_destroy": (C.c_int, [vp]),
        "st355_comm_all_reduce": (C.c_int, [vp, vp, vp, i64, i32]),
        "st355_comm_reduce_scatter": (C.c_int, [vp, vp, vp, vp, i64, i32]),
        "st355_comm_all_gather": (C.c_int, [vp, vp, vp, vp, i64, i32]),
        "st355_conv_grid_rows": (i64, [i32, i32, i32]),
        "st355_conv_bf16": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32]),
        "st355_conv_wgrad_bf16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i64]),
        "st355_grid_from_nchw": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32]),
        "st355_grid_to_nchw": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32]),
        "st355_im2col3x3": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32]),
        "st355_col2im3x3": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32]),
        "st355_softmax_rows": (C.c_int, [vp, vp, i64, i64, i32, f32]),
        "st355_vae_encode_workspace": (sz, [C.POINTER(VaeEncoder), i32, i32, i32]),
        "st355_vae_encode": (C.c_int, [vp, C.POINTER(VaeEncoder), vp, vp, i32, i32, i32, vp, sz]),
        "st355_block_flux_single_fwd": (C.c_int, [vp, C.POINTER(FluxSingleFwdArgs)]),
        "st355_block_flux_single_bwd": (C.c_int, [vp, C.POINTER(FluxSingleBwdArgs)]),
        "st355_block_flux_double_fwd": (C.c_int, [vp, C.POINTER(FluxDoubleFwdArgs)]),
        "st355_block_flux_double_bwd": (C.c_int, [vp, C.POINTER(FluxDoubleBwdArgs)]),
        "st355_block_pixart_fwd": (C.c_int, [vp, C.POINTER(PixartBlockFwdArgs)]),
        "st355_block_pixart_bwd": (C.c_int, [vp, C.POINTER(PixartBlockBwdArgs)]),
        "st355_block_sd3_joint_fwd": (C.c_int, [vp, C.POINTER(Sd3JointFwdArgs)]),
        "st355_block_sd3_joint_bwd": (C.c_int, [vp, C.POINTER(Sd3JointBwdArgs)]),
        "st355_upsample2x": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
        "st355_upsample2x_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
        "st355_tokens_to_grid": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32]),
        "st355_grid_to_tokens": (C.c_int, [vp, vp, vp, i32, i32, i32, i32]),
        "st355_groupnorm_workspace": (sz, [i32, i32, i32, i32]),
        "st355_groupnorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp]),
        "st355_groupnorm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
        "st355_layernorm_fwd": (C.c_int, [vp, vp, i64, vp, vp, vp, i64, i64, i32, f32]),
        "st355_layernorm_bwd": (C.c_int, [vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i32, f32]),
        "st355_layernorm_param_grads_workspace": (sz, [i32]),
        "st355_layernorm_param_grads": (C.c_int, [vp, vp, i64, vp, i64, i64, i32, f32, vp, vp, i32, vp]),
        "st355_geglu_fwd": (C.c_int, [vp, vp, i64, vp, i64, i32]),
        "st355_geglu_bwd": (C.c_int, [vp, vp, i64, vp, vp, i64, i64, i32]),
        "st355_head_split": (C.c_int, [vp, vp, i64, vp, vp, i32, i32, i32, i32, i32]),
        "st355_head_merge": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, i32]),
        "st355_head_split_pad": (C.c_int, [vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32]),
        "st355_head_merge_pad": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, i32]),
        "st355_softmax_rows_bwd": (C.c_int, [vp, vp, vp, i64, i64, i32, f32]),
        "st355_attn_cross_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, f32]),
        "st355_attn_cross_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i64,
                                            i32, i32, i32, i32, i32, i32, i32, f32, vp]),
        "st355_attn_fwd_res": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, f32]),
        "st355_attn_bwd_res": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, i64,
                                          i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load (once) and return the ctypes handle.  Raises St355Unavailable if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("ST355_LIB", LIB_PATH))
    if not path.exists():
        raise St355Unavailable(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950).  There is no CPU/eager fallback for the train step."
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise St355Unavailable(f"cannot load {path}: {e}") from e
    _lib = _declare(lib)
    return _lib


def is_built() -> bool:
    return Path(os.environ.get("ST355_LIB", LIB_PATH)).exists()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().st355_last_error().decode("utf-8", "replace")
        raise St355Error(f"{what or 'st355 call'} failed (rc={rc}): {msg}")
