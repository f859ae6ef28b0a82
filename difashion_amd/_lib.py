"""ctypes binding of libdifashion_hip.so (C ABI: include/difashion_hip.h).

There is NO fallback: if the HIP library is missing, import of the compute path raises.  PyTorch is
used on the host side only for device memory, streams and torch.distributed (plumbing).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# DFH_LIB=<path>: load another build of the library (same-box A/B probes); the product path never sets it
LIB_PATH = os.environ.get("DFH_LIB") or os.path.join(CSRC, "libdifashion_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "difashion_hip.h")

DFH_MAX_BLOCKS = 4


class DfhError(RuntimeError):
    pass


class UNetConfigC(C.Structure):
    _fields_ = [
        ("sample_size", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int), ("num_blocks", C.c_int),
        ("block_out_channels", C.c_int * DFH_MAX_BLOCKS), ("layers_per_block", C.c_int),
        ("cross_attention_dim", C.c_int), ("num_heads", C.c_int * DFH_MAX_BLOCKS),
        ("down_attn", C.c_int * DFH_MAX_BLOCKS), ("use_linear_projection", C.c_int),
        ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("text_len", C.c_int),
    ]


class VAEConfigC(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int), ("num_blocks", C.c_int),
                ("block_out_channels", C.c_int * DFH_MAX_BLOCKS), ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int)]


class CLIPConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_hidden_layers", C.c_int),
                ("num_attention_heads", C.c_int), ("max_position_embeddings", C.c_int), ("hidden_act", C.c_int),
                ("layer_norm_eps", C.c_float)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("conv_src", C.c_void_p), ("conv_c", C.c_int), ("conv", C.c_int),
        ("batch", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("stride", C.c_int), ("upsample", C.c_int),
        ("a0", C.c_void_p), ("a0_c", C.c_int), ("a1", C.c_void_p), ("a1_c", C.c_int),
        ("W", C.c_void_p), ("ldw", C.c_int),
        ("M", C.c_int), ("N", C.c_int),
        ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("rv_ld", C.c_int), ("rv_off", C.c_int), ("rows_per_b", C.c_int),
        ("resid", C.c_void_p), ("ld_res", C.c_int),
        ("act", C.c_int),
        ("out", C.c_void_p), ("ld_out", C.c_int), ("out_mode", C.c_int),
        ("partial", C.c_void_p), ("partial_floats", C.c_size_t),
        ("zero_page", C.c_void_p),
        ("force_tile", C.c_int), ("force_split", C.c_int), ("force_order", C.c_int),
        ("gstat", C.c_void_p), ("gstat_cpg", C.c_int), ("gstat_hw", C.c_int),
        ("w_img_stride", C.c_size_t),
    ]


class Fp8GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int),
        ("sA", C.c_void_p), ("sa_div", C.c_int), ("sa_mul", C.c_float),
        ("sx", C.c_void_p),
        ("W", C.c_void_p), ("sW", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("ld_res", C.c_int), ("act", C.c_int),
        ("out", C.c_void_p), ("ld_out", C.c_int), ("out_mode", C.c_int), ("out_sx", C.c_void_p), ("rows_per_b", C.c_int), ("amax", C.c_void_p),
        ("zero_page", C.c_void_p),
    ]


class StepCoef(C.Structure):
    _fields_ = [("kind", C.c_int), ("vpred", C.c_int), ("sqrt_a_t", C.c_float), ("sqrt_b_t", C.c_float),
                ("sqrt_a_prev", C.c_float), ("dir_coef", C.c_float), ("std_dev", C.c_float)]


class ProfClass(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); status-returning functions are wrapped to raise DfhError
SIGNATURES = {
    "dfh_abi_version": (_i, []),
    "dfh_last_error": (C.c_char_p, []),
    "dfh_build_info": (C.c_char_p, []),
    "dfh_prof_begin": (_i, []),
    "dfh_prof_end": (_i, [C.POINTER(ProfClass), _i]),
    "dfh_census_reset": (None, []),
    "dfh_census_count": (_i, []),
    "dfh_census_name": (C.c_char_p, [_i]),
    "dfh_census_get": (C.c_long, [_i]),
    "dfh_unet_create": (_i, [C.POINTER(UNetConfigC), C.POINTER(_vp)]),
    "dfh_unet_destroy": (None, [_vp]),
    "dfh_unet_num_params": (_i, [_vp]),
    "dfh_unet_param_name": (C.c_char_p, [_vp, _i]),
    "dfh_unet_param_ndim": (_i, [_vp, _i]),
    "dfh_unet_param_dim": (_i, [_vp, _i, _i]),
    "dfh_unet_arena16_bytes": (_sz, [_vp]),
    "dfh_unet_arena32_bytes": (_sz, [_vp]),
    "dfh_unet_workspace_bytes": (_sz, [_vp, _i]),
    "dfh_unet_bind": (_i, [_vp, _vp, _vp, _vp, _sz, _i]),
    "dfh_unet_pack": (_i, [_vp, C.POINTER(_vp), _i, _vp]),
    "dfh_unet_forward": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "dfh_unet_run_cache_bytes": (_sz, [_vp, _i, _i]),
    "dfh_unet_run_cache": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "dfh_unet_set_dup_tail": (_i, [_vp, _i]),
    "dfh_unet_forward_cached": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "dfh_unet_debug_tap": (_i, [_vp, C.c_char_p, _vp, _sz, _vp]),
    "dfh_clip_create": (_i, [C.POINTER(CLIPConfigC), C.POINTER(_vp)]),
    "dfh_clip_destroy": (None, [_vp]),
    "dfh_clip_num_params": (_i, [_vp]),
    "dfh_clip_param_name": (C.c_char_p, [_vp, _i]),
    "dfh_clip_param_ndim": (_i, [_vp, _i]),
    "dfh_clip_param_dim": (_i, [_vp, _i, _i]),
    "dfh_clip_workspace_bytes": (_sz, [_vp, _i, _i]),
    "dfh_clip_encode": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _vp]),
    "dfh_vae_create": (_i, [C.POINTER(VAEConfigC), C.POINTER(_vp)]),
    "dfh_vae_destroy": (None, [_vp]),
    "dfh_vae_num_params": (_i, [_vp]),
    "dfh_vae_param_name": (C.c_char_p, [_vp, _i]),
    "dfh_vae_param_ndim": (_i, [_vp, _i]),
    "dfh_vae_param_dim": (_i, [_vp, _i, _i]),
    "dfh_vae_arena16_bytes": (_sz, [_vp]),
    "dfh_vae_arena32_bytes": (_sz, [_vp]),
    "dfh_vae_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "dfh_vae_bind": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "dfh_vae_pack": (_i, [_vp, C.POINTER(_vp), _i, _vp]),
    "dfh_vae_encode": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "dfh_vae_decode": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "dfh_unet_arena16t_bytes": (_sz, [_vp]),
    "dfh_unet_grad16_bytes": (_sz, [_vp]),
    "dfh_unet_grad32_bytes": (_sz, [_vp]),
    "dfh_unet_train_workspace_bytes": (_sz, [_vp, _i]),
    "dfh_unet_bind_train": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i]),
    "dfh_unet_pack_train": (_i, [_vp, C.POINTER(_vp), _i, _vp]),
    "dfh_unet_pack_all": (_i, [_vp, C.POINTER(_vp), _i, _vp]),
    "dfh_unet_grad_sumsq": (_i, [_vp, _vp]),
    "dfh_unet_forward_train": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "dfh_unet_backward": (_i, [_vp, _vp, _vp, C.POINTER(_vp), _i, _i, _vp]),
    "dfh_unet_backward_begin": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "dfh_unet_backward_next": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz), _vp]),
    "dfh_unet_backward_finish": (_i, [_vp, C.POINTER(_vp), _i, _i, _vp]),
    "dfh_gemm_partial_floats": (_sz, [C.POINTER(GemmDesc)]),
    "dfh_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "dfh_gemm_wgrad": (_i, [C.POINTER(GemmDesc), _vp, _i, _vp, _i, _i, _vp]),
    "dfh_gemm_wgrad_partial_floats": (_sz, [C.POINTER(GemmDesc), _i]),
    "dfh_groupnorm_fold": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, C.c_float, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dfh_gemm_wgrad_plan": (_i, [C.POINTER(GemmDesc), _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfh_colsum": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "dfh_groupnorm": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "dfh_gemm_gstat": (_i, [C.POINTER(GemmDesc), _vp, C.POINTER(C.c_int)]),
    "dfh_groupnorm_pre": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _vp, _vp]),
    "dfh_groupnorm_stats": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "dfh_groupnorm_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "dfh_attention_lse": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "dfh_attention_delta": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "dfh_attention_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "dfh_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp]),
    "dfh_pack_matrix_t": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dfh_pack_conv3x3_t": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dfh_unpack_matrix": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dfh_unpack_conv3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dfh_unpack_vector": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "dfh_pool2x2_sum": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dfh_add_bf16": (_i, [_vp, _vp, _sz, _i, _vp]),
    "dfh_geglu_fwd": (_i, [_vp, _vp, _sz, _i, _vp]),
    "dfh_geglu_bwd": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "dfh_act_fwd": (_i, [_vp, _vp, _sz, _i, _vp]),
    "dfh_act_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _f, _vp]),
    "dfh_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "dfh_transpose_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _sz, _sz, _vp]),
    "dfh_mse_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "dfh_assemble_bwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "dfh_sumsq": (_i, [_vp, _sz, _vp, _vp]),
    "dfh_adamw": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _f, _vp]),
    "dfh_ema": (_i, [_vp, _vp, _sz, _f, _vp]),
    "dfh_adamw_ema": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp]),
    "dfh_wire_pack": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "dfh_wire_shard_mean": (_i, [_vp, _vp, _i, _sz, _vp]),
    "dfh_wire_unpack": (_i, [_vp, _vp, _sz, _vp]),
    "dfh_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "dfh_gemm_out2": (_i, [C.POINTER(GemmDesc), _vp, _i, _i, _vp]),
    "dfh_ups_phase_fold": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "dfh_conv_up2x": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "dfh_wino_weights": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "dfh_wino_blocked": (_i, [_i, _i]),
    "dfh_prof_saved_flops": (C.c_double, []),
    "dfh_wino_input": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dfh_gn_wino_input_ok": (_i, [_i, _i, _i, _i, _i]),
    "dfh_gn_wino_input_chain": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _i, _i, _i, _vp]),
    "dfh_gn_wino_input": (_i, [_vp, _i, _vp, _i, _vp, _vp, _f, _i, _vp, _i, _i, _i, _vp]),
    "dfh_conv3x3_wino_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "dfh_conv3x3_wino": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "dfh_gemm_batched": (_i, [C.POINTER(GemmDesc), _i, C.c_long, C.c_long, C.c_long, _vp]),
    "dfh_ln_fold": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dfh_gemm_ln": (_i, [C.POINTER(GemmDesc), _vp, C.POINTER(C.c_int), _vp, _i, _i, _f, _vp, _vp]),
    "dfh_quantize_rows_fp8": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp]),
    "dfh_layernorm_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "dfh_gemm_fp8": (_i, [C.POINTER(Fp8GemmDesc), _vp]),
    "dfh_mlp_fused_image_bytes": (_sz, []),
    "dfh_mlp_fused_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dfh_mlp_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "dfh_groupnorm_fp8": (_i, [_vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "dfh_attention_fp8out": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "dfh_amax_slabs": (_i, [_vp, C.c_long, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "dfh_attention_fp8": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "dfh_attn_scales": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dfh_unet_enable_fp8": (_i, [_vp]),
    "dfh_unet_enable_fp8_attention": (_i, [_vp, _i]),
    "dfh_unet_arena8_bytes": (_sz, [_vp]),
    "dfh_unet_bind_fp8": (_i, [_vp, _vp]),
    "dfh_attention": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "dfh_timestep_embedding": (_i, [_vp, _vp, _i, _i, _vp]),
    "dfh_nchw_to_nhwc_bf16": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "dfh_cast_f32_to_bf16": (_i, [_vp, _vp, _sz, _vp]),
    "dfh_pack_conv3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dfh_pack_matrix": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dfh_pack_vector": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dfh_mutual_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "dfh_assemble_input": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "dfh_cfg_step": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _f, _f, _f, C.POINTER(StepCoef), _vp]),
    "dfh_noise_mix": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dfh_mse_rows": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
}
_NO_STATUS = {"dfh_abi_version", "dfh_census_count", "dfh_unet_num_params", "dfh_unet_param_ndim", "dfh_unet_param_dim", "dfh_vae_num_params",
              "dfh_vae_param_ndim", "dfh_vae_param_dim", "dfh_clip_num_params", "dfh_clip_param_ndim", "dfh_clip_param_dim"}

_lib = None
_UNBOUND = set()
ABI_VERSION = 7          # == DFH_ABI_VERSION of include/difashion_hip.h (checked when the library is loaded)


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        raise DfhError("building libdifashion_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def raw():
    """The ctypes CDLL with prototypes set.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfhError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU / PyTorch fallback for the compute path)")
        lib = C.CDLL(LIB_PATH)
        # DFH_LIB=<path> loads another build; it is held to the same ABI check as the default path.  Only with the explicit
        # DFH_LIB_ALLOW_ABI_MISMATCH=1 (same-box A/B probes against LAST ROUND's library) are missing entry points left unbound and
        # the ABI number ignored -- and then call() refuses an unbound entry point instead of calling it through ctypes' default
        # int prototype (which truncates pointers and size_t).  Never on the product path.
        probe = bool(os.environ.get("DFH_LIB")) and os.environ.get("DFH_LIB_ALLOW_ABI_MISMATCH") == "1"
        unbound = set()
        for name, (res, args) in SIGNATURES.items():
            if probe and not hasattr(lib, name):
                unbound.add(name)
                continue
            fn = getattr(lib, name)       # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        got = lib.dfh_abi_version()
        if got != ABI_VERSION and not probe:
            raise DfhError(f"{LIB_PATH} reports ABI {got}, this Python side binds ABI {ABI_VERSION} (include/difashion_hip.h "
                           "DFH_ABI_VERSION): rebuild the library (`python -c 'import __graft_entry__ as g; g.build()'`)"
                           + ("; DFH_LIB_ALLOW_ABI_MISMATCH=1 loads it anyway for an A/B probe" if os.environ.get("DFH_LIB") else ""))
        global _UNBOUND
        _UNBOUND = unbound
        _lib = lib
    return _lib


def last_error() -> str:
    return raw().dfh_last_error().decode()


def call(name: str, *args):
    """Call a status-returning entry point; raise DfhError with the library's message on failure."""
    lib = raw()
    if name in _UNBOUND:
        raise DfhError(f"{name} is not exported by {LIB_PATH} (an older build loaded under DFH_LIB_ALLOW_ABI_MISMATCH=1)")
    rc = getattr(lib, name)(*args)
    if name not in _NO_STATUS and SIGNATURES[name][0] is _i and rc != 0:
        raise DfhError(f"{name} failed ({rc}): {last_error()}")
    return rc


def call_count(name: str, *args):
    """Call an entry point that returns a count (>= 0) or a negative status."""
    lib = raw()
    if name in _UNBOUND:
        raise DfhError(f"{name} is not exported by {LIB_PATH} (an older build loaded under DFH_LIB_ALLOW_ABI_MISMATCH=1)")
    rc = getattr(lib, name)(*args)
    if rc < 0:
        raise DfhError(f"{name} failed ({rc}): {last_error()}")
    return rc


_WEIGHT_EPOCH = 0


def weight_epoch() -> int:
    """Bumped by the fused optimizer after it rewrites master weights in place (torch version counters do not see a
    native kernel's writes); part of every pack signature."""
    return _WEIGHT_EPOCH


def bump_weight_epoch():
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def prof_begin():
    call("dfh_prof_begin")


def prof_end():
    """-> {class name: dict(launches, ms, flops, bytes)} since prof_begin (synchronises the device)."""
    arr = (ProfClass * 16)()
    n = raw().dfh_prof_end(arr, 16)
    if n < 0:
        raise DfhError(f"dfh_prof_end failed: {last_error()}")
    return {arr[i].name.decode(): dict(launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes)
            for i in range(n)}


def census_reset():
    raw().dfh_census_reset()


def census():
    """-> {kernel family: launches since census_reset()} (host-side counters of the launchers; test infrastructure)."""
    lib = raw()
    return {lib.dfh_census_name(i).decode(): int(lib.dfh_census_get(i)) for i in range(lib.dfh_census_count())}


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---- gradient epochs: FusedAdamW.zero_grad() opens an epoch; whatever writes a gradient (autograd accumulation, the native
#      U-Net backward) stamps the parameter with the epoch it wrote in; step() updates only parameters stamped since the last
#      zero_grad -- torch.optim.AdamW's "skip parameters whose .grad is None" for gradient views that are never None.
_GRAD_EPOCH = [1]


def grad_epoch() -> int:
    return _GRAD_EPOCH[0]


def next_grad_epoch() -> int:
    _GRAD_EPOCH[0] += 1
    return _GRAD_EPOCH[0]


def stamp_grads(params) -> None:
    e = _GRAD_EPOCH[0]
    for p in params:
        p._dfh_grad_epoch = e
