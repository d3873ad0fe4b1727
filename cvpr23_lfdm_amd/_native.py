"""ctypes binding of liblfdm_hip.so (C ABI: include/lfdm_hip.h).

The product path has no fallback: if the library is missing or a tensor is not on the GPU the
call raises.  (tests/emu/ can inject an x86 emulation build of the same kernels for logic checks
on a GPU-less box through `_set_library_for_tests`; that library never ships.)
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("LFDM_HIP_LIB", os.path.join(_PKG_DIR, "liblfdm_hip.so"))   # override: kernel probes only

f32p = C.c_void_p
i32 = C.c_int
i64 = C.c_int64
f32 = C.c_float
sz = C.c_size_t
stream_t = C.c_void_p


class ConvParams(C.Structure):
    _fields_ = [
        ("src0", f32p), ("src1", f32p),
        ("c0", i32), ("c1", i32), ("ld0", i32), ("ld1", i32),
        ("n_img", i32), ("hi", i32), ("wi", i32),
        ("hq", i32), ("wq", i32),
        ("stride", i32), ("upsample", i32), ("pad_mode", i32),
        ("kh", i32), ("kw", i32), ("pad_y", i32), ("pad_x", i32),
        ("weight", f32p),
        ("cout", i32), ("coutp", i32),
        ("bias", f32p),
        ("out", f32p),
        ("ldo", i32), ("ho", i32), ("wo", i32),
        ("out_scale", i32), ("out_off_y", i32), ("out_off_x", i32),
        ("residual", f32p),
        ("ldr", i32),
        ("act", i32),
        ("ksplit", i32),
        ("partial", f32p),
        ("gn_partial", f32p),
        ("gn_groups", i32), ("gn_pixels", i32),
        ("ln_wsum", f32p), ("ln_eps", f32),
        ("tile_counters", C.c_void_p), ("tile_counters_len", i32), ("weight_wino", C.c_void_p),
        ("deconv4", i32), ("groups", i32),
        ("pool2", i32),
        ("weight_wino4", C.c_void_p),
        ("defer_reduce", i32),
        ("weight_pw", f32p),
        ("gn_in_partial", f32p),
        ("gn_in_nchunk", i32), ("gn_in_groups", i32), ("gn_in_pixels", i32),
        ("gn_in_gamma", f32p), ("gn_in_beta", f32p), ("gn_in_ss", f32p),
        ("gn_in_ss_ld", i32), ("gn_in_eps", f32),
        ("res_gn_partial", f32p),
        ("res_gn_nchunk", i32), ("res_gn_groups", i32), ("res_gn_pixels", i32),
        ("res_gn_gamma", f32p), ("res_gn_beta", f32p), ("res_gn_eps", f32),
    ]


class WarpParams(C.Structure):
    _fields_ = [
        ("src", f32p), ("prev", f32p), ("out", f32p),
        ("batch", i32), ("frames", i32), ("h", i32), ("w", i32), ("c", i32),
        ("ld_src", i32), ("ld_prev", i32), ("ld_out", i32),
        ("flow_x", f32p), ("flow_y", f32p), ("occ", f32p),
        ("fh", i32), ("fw", i32),
        ("fsb", i64), ("fst", i64),
        ("occ_scale", f32), ("occ_bias", f32),
        ("prev_is_cl", i32),
    ]


class WgradParams(C.Structure):
    _fields_ = [
        ("x", f32p), ("cin", i32), ("ldx", i32),
        ("n_img", i32), ("hi", i32), ("wi", i32), ("hq", i32), ("wq", i32),
        ("stride", i32), ("kh", i32), ("kw", i32), ("pad_y", i32), ("pad_x", i32),
        ("dy", f32p), ("cout", i32), ("lddy", i32),
        ("dw", f32p), ("dw_layout", i32), ("dw_cin_total", i32), ("dw_ci_off", i32), ("dbias", f32p),
    ]


ML_MAX = 32


class MultiLinearParams(C.Structure):
    _fields_ = [
        ("n_blocks", i32), ("rows", i32), ("k", i32), ("act", i32),
        ("x", f32p),
        ("w", f32p * ML_MAX), ("bias", f32p * ML_MAX), ("n", i32 * ML_MAX), ("y", f32p * ML_MAX),
        ("dy", f32p * ML_MAX), ("dw", f32p * ML_MAX), ("dbias", f32p * ML_MAX),
        ("dx", f32p),
    ]


class BlurParams(C.Structure):
    _fields_ = [
        ("x", f32p), ("wgt", f32p), ("out", f32p), ("dy", f32p), ("dx", f32p),
        ("xs_n", i64), ("xs_c", i64), ("xs_h", i64), ("xs_w", i64), ("os_n", i64), ("os_c", i64), ("os_h", i64), ("os_w", i64),
        ("n_img", i32), ("channels", i32), ("c_store", i32), ("h", i32), ("w", i32), ("k", i32), ("pad_lo", i32), ("pad_hi", i32),
        ("stride", i32),
        ("scale", f32p), ("bias", f32p),
    ]


class WarpBwdParams(C.Structure):
    _fields_ = [
        ("src", f32p), ("dout", f32p), ("prev", f32p), ("dprev", f32p), ("dsrc_fix", C.c_void_p), ("amax_bits", C.c_void_p),
        ("dmaps", f32p), ("flow_x", f32p), ("flow_y", f32p), ("occ", f32p),
        ("n_img", i32), ("h", i32), ("w", i32), ("c", i32), ("fh", i32), ("fw", i32), ("n_div", i32), ("layout_cl", i32),
        ("fsn", i64),
        ("ld_src", i32), ("ld_dout", i32), ("ld_prev", i32), ("ld_dprev", i32),
        ("ss_n", i64), ("ss_c", i64), ("ss_h", i64), ("ss_w", i64), ("ds_n", i64), ("ds_c", i64), ("ds_h", i64), ("ds_w", i64),
        ("ps_n", i64), ("ps_c", i64), ("ps_h", i64), ("ps_w", i64), ("dps_n", i64), ("dps_c", i64), ("dps_h", i64), ("dps_w", i64),
    ]


class GridSampleParams(C.Structure):
    _fields_ = [
        ("x", f32p), ("grid", f32p), ("out", f32p), ("dout", f32p), ("dgrid", f32p),
        ("xs_n", i64), ("xs_c", i64), ("xs_h", i64), ("xs_w", i64), ("os_n", i64), ("os_c", i64), ("os_h", i64), ("os_w", i64),
        ("n_img", i32), ("channels", i32), ("h", i32), ("w", i32), ("ho", i32), ("wo", i32), ("n_div", i32), ("pad_mode", i32),
    ]


BN_TICKETS = 1024

_SIGNATURES = {
    # name: (restype, argtypes)
    "lfdm_last_error": (C.c_char_p, []),
    "lfdm_abi_version": (i32, []),
    "lfdm_calib_mfma_f32": (i32, [f32p, i32, i32, stream_t]),
    "lfdm_conv2d_cl_f32": (i32, [C.POINTER(ConvParams), stream_t]),
    "lfdm_conv2d_partial_bytes": (sz, [C.POINTER(ConvParams)]),
    "lfdm_conv2d_plan": (i32, [C.POINTER(ConvParams), C.POINTER(i32), C.POINTER(i32)]),
    "lfdm_conv2d_plan_slabs": (i32, [C.POINTER(ConvParams)]),
    "lfdm_conv2d_schedule": (i32, [C.POINTER(ConvParams)]),
    "lfdm_groupnorm_ws_bytes": (sz, [i32, i32, i32]),
    "lfdm_groupnorm_silu_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, i32, f32p,
                                        f32, i32, C.c_void_p, sz, stream_t]),
    "lfdm_groupnorm_apply_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, i32, f32p, f32, i32,
                                         f32p, i32, C.c_void_p, sz, stream_t]),
    "lfdm_layernorm_cl_f32": (i32, [f32p, f32p, i64, i32, f32p, f32, stream_t]),
    "lfdm_attention_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, stream_t]),
    "lfdm_temporal_attention_fused_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, i32, i32, i32, f32p, f32p, f32p, f32,
                                                  stream_t]),
    "lfdm_temporal_attention_fused_out_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, f32,
                                                      stream_t]),
    "lfdm_linear_attention_lowres_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, f32p, i32, i32, f32, stream_t]),
    "lfdm_attention_lowres_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, f32, stream_t]),
    "lfdm_pack_wino_weight_f32": (i32, [f32p, i32, i32, i32, i32, i32, f32p, stream_t]),
    "lfdm_pack_wino4_weight_f32": (i32, [f32p, i32, i32, i32, i32, f32p, stream_t]),
    "lfdm_pack_conv_weight_f32": (i32, [f32p, i32, i32, i32, i64, i64, i32, f32p, stream_t]),
    "lfdm_lfae_motion_inputs_f32": (i32, [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_float, i32, i32, i32, i32, i32, i32,
                                          f32p, i32, f32p, stream_t]),
    "lfdm_lfae_region_stats_f32": (i32, [f32p, i32, i32, i32, i32, i32, C.c_float, f32p, f32p, f32p, f32p, f32p, f32p, stream_t]),
    "lfdm_svd2x2_sym_f32": (i32, [f32p, i64, f32p, f32p, stream_t]),
    "lfdm_lfae_motion_combine_f32": (i32, [f32p, i32, f32p, i32, i32, i32, f32p, f32p, stream_t]),
    "lfdm_conv2d_smalln_cl_f32": (i32, [f32p, i32, i32, i32, i32, i32, f32p, f32p, f32p, i32, i32, i32, i32, stream_t]),
    "lfdm_linear_attention_fused_ws_bytes": (sz, [i32, i32]),
    "lfdm_linear_attention_fused_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, i32, i32, f32, C.c_void_p, sz, stream_t]),
    "lfdm_linear_attention_fused_out_cl_f32": (i32, [f32p, i32, i32, f32p, f32p, f32p, f32p, i32, i32, i32, f32, C.c_void_p, sz, stream_t]),
    "lfdm_linear_attention_ws_bytes": (sz, [i32]),
    "lfdm_linear_attention_cl_f32": (i32, [f32p, f32p, i32, i32, C.c_void_p, sz, stream_t]),
    "lfdm_linear_small_f32": (i32, [f32p, f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_step_cond_f32": (i32, [f32p, f32p, C.c_void_p, f32p, i32, i32, stream_t]),
    "lfdm_sinusoidal_f32": (i32, [C.c_void_p, i32, f32p, f32p, i32, i32, i32, stream_t]),
    "lfdm_conv_planar_in_cl_f32": (i32, [f32p, i32, i32, i32, i32, i32, i32, f32p, i32, i32, i32,
                                        f32p, f32p, f32p, i32, i32, stream_t]),
    "lfdm_heads_cl_to_planar_f32": (i32, [f32p, f32p, i32, i32, f32p, f32p, f32p, f32p, f32p, i32, i32,
                                         i32, stream_t]),
    "lfdm_heads_res_cl_to_planar_f32": (i32, [f32p, f32p, i32, i32, f32p, f32p, f32p, f32p, f32p, i32, i32, f32p, i32, i32, f32p, f32p, i32, i32,
                                              i32, stream_t]),
    "lfdm_heads_gn_res_cl_to_planar_f32": (i32, [f32p, i32, i32, f32p, i32, i32, f32p, f32p, f32, f32p, f32p, f32p, f32p, f32p, i32, i32, f32p, i32, i32,
                                                 f32p, f32p, i32, i32, i32, stream_t]),
    "lfdm_sampler_ws_bytes": (sz, [i32, i64]),
    "lfdm_sampler_ws_init": (i32, [C.c_void_p, sz, i32, i64, stream_t]),
    "lfdm_sampler_step_f32": (i32, [f32p, f32p, f32p, f32p, i32, i64, f32p, C.c_void_p, f32, i32,
                                   C.c_void_p, sz, stream_t]),
    "lfdm_cfg_combine_f32": (i32, [f32p, f32p, f32, f32p, i64, stream_t]),
    "lfdm_abs_quantile_f32": (i32, [f32p, i32, i64, f32, f32p, C.c_void_p, sz, stream_t]),
    "lfdm_warp_cl_f32": (i32, [C.POINTER(WarpParams), stream_t]),
    "lfdm_warp_planar_f32": (i32, [C.POINTER(WarpParams), stream_t]),
    "lfdm_affine_act_cl_f32": (i32, [f32p, f32p, i64, i32, i32, i32, f32p, f32p, i32, stream_t]),
    "lfdm_avgpool2_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, stream_t]),
    "lfdm_planar_to_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, stream_t]),
    "lfdm_cl_to_planar_f32": (i32, [f32p, f32p, i32, i32, i32, i32, stream_t]),
    # ---- training (backward) kernels
    "lfdm_multi_linear_f32": (i32, [C.POINTER(MultiLinearParams), stream_t]),
    "lfdm_multi_linear_bwd_ws_bytes": (sz, [C.POINTER(MultiLinearParams)]),
    "lfdm_multi_linear_bwd_f32": (i32, [C.POINTER(MultiLinearParams), C.c_void_p, sz, stream_t]),
    "lfdm_conv2d_wgrad_ws_bytes": (sz, [C.POINTER(WgradParams)]),
    "lfdm_conv2d_wgrad_cl_f32": (i32, [C.POINTER(WgradParams), C.c_void_p, sz, stream_t]),
    "lfdm_sum_leading_f32": (i32, [f32p, f32p, i64, i32, stream_t]),
    "lfdm_colsum_ws_bytes": (sz, [i64, i32]),
    "lfdm_colsum_f32": (i32, [f32p, i64, i32, i32, f32p, C.c_void_p, sz, stream_t]),
    "lfdm_groupnorm_bwd_ws_bytes": (sz, [i32, i32, i32]),
    "lfdm_groupnorm_silu_bwd_cl_f32": (i32, [f32p, f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, i32, f32, i32,
                                            f32p, i32, f32p, f32p, i32, C.c_void_p, sz, stream_t]),
    "lfdm_attention_bwd_ws_bytes": (sz, [i32, i32, i32, i32]),
    "lfdm_attention_bwd_cl_f32": (i32, [f32p, f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, f32p, C.c_void_p, sz,
                                       stream_t]),
    "lfdm_linear_attention_bwd_ws_bytes": (sz, [i32]),
    "lfdm_linear_attention_bwd_cl_f32": (i32, [f32p, f32p, f32p, i32, i32, C.c_void_p, sz, stream_t]),
    "lfdm_adam_step_f32": (i32, [f32p, f32p, f32p, f32p, i64, f32, f32, f32, f32, f32, i32, f32, stream_t]),
    "lfdm_depthwise_down_planar_f32": (i32, [f32p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_upsample2_pad_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_layernorm_bwd_ws_bytes": (sz, [i64, i32]),
    "lfdm_layernorm_bwd_cl_f32": (i32, [f32p, f32p, f32p, i64, i32, f32p, f32, f32p, C.c_void_p, sz, stream_t]),
    "lfdm_layernorm_bwd_add_cl_f32": (i32, [f32p, f32p, f32p, f32p, i64, i32, f32p, f32, f32p, C.c_void_p, sz, stream_t]),
    # ---- LFAE stage-1 training glue (ABI 10)
    "lfdm_batchnorm_train_ws_bytes": (sz, [i64, i32, i32]),
    "lfdm_batchnorm_train_fwd_cl_f32": (i32, [f32p, f32p, i64, i32, i32, i32, i32, f32p, f32p, f32p, f32p, f32, f32, i32, f32p, C.c_void_p, sz,
                                            C.c_void_p, stream_t]),
    "lfdm_batchnorm_train_bwd_cl_f32": (i32, [f32p, f32p, f32p, i64, i32, i32, i32, i32, i32, f32p, i32, f32p, f32p, f32p, i32, f32p, f32p, C.c_void_p, sz,
                                            C.c_void_p, stream_t]),
    "lfdm_blur_down_fwd_f32": (i32, [C.POINTER(BlurParams), stream_t]),
    "lfdm_blur_down_bwd_f32": (i32, [C.POINTER(BlurParams), stream_t]),
    "lfdm_warp_bwd_f32": (i32, [C.POINTER(WarpBwdParams), stream_t]),
    "lfdm_absmax_f32": (i32, [f32p, i64, i32, i64, C.c_void_p, stream_t]),
    "lfdm_fix_finalize_f32": (i32, [C.c_void_p, f32p, i64, i32, i64, C.c_void_p, i64, C.c_void_p, stream_t]),
    "lfdm_resize_adjoint_f32": (i32, [f32p, f32p, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_grid_sample_fwd_f32": (i32, [C.POINTER(GridSampleParams), stream_t]),
    "lfdm_grid_sample_bwd_f32": (i32, [C.POINTER(GridSampleParams), stream_t]),
    "lfdm_svd2x2_sym_bwd_f32": (i32, [f32p, f32p, f32p, f32p, f32p, i64, stream_t]),
    "lfdm_pack_wino_weights_multi_f32": (i32, [C.c_void_p, i32, i32, stream_t]),
    "lfdm_im2col_cl_f32": (i32, [f32p, f32p, i32, i32, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_pool2_cl_f32": (i32, [f32p, f32p, f32p, i32, i32, i32, i32, i32, stream_t]),
    "lfdm_relu_bwd_f32": (i32, [f32p, f32p, f32p, i64, stream_t]),
    "lfdm_l1_mean_fwd_f32": (i32, [f32p, f32p, i64, f32, f32p, C.c_void_p, sz, C.c_void_p, stream_t]),
    "lfdm_l1_mean_bwd_f32": (i32, [f32p, f32p, i64, f32, f32p, f32p, stream_t]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class NativeLibrary:
    def __init__(self, path, kind):
        if not os.path.exists(path):
            raise RuntimeError(
                "native library %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'`"
                " (hipcc --offload-arch=gfx950); there is no fallback path" % path)
        self.path = path
        self.kind = kind  # "hip" (product) or "emu" (tests only)
        self.dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, rc, what):
        if rc != 0:
            msg = self.lfdm_last_error()
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


_active = None


def library():
    """The active native library; loads liblfdm_hip.so on first use, raises if it is absent."""
    global _active
    if _active is None:
        _active = NativeLibrary(HIP_LIB_PATH, "hip")
    return _active


def _set_library_for_tests(lib):
    global _active
    _active = lib
