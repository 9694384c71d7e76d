"""ctypes binding of the gfx950 C-ABI library (include/lkgd_hip.h).

The product path has NO fallback: if ``liblkgd_hip.so`` is missing or fails to load, every op raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C lkgd_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: LKGD_HIP_LIB points A/B measurements at another build of the same ABI (tools/); the product path is the in-tree build
LIB_PATH = os.environ.get("LKGD_HIP_LIB") or os.path.join(_HERE, "liblkgd_hip.so")

ERRORS = {-1: "LKGD_E_NULL", -2: "LKGD_E_SHAPE", -3: "LKGD_E_ALIGN", -4: "LKGD_E_MODE", -5: "LKGD_E_LAUNCH"}


class LkgdHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """struct lkgd_gemm_desc (include/lkgd_hip.h)."""
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("res1", C.c_void_p), ("res2", C.c_void_p), ("out", C.c_void_p), ("zeros", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda0", C.c_int32), ("lda1", C.c_int32), ("csplit", C.c_int32),
        ("mode", C.c_int32), ("Cin", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("stride", C.c_int32), ("ups", C.c_int32),
        ("F", C.c_int32), ("HW", C.c_int32), ("Floc", C.c_int32), ("f_off", C.c_int32),
        ("ldrb", C.c_int32), ("rb_d1", C.c_int32), ("rb_m1", C.c_int32), ("rb_d2", C.c_int32), ("rb_md", C.c_int32),
        ("rb_c0", C.c_int32),
        ("ldr1", C.c_int32), ("ldr2", C.c_int32), ("ldc", C.c_int32),
        ("s_acc", C.c_float), ("r1", C.c_float), ("r2", C.c_float),
        ("geglu", C.c_int32), ("pad_off", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("colstats", C.c_void_p),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("cs_rows", C.c_int32),
    ]


#: every symbol include/lkgd_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
class FsmDesc(C.Structure):
    """struct lkgd_fsm_desc (include/lkgd_hip.h section 9)."""
    _fields_ = [
        ("a", C.c_void_p), ("res", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("csr_off", C.c_void_p), ("csr_pt", C.c_void_p), ("gather_idx", C.c_void_p), ("vis", C.c_void_p),
        ("a_pair_rows", C.c_int64), ("a_off", C.c_int64), ("r_pair_rows", C.c_int64), ("r_off", C.c_int64),
        ("o_pair_rows", C.c_int64), ("o_off", C.c_int64),
        ("lda", C.c_int32), ("ldr", C.c_int32), ("ldb", C.c_int32), ("ldo", C.c_int32),
        ("bias_mul", C.c_int32), ("bias_add", C.c_int32), ("bias_div", C.c_int32),
        ("pairs", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("P", C.c_int32),
    ]


SYMBOLS = {
    "lkgd_gemm_f16": (_i32, [C.POINTER(GemmDesc), _vp]),
    "lkgd_gemm_colstats_block": (_i32, [C.POINTER(GemmDesc)]),
    "lkgd_gemm_wide_tile_n": (_i32, [_i32]),
    "lkgd_groupnorm_stats_cols": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _i64, _f32, _i32, _vp, _vp]),
    "lkgd_groupnorm_chunks": (_i32, [_i64, _i32]),
    "lkgd_groupnorm_stats": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _i64, _f32, _vp, _vp, _vp]),
    "lkgd_groupnorm_sums": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _vp]),
    "lkgd_groupnorm_finalize": (_i32, [_vp, _i64, C.c_double, _f32, _vp, _vp]),
    "lkgd_groupnorm_finalize_parts": (_i32, [_vp, _i32, _i64, _i64, _i64, C.c_double, _f32, _vp, _vp]),
    "lkgd_groupnorm_apply_segments": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "lkgd_groupnorm_silu": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    "lkgd_groupnorm_apply": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _vp, _i32,
                                    _vp]),
    "lkgd_layernorm": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32,
                              _vp]),
    "lkgd_tattn_block_c320": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "lkgd_ln_qkv_c320": (_i32, [_vp, _i32, _i64, _vp, _f32, _vp, _i32, _vp]),
    "lkgd_ln_qkv_c640": (_i32, [_vp, _i32, _i64, _vp, _f32, _vp, _i32, _vp]),
    "lkgd_ff_fused_c320": (_i32, [_vp, _i32, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _i32, _f32, _vp, _i32, _vp]),
    "lkgd_attn_spatial": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp]),
    "lkgd_attn_spatial_qk": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _vp]),
    "lkgd_attn_temporal": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _f32,
                                  _vp]),
    "lkgd_tattn_front": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    "lkgd_prepare_unet_input": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "lkgd_cfg_euler_step": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "lkgd_shard_rows": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "lkgd_tokens_to_nchw": (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "lkgd_nchw_to_tokens": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "lkgd_timestep_embedding": (_i32, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "lkgd_silu": (_i32, [_vp, _vp, _i64, _vp]),
    "lkgd_add": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "lkgd_scale": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "lkgd_euler_step": (_i32, [_vp, _vp, _i32, _vp, _i64, _f32, _f32, _i32, _vp]),
    "lkgd_attn_cross": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                _f32, _vp]),
    "lkgd_attn_dense": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "lkgd_euler_step_churn": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "lkgd_fsm_rows": (_i32, [C.POINTER(FsmDesc), _vp]),
    "lkgd_conv3x3_small": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    "lkgd_conv1d_reflect": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp]),
    "lkgd_resize_bicubic_ac": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _vp]),
    "lkgd_softmax_rows": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "lkgd_time_conv_out": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "lkgd_vit_patchify": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "lkgd_gelu_tanh": (_i32, [_vp, _vp, _i64, _vp]),
    "lkgd_gated_add": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _i32, _vp]),
    "lkgd_lk_fuse": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "lkgd_version": (C.c_char_p, []),
    # debug / measurement knobs (process-global, not thread-safe: include/lkgd_hip.h, last section)
}

#: include/lkgd_hip_debug.h: A/B and test knobs, per host thread; bound on the same library object, not part of the product interface
DEBUG_SYMBOLS = {
    "lkgd_debug_set_wide_tile_n": (None, [_i32]),
    "lkgd_debug_set_wide_tile_m": (None, [_i32]),
    "lkgd_debug_set_gemm_variant": (None, [_i32]),
    "lkgd_debug_set_gemm_splitk": (None, [_i32]),
    "lkgd_debug_set_mid_model": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _i32]),
    "lkgd_debug_set_wide_ksplit": (None, [_i32]),
    "lkgd_debug_set_wide_lds_out": (None, [_i32]),
    "lkgd_debug_set_attn_waves": (None, [_i32]),
    "lkgd_debug_set_attn_kvb": (None, [_i32]),
    "lkgd_debug_set_attn_pipe": (None, [_i32]),
    "lkgd_debug_set_gn_apply_kb": (None, [_i32]),
    "lkgd_debug_set_gn_stats_kb": (None, [_i32]),
    "lkgd_debug_set_gn_target_wgs": (None, [_i32]),
    "lkgd_debug_set_gn_small": (None, [_i32]),
    "lkgd_debug_set_gn_small_limits": (None, [_i64]),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises LkgdHipError when it is absent - there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LkgdHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C lkgd_amd/csrc` "
                "(or __graft_entry__.build()). lkgd_amd has no CPU fallback.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise LkgdHipError(f"cannot load {LIB_PATH}: {e}") from e
        for table in (SYMBOLS, DEBUG_SYMBOLS):
            for name, (res, args) in table.items():
                fn = getattr(l, name)       # AttributeError here = header / library mismatch
                fn.restype, fn.argtypes = res, args
        _lib = l
    if _ENV_KNOBS and not getattr(_knob_tls, "done", False):
        # A/B measurements only (same-box bench pairs).  The knobs are per host thread in the library, so an environment knob is
        # applied once on every thread that reaches the library
        _knob_tls.done = True
        for fn, v in _ENV_KNOBS:
            getattr(_lib, fn)(v)
    return _lib


#: LKGD_ATTN_PIPE: 1 = never the software-pipelined attention program, 2 = wherever legal; LKGD_GN_TARGET_WGS, LKGD_GEMM_VARIANT,
#: LKGD_GN_SMALL: see include/lkgd_hip_debug.h
_ENV_KNOBS = [(fn, int(os.environ[env])) for env, fn in (
    ("LKGD_ATTN_PIPE", "lkgd_debug_set_attn_pipe"), ("LKGD_GN_TARGET_WGS", "lkgd_debug_set_gn_target_wgs"),
    ("LKGD_GEMM_VARIANT", "lkgd_debug_set_gemm_variant"), ("LKGD_GN_SMALL", "lkgd_debug_set_gn_small")) if os.environ.get(env)]
_knob_tls = __import__("threading").local()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise LkgdHipError(f"{what} failed: {ERRORS.get(rc, rc)}")
