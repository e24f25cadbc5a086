"""ctypes binding of liboctic_hip.so (the C ABI declared in include/octic_hip.h).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing the ops fails
loudly.  Nothing in this package routes through a CPU or composite-torch implementation of a hot op.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCTIC_LIB") or os.path.join(HERE, "liboctic_hip.so")   # OCTIC_LIB: developer A/B builds
HEADER_PATH = os.path.join(HERE, "..", "include", "octic_hip.h")

F32, BF16 = 0, 1
ABI_VERSION = 20
# knobs of octic_route_override (include/octic_hip.h)
(ROUTE_DENSE_TILE, ROUTE_DENSE_SPLIT, ROUTE_WGRAD_SLABS, ROUTE_WGRAD_TILE, ROUTE_LINEAR_RING, ROUTE_RING_EVEN,
 ROUTE_ATTN_LEGACY, ROUTE_ATTN_ONLINE, ROUTE_ATTN_BWD_PAIR, ROUTE_DENSE_IMAGE, ROUTE_DENSE_CLS2) = range(11)

c_i64, c_int, c_float, c_void_p = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class OcticView(ctypes.Structure):
    _fields_ = [("ptr", c_void_p * 5), ("ld", c_i64 * 5)]


PtrArray5 = c_void_p * 5
VP = ctypes.POINTER(OcticView)

# name -> (restype, argtypes)
_PROTOS = {
    "octic_abi_version": (c_int, []),
    "octic_strerror": (ctypes.c_char_p, [c_int]),
    "octic_route_override": (c_int, [c_int, c_int]),
    "octic_gelu_d8_fwd": (c_int, [VP, VP, c_i64, c_int, c_int, c_void_p]),
    "octic_gelu_d8_bwd": (c_int, [VP, VP, VP, c_i64, c_int, c_int, c_void_p]),
    "octic_layernorm_d8_fwd": (c_int, [VP, VP, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_int, c_void_p]),
    "octic_layernorm_d8_bwd_blocks": (c_int, [c_i64]),
    "octic_layernorm_d8_bwd": (c_int, [VP, VP, c_void_p, c_void_p, VP, VP, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "octic_layernorm_d8_bwd_cast": (c_int, [VP, VP, c_void_p, c_void_p, VP, VP, c_void_p, c_i64, c_int, c_void_p, c_i64,
                                            c_void_p, c_void_p]),
    "octic_layernorm_d8_bwd_finish": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "octic_layernorm_d8_bwd_finish_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "octic_sample_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "octic_linear_d8_fwd": (c_int, [VP, c_void_p, c_void_p, VP, VP, c_void_p, c_i64, c_void_p, c_i64, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "octic_linear_d8_tile_n": (c_int, [c_i64, c_int, c_int]),
    "octic_linear_d8_ring_order": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "octic_linear_d8_wgrad_tile": (c_int, [c_i64, c_int, c_int]),
    "octic_linear_d8_wgrad_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "octic_linear_d8_wgrad_splits": (c_int, [c_i64, c_int, c_int]),
    "octic_linear_d8_wgrad": (c_int, [VP, VP, c_i64, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "octic_linear_d8_wgrad_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "octic_linear_d8_wgrad_finish_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "octic_linear_d8_wgrad_has_colsum": (c_int, [c_int, c_int, c_int]),
    "octic_lamb_workspace_floats": (c_i64, [c_int, c_int]),
    "octic_lamb_step": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                                  c_int, c_float, c_void_p, c_void_p]),
    "octic_adamw_step": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_float,
                                                   c_int, c_float, c_void_p, c_void_p]),
    "octic_dense_blocks": (c_int, [c_i64]),
    "octic_dense_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                                          c_float, c_void_p]),
    "octic_dense_resid_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_int,
                                                c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_void_p]),
    "octic_dense_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_i64, c_int, c_void_p]),
    "octic_dense_layernorm_bwd_tail": (c_int, [c_void_p] * 10 + [c_i64, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "octic_dense_finish": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "octic_dense_finish_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "octic_dense_gelu_blocks": (c_int, []),
    "octic_dense_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "octic_dense_layernorm_fwd_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float,
                                               c_void_p, c_void_p, c_void_p]),
    "octic_dense_layernorm_bwd_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                               c_int, c_void_p, c_void_p]),
    "octic_scale_residual_fwd_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int,
                                              c_void_p, c_void_p]),
    "octic_scale_residual_bwd_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64,
                                              c_int, c_void_p, c_void_p]),
    "octic_softmax_center": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_float, c_void_p, c_i64, c_int, c_void_p]),
    "octic_soft_ce_fwd": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_i64, c_float, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                                  c_void_p]),
    "octic_soft_ce_bwd": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_i64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                                  c_i64, c_int, c_void_p]),
    "octic_scale_residual_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int,
                                         c_void_p]),
    "octic_scale_residual_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p,
                                         c_i64, c_int, c_void_p]),
    "octic_colsum_blocks": (c_int, [c_i64]),
    "octic_colsum_a1": (c_int, [VP, c_i64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "octic_cast_rowscale": (c_int, [VP, VP, c_void_p, c_i64, c_i64, c_int, c_int, c_void_p]),
    "octic_linear_d8_prep": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "octic_linear_d8_prep_batch_blocks": (c_int, [c_int, c_int]),
    "octic_linear_d8_prep_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "octic_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int,
                               c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_float, c_void_p]),
    "octic_attn_fwd_packed": (c_int, [c_void_p] * 3 + [c_i64, c_int, c_int, c_int, c_i64, c_i64, c_float, c_void_p]),
    "octic_attn_bwd_packed": (c_int, [c_void_p] * 6 + [c_i64, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_float, c_int, c_void_p]),
    "octic_attn_bwd": (c_int, [c_void_p] * 10 + [c_i64, c_int, c_int, c_int] + [c_i64] * 9 + [c_float, c_int, c_void_p]),
    "octic_attn_pack_heads": (c_int, [VP, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "octic_attn_unpack_heads": (c_int, [c_void_p, VP, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "octic_handoff_cat_fwd": (c_int, [VP, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "octic_handoff_cat_bwd": (c_int, [c_void_p, VP, c_i64, c_int, c_void_p]),
    "octic_power_spectrum_fwd": (c_int, [VP, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "octic_power_spectrum_bwd": (c_int, [c_void_p, VP, VP, c_i64, c_int, c_void_p]),
    "octic_im2col_patches": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "octic_lift_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int,
                                c_int, c_void_p]),
    "octic_lift_wgrad_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "octic_lift_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_int, c_void_p]),
    "octic_dense_prep_batch_blocks": (c_int, [c_int, c_int]),
    "octic_dense_prep_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "octic_dense_gemm_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "octic_dense_gemm_colsum_rows": (c_int, [c_int, c_int, c_int]),
    "octic_dense_gemm_tile": (c_int, [c_int, c_int, c_int, c_int]),
    "octic_dense_gemm_nt": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_int, c_void_p, c_void_p, c_i64,
                                    c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "octic_dense_gemm_nt_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_int, c_void_p, c_void_p,
                                           c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_void_p]),
    "octic_dense_gemm_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    "octic_dense_colsum": (c_int, [c_void_p, c_i64, c_int, c_i64, c_void_p, c_void_p]),
    "octic_dense_wgrad_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "octic_dense_wgrad_tile": (c_int, [c_int, c_int, c_int]),
    "octic_dense_wgrad_tn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p]),
    "octic_dense_wgrad_pair_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "octic_dense_wgrad_tn_pair": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64,
                                          c_void_p, c_int, c_int, c_void_p, c_void_p]),
}


def header_symbols():
    """Every function name declared in include/octic_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(octic_[a-z0-9_]+)\s*\(", text)))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"octic_vits_amd: HIP library {LIB_PATH} is missing. Build it with "
                "`python -m octic_vits_amd.build` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported: fail loudly
            fn.restype, fn.argtypes = res, args
        if L.octic_abi_version() != ABI_VERSION:
            raise RuntimeError("octic_vits_amd: ABI version mismatch between _lib.py and liboctic_hip.so")
        _LIB = L
    return _LIB


def route_override(knob: int, value: int) -> int:
    """Force a kernel / tiling choice for an A/B or a test (0 = automatic); returns the previous value."""
    old = lib().octic_route_override(knob, value)
    if old < 0:
        raise ValueError(f"octic_route_override: unknown knob {knob}")
    return old


def check(code: int):
    if code != 0:
        raise RuntimeError(f"octic HIP call failed ({code}): {lib().octic_strerror(code).decode()}")
