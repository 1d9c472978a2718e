"""ctypes binding of libmagicdec_hip.so (the C ABI declared in include/magicdec_hip.h).

There is exactly one backend: the HIP library.  If it is missing or fails to
load, every op raises -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAGICDEC_HIP_LIB: development only -- another build of the same sources (e.g. `make TIMING=1`'s instrumented library)
LIB_PATH = os.environ.get("MAGICDEC_HIP_LIB") or os.path.join(_HERE, "libmagicdec_hip.so")
ABI_VERSION = 10

_lib = None
_err = None

P = c_void_p
I = c_int
L = c_int64

class FusedLinearArgs(ctypes.Structure):
    """md_fused_linear_args of include/magicdec_hip.h (field order and types must match)."""
    _fields_ = [("x", P), ("w_packed", P), ("bias", P), ("out", P), ("resid", P),
                ("ldx", L), ("ldo", L), ("ldr", L),
                ("M", I), ("N", I), ("K", I), ("epilogue", I),
                ("H", I), ("KH", I), ("D", I), ("rows_per_req", I), ("max_pos", I), ("page_size", I), ("kv_dtype", I),
                ("offsets", P), ("cos_sin", P),
                ("cache", P), ("page_indices", P), ("page_indptr", P), ("last_page_len", P),
                ("cache2", P), ("page_indices2", P), ("page_indptr2", P), ("last_page_len2", P),
                ("k_scale", P), ("v_scale", P),
                ("ssq_out", P), ("pro_ssq", P), ("pro_norm_w", P), ("pro_eps", c_float), ("pro_tiles", I)]


_SIGNATURES = {
    "md_abi_version": (c_int, []),
    "md_last_error_string": (c_char_p, []),
    "md_clear_last_hip_error": (c_int, []),
    "md_page_overflow_count": (c_int, [P, I]),
    "md_append_paged_kv": (c_int, [P, P, L, L, P, P, P, P, P, I, I, I, I, I, I, P, P, P]),
    "md_rope": (c_int, [P, P, L, L, P, P, P, P, I, I, I, I, I, P, I, P]),
    "md_rope_fill_table_host": (c_int, [P, I, I, c_double, c_double, c_double, c_double, c_double]),
    "md_rope_append": (c_int, [P, P, P, L, L, L, P, P, P, I, I, I, I, I, P, I, P, P, P, P, P, P, P, P, I, I, P, P, P]),
    "md_paged_attn_workspace_bytes": (c_size_t, [I, I, I, I, I, I, I]),
    "md_debug_attn_timing": (None, [I, I]),
    "md_debug_attn_timing_read": (c_int, [P, I]),
    "md_paged_attn": (c_int, [P, L, P, P, P, P, P, P, I, I, I, I, I, I, I, c_float, I, I, P, P, P, c_size_t, P]),
    "md_snapkv_workspace_bytes": (c_size_t, [I, I, I, I, I]),
    "md_snapkv_scores_offset": (c_size_t, [I, I, I, I, I]),
    "md_snapkv_select": (c_int, [P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P, P, I, P, P, P, c_size_t, P]),
    "md_ar_create": (c_int, [I, I, c_size_t, P]),
    "md_ar_get_handles": (c_int, [P, P]),
    "md_ar_open_peers": (c_int, [P, P]),
    "md_allreduce_oneshot": (c_int, [P, P, P, c_size_t, P]),
    "md_allreduce": (c_int, [P, P, P, c_size_t, I, P]),
    "md_allreduce_add_rmsnorm": (c_int, [P, P, P, P, P, P, I, I, c_float, I, P]),
    "md_ar_set_publish": (c_int, [P, I]),
    "md_ar_status": (c_int, [P, P]),
    "md_ar_status_async": (c_int, [P, P, P]),
    "md_ar_destroy": (c_int, [P]),
    "md_streaming_shift_append": (c_int, [P, P, L, L, P, I, I, I, I, I, I, I, I, P]),
    "md_streaming_rotate": (c_int, [P, P, I, I, I, I, I, I, P, I, P]),
    "md_linear_supported": (c_int, [I, I, I, I]),
    "md_linear_workspace_bytes": (c_size_t, [I, I, I, I]),
    "md_linear": (c_int, [P, L, P, I, I, P, P, P, L, I, I, I, I, P, c_size_t, P]),
    "md_linear_normed": (c_int, [P, L, P, I, P, c_float, P, I, P, P, L, I, I, I, I, P, c_size_t, P]),
    "md_linear_add_rmsnorm_supported": (c_int, [I, I, I]),
    "md_linear_add_rmsnorm": (c_int, [P, L, P, I, I, P, P, P, L, P, c_float, P, P, I, I, I, P, c_size_t, P]),
    "md_linear_block_supported": (c_int, [I, I, I, I]),
    "md_linear_block_workspace_bytes": (c_size_t, [I, I, I, I]),
    "md_linear_block": (c_int, [P, L, P, P, P, L, I, I, I, I, P, c_size_t, P]),
    "md_linear_block_add_rmsnorm": (c_int, [P, L, P, P, P, L, P, c_float, P, P, I, I, I, P, c_size_t, P]),
    "md_linear_fused_supported": (c_int, [I, I, I, I]),
    "md_linear_fused": (c_int, [ctypes.POINTER(FusedLinearArgs), P]),
    "md_linear_fused_split_workspace_bytes": (c_size_t, [I, I, I]),
    "md_linear_fused_split": (c_int, [P, L, P, P, P, L, I, I, I, P, c_size_t, P]),
    "md_linear_fused_split_add_rmsnorm": (c_int, [P, L, P, P, P, L, P, c_float, P, P, I, I, I, P, c_size_t, P]),
    "md_rmsnorm": (c_int, [P, P, P, I, I, c_float, P]),
    "md_add_rmsnorm": (c_int, [P, P, P, P, P, I, I, c_float, P]),
    "md_silu_mul": (c_int, [P, P, L, L, P, I, I, P]),
    "md_argmax": (c_int, [P, L, I, I, L, P, P, P]),
    "md_argmax_tp_slots": (c_int, [P, L, I, I, L, I, I, P, P, P]),
    "md_tp_argmax_merge": (c_int, [P, P, I, I, P, P]),
    "md_accept_rollback": (c_int, [P, P, P, I, P, P, P, P, P, I, I, I, I, L, L, L, P, P, P, P, P, P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# include/magicdec_hip_dev.h: tuning knobs, present only in a -DMD_DEV_KNOBS build (bound when the library has them;
# nothing under magicdec_amd/ needs them)
_DEV_SIGNATURES = {
    "md_debug_set_attn_target_wgs": (None, [I]),
    "md_debug_set_attn_waves": (None, [I]),
    "md_debug_set_prefill_kt": (None, [I, I]),
    "md_debug_set_prefill_mfma32": (None, [I]),
    "md_debug_set_gemm_target_blocks": (None, [I]),
    "md_debug_set_gemm_waves": (None, [I]),
    "md_debug_set_gemm_ring": (None, [I]),
    "md_debug_set_fused_nw": (None, [I]),
    "md_debug_set_fused_split": (None, [I]),
    "md_debug_set_tile_timing": (None, [P]),
    "md_debug_set_block_gemm": (None, [I, I]),
}
DEV_SYMBOLS = tuple(_DEV_SIGNATURES)


class MagicDecHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise MagicDecHipError if that is impossible."""
    global _lib, _err
    if _lib is not None:
        return _lib
    if _err is not None:
        raise MagicDecHipError(_err)
    if not os.path.exists(LIB_PATH):
        _err = (f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C magicdec_amd/csrc`.  magicdec_amd has no CPU fallback.")
        raise MagicDecHipError(_err)
    try:
        import torch  # noqa: F401  (loads the HIP runtime torch was built with first)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _DEV_SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        if lib.md_abi_version() != ABI_VERSION:
            raise OSError(f"ABI version {lib.md_abi_version()} != expected {ABI_VERSION}")
    except (OSError, AttributeError) as e:  # missing symbol / loader failure
        _err = f"cannot load {LIB_PATH}: {e}"
        raise MagicDecHipError(_err) from e
    v = os.environ.get("MAGICDEC_PREFILL_MFMA32")          # development A/B switch of the prefill attention kernel
    if v is not None and hasattr(lib, "md_debug_set_prefill_mfma32"):
        lib.md_debug_set_prefill_mfma32(int(v))
    _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().md_last_error_string()
        raise MagicDecHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
