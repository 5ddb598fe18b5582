"""ctypes binding of the C-ABI declared in include/mi355x_hotpath.h.

This module only loads the library and declares prototypes; it performs no
compute at import.  The product path fails loudly (ImportError / RuntimeError) when
the HIP library has not been built — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libmi355x_hotpath.so"

F16, BF16, F32 = 0, 1, 2
KV_AUTO, KV_FP8_E4M3, KV_FP8_E5M2 = 0, 1, 2          # mi355x_kv_cache_dtype
ABI_VERSION = 5                      # the MI355X_ABI_VERSION the PROTOTYPES below were written for

_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float

# name -> (restype, argtypes); kept in the same order as the header.
PROTOTYPES = {
    "mi355x_abi_version": (_I, []),
    "mi355x_last_error": (c_char_p, []),
    "mi355x_get_device_attribute": (_L, [_L, _L]),
    "mi355x_get_max_shared_memory_per_block_device_attribute": (_L, [_L]),
    "mi355x_reshape_and_cache": (
        _I, [_P, _P, _P, _P, _P, _I, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mi355x_reshape_and_cache_flash": (
        _I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mi355x_convert_fp8": (_I, [_P, _P, _L, _F, _I, _I, _P]),
    "mi355x_copy_blocks": (_I, [_P, _P, _I, _P, _I, _L, _P]),
    "mi355x_swap_blocks": (_I, [_P, _P, _P, _I, _L, _I, _P]),
    "mi355x_paged_attention_v1": (
        _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P, _L, _L, _L, _I, _I, _P, _P,
             _P]),
    "mi355x_paged_attention_v1_max_seq_len": (_I, [_I, _I, _I, _I, _I, _I]),
    "mi355x_paged_attention_v2": (
        _I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P, _L, _L, _L,
             _I, _I, _P, _P, _P]),
    "mi355x_paged_attention_v2_ps": (
        _I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _P, _L, _L, _L,
             _I, _I, _P, _P, _I, _P]),
    "mi355x_paged_prefill_attention": (
        _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _I, _L, _L, _L, _L, _I, _I, _P,
             _P, _I, _F, _P]),
    "mi355x_paged_prefill_attention_alibi": (
        _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _I, _L, _L, _L, _L, _I, _I, _P,
             _P, _I, _F, _P, _P]),
    "mi355x_rms_norm": (_I, [_P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_fused_add_rms_norm": (_I, [_P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_fused_add_rms_norm_slabs": (_I, [_P, _P, _P, _P, _I, _F, _I, _I, _L, _I, _P]),
    "mi355x_rms_norm_static_fp8_quant": (_I, [_P, _P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_fused_add_rms_norm_static_fp8_quant": (
        _I, [_P, _P, _P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_rms_norm_dynamic_per_token_quant": (
        _I, [_P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
    "mi355x_static_scaled_fp8_quant": (_I, [_P, _P, _P, _I, _I, _L, _L, _I, _P]),
    "mi355x_dynamic_scaled_fp8_quant": (_I, [_P, _P, _P, _I, _I, _L, _L, _I, _P]),
    "mi355x_dynamic_per_token_scaled_fp8_quant": (_I, [_P, _P, _P, _P, _I, _I, _L, _L, _I, _P]),
    "mi355x_rotary_embedding": (
        _I, [_P, _P, _P, _P, _I, _I, _L, _L, _L, _I, _I, _I, _I, _I, _P]),
    "mi355x_batched_rotary_embedding": (
        _I, [_P, _P, _P, _P, _P, _I, _I, _L, _L, _L, _I, _I, _I, _I, _I, _P]),
    "mi355x_silu_and_mul": (_I, [_P, _P, _I, _I, _I, _P]),
    "mi355x_silu_and_mul_quant": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mi355x_silu_and_mul_per_token_quant": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mi355x_awq_to_gptq_4bit": (_I, [_P, _P, _I, _I, _P]),
    "mi355x_awq_dequantize": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mi355x_awq_gemm": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _L, _I, _P]),
    "mi355x_awq_gemm_deferred": (
        _I, [_P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _L, _I, _P, _P]),
    "mi355x_awq_gemm_silu_mul": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _P]),
    "mi355x_awq_gemm_silu_mul_packed": (
        _I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _P]),
    "mi355x_awq_gemm_packed_a": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "mi355x_w4a16_prepack": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mi355x_w4a16_gemm_prepacked": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _L, _I, _I, _P]),
    "mi355x_w4a16_prepacked_split_elems": (_L, [_I, _I, _I]),
    "mi355x_gptq_shuffle": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mi355x_gptq_gemm": (
        _I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "mi355x_merge_attn_states": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mi355x_qkv_rope_cache": (
        _I, [_P, _L, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _I, _P]),
    "mi355x_rotary_reshape_and_cache": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _I, _I, _I, _I, _I, _P]),
    "mi355x_paged_prefill_attention_image": (
        _I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _I, _I, _L, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    "mi355x_rms_norm_image": (_I, [_P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_fused_add_rms_norm_image": (_I, [_P, _P, _P, _P, _F, _I, _I, _L, _I, _P]),
    "mi355x_paged_attention_fused_qkv": (
        _I, [_P, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _L,
             _L, _I, _I, _P]),
    "mi355x_greedy_advance": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mi355x_scaled_mm_int8": (
        _I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _I, _I, _I, _L, _L, _L, _I, _P]),
    "mi355x_static_scaled_int8_quant": (_I, [_P, _P, _P, _I, _I, _L, _I, _P]),
    "mi355x_dynamic_scaled_int8_quant": (_I, [_P, _P, _P, _I, _I, _L, _I, _P]),
    "mi355x_scaled_mm_fp8": (
        _I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _I, _I, _I, _L, _L, _L, _I, _P]),
    "mi355x_scaled_mm_prepack": (_I, [_P, _P, _I, _I, _L, _P]),
    "mi355x_scaled_mm_split_elems": (_L, [_I, _I, _I]),
    "mi355x_scaled_mm_prepacked": (
        _I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _I, _I, _I, _L, _L, _I, _I, _P]),
    "mi355x_scaled_mm_fp8_deferred": (
        _I, [_P, _P, _P, _P, _I, _P, _I, _P, _L, _I, _I, _I, _L, _L, _L, _I, _P, _P]),
    "mi355x_paged_attention_fused_qkv_w8": (
        _I, [_P, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _L,
             _L, _I, _I, _P, _I, _P, _I, _P, _P, _P]),
    "mi355x_silu_and_mul_per_token_quant_slabs": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "mi355x_rms_norm_dynamic_per_token_quant_slabs": (
        _I, [_P, _P, _I, _P, _I, _P, _I, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libmi355x_hotpath.so (once) and attach prototypes to every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get("MI355X_HOTPATH_LIB"):      # kernel experiments: an alternative build
        path = type(LIB_PATH)(os.environ["MI355X_HOTPATH_LIB"])
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build it with `python -m vllm_metax_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(str(path))
    # a library of another ABI version may still export every symbol, with parameters inserted in the middle of
    # the lists (version 2 did that): calling it would shift arguments into wild device pointers
    lib.mi355x_abi_version.restype = _I
    lib.mi355x_abi_version.argtypes = []
    got = lib.mi355x_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{path} has C-ABI version {got}, these bindings are written for version {ABI_VERSION} "
                          "(include/mi355x_hotpath.h): rebuild with `python -m vllm_metax_amd.build`")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().mi355x_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")
