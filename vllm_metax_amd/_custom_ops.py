"""Python op surface of the MI355X hot path.

Mirrors, name for name and argument for argument, the ops the reference exposes
through ``torch.ops._C`` / ``torch.ops._C_cache_ops`` (schemas:
/root/reference csrc/torch_bindings.cpp:45-69,112-113,154-163,188-209,215-219,233-247,
251-256,313-322,327-345,379-411,461-468) and its own wrappers
``vllm_metax/_custom_ops.py:7-62`` (awq_gemm, awq_to_gptq_4bit, gptq_gemm,
gptq_shuffle).  Every function extracts raw pointers / strides from the torch tensors
and calls the C-ABI (include/mi355x_hotpath.h) on torch's current HIP stream — torch
is plumbing only.  Errors surface as RuntimeError, like the reference's TORCH_CHECK.

There is deliberately no CPU implementation here: calling an op without the built
HIP library, or with CPU tensors, raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

from . import _abi

_DT = {torch.float16: _abi.F16, torch.bfloat16: _abi.BF16, torch.float32: _abi.F32}
PARTITION_SIZE = 512  # == MI355X_PA_PARTITION_SIZE (paged_attention_v2.cu:45)


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() or None


def _dev(*ts: Optional[torch.Tensor]) -> None:
    """Every tensor on the GPU, and on the CURRENT device: the C-ABI launches on torch's current stream,
    which belongs to the current device (torch_bindings.cpp has a device guard for the same reason)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("expected a tensor on the GPU (hip) device, got " + str(t.device))
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{cur}: wrap the call in "
                               f"torch.cuda.device({t.device.index})")


_RETIRED: list = []   # scratch buffers that were outgrown: kept alive, a captured HIP graph may still use them


def _grow(table: dict, key, nbytes_or_elems: int, dtype) -> torch.Tensor:
    """Per-device scratch, grown on demand.  An outgrown buffer is retired, not freed: a HIP graph
    captured earlier keeps launching kernels that write to its address (scratch contents never outlive
    a call, so the only hazard would be the allocator handing that memory to somebody else)."""
    buf = table.get(key)
    if buf is None or buf.numel() < nbytes_or_elems:
        if buf is not None:
            _RETIRED.append(buf)
        buf = torch.empty(nbytes_or_elems, dtype=dtype, device=key)
        table[key] = buf
    return buf


def _kv_dtype(kv_cache_dtype: str, cache: torch.Tensor, k_scale, v_scale):
    """`str kv_cache_dtype` of the schema -> (mi355x_kv_cache_dtype, k_scale ptr, v_scale ptr).
    "auto": the cache holds scalar_t (all the reference accepts: csrc/quantization/fp8/metax/
    quant_utils.cuh:29-42).  "fp8" / "fp8_e4m3": OCP e4m3fn bytes (upstream vLLM's
    names), "fp8_e5m2": e5m2 bytes (csrc/attention/dtype_fp8.cuh:9-13 lists both), cache tensors of a 1-byte dtype, k_scale / v_scale one float32 each on the
    device.  Anything else raises like the reference's TORCH_CHECK."""
    if kv_cache_dtype == "auto":
        return _abi.KV_AUTO, None, None
    if kv_cache_dtype in ("fp8", "fp8_e4m3", "fp8_e5m2"):
        if cache.element_size() != 1:
            raise RuntimeError(f"kv_cache_dtype {kv_cache_dtype!r} needs a 1-byte cache, got {cache.dtype}")
        for name, t in (("k_scale", k_scale), ("v_scale", v_scale)):
            if t is None or not t.is_cuda or t.dtype != torch.float32 or t.numel() != 1:
                raise RuntimeError(f"{name} must be one float32 element on the GPU for an fp8 KV cache")
        return (_abi.KV_FP8_E5M2 if kv_cache_dtype == "fp8_e5m2" else _abi.KV_FP8_E4M3), _ptr(k_scale), _ptr(v_scale)
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


# ----------------------------------------------------------------------------- utils
def get_device_attribute(attribute: int, device_id: int) -> int:
    v = _abi.load().mi355x_get_device_attribute(attribute, device_id)
    if v < 0:
        raise RuntimeError(_abi.last_error())
    return v


def get_max_shared_memory_per_block_device_attribute(device_id: int) -> int:
    v = _abi.load().mi355x_get_max_shared_memory_per_block_device_attribute(device_id)
    if v < 0:
        raise RuntimeError(_abi.last_error())
    return v


# ------------------------------------------------------------------------- cache ops
def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                      kv_cache_dtype: str = "auto", k_scale: Optional[torch.Tensor] = None,
                      v_scale: Optional[torch.Tensor] = None) -> None:
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    _dev(key, value, key_cache, value_cache, slot_mapping)
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    num_tokens = slot_mapping.size(0)
    num_heads, head_size = key.size(1), key.size(2)
    block_size, x = key_cache.size(3), key_cache.size(4)
    rc = _abi.load().mi355x_reshape_and_cache(
        _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
        num_tokens, key.stride(0), value.stride(0), num_heads, head_size, block_size, x,
        _dt(key), kvd, ks, vs, _stream())
    _abi.check(rc, "reshape_and_cache")


def reshape_and_cache_flash(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                            kv_cache_dtype: str = "auto",
                            k_scale: Optional[torch.Tensor] = None,
                            v_scale: Optional[torch.Tensor] = None) -> None:
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    _dev(key, value, key_cache, value_cache, slot_mapping)
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    if key_cache.stride(0) != value_cache.stride(0):
        raise RuntimeError("key_cache.stride(0) != value_cache.stride(0)")
    num_tokens = slot_mapping.size(0)  # key may be longer (graph padding)
    rc = _abi.load().mi355x_reshape_and_cache_flash(
        _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
        num_tokens, key_cache.stride(0), key_cache.stride(1), key_cache.stride(2),
        key.stride(0), value.stride(0), key.size(1), key.size(2), key_cache.size(1),
        _dt(key), kvd, ks, vs, _stream())
    _abi.check(rc, "reshape_and_cache_flash")


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0,
                kv_dtype: str = "fp8") -> None:
    """torch.ops._C_cache_ops.convert_fp8 (csrc/torch_bindings.cpp:424-426, cache_kernels.cu:564-612):
    elementwise over the flat cache; direction from the dtypes (1-byte side = e4m3 bytes)."""
    _dev(output, input)
    if kv_dtype not in ("fp8", "fp8_e4m3", "fp8_e5m2"):
        raise RuntimeError(f"Unsupported data type: {kv_dtype}")
    if output.numel() != input.numel() or not (output.is_contiguous() and input.is_contiguous()):
        raise RuntimeError("convert_fp8: tensors must be contiguous and of the same size")
    to_fp8 = output.element_size() == 1
    wide = input if to_fp8 else output
    if (input if to_fp8 else output).element_size() == 1 or (output if to_fp8 else input).element_size() != 1:
        raise RuntimeError("convert_fp8: exactly one of the tensors must be a 1-byte (fp8) tensor")
    rc = _abi.load().mi355x_convert_fp8(_ptr(output), _ptr(input), input.numel(), float(scale),
                                        (1 if to_fp8 else 0) + (2 if kv_dtype == "fp8_e5m2" else 0), _dt(wide),
                                        _stream())
    _abi.check(rc, "convert_fp8")


def copy_blocks(key_caches: Sequence[torch.Tensor], value_caches: Sequence[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    n = len(key_caches)
    if n != len(value_caches):
        raise RuntimeError("copy_blocks: key_caches and value_caches differ in length")
    if n == 0:
        return
    _dev(*key_caches, *value_caches, block_mapping)
    if block_mapping.dtype != torch.int64:
        raise RuntimeError("block_mapping must be int64")
    kp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in key_caches])
    vp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in value_caches])
    bytes_per_block = key_caches[0][0].numel() * key_caches[0].element_size()
    bm = block_mapping.contiguous()
    rc = _abi.load().mi355x_copy_blocks(kp, vp, n, _ptr(bm), bm.size(0), bytes_per_block,
                                        _stream())
    _abi.check(rc, "copy_blocks")


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    if block_mapping.device.type != "cpu":
        raise RuntimeError("block_mapping must be on CPU")
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")
        kind = 0
    elif src.is_cuda and dst.device.type == "cpu":
        kind = 1
    elif src.device.type == "cpu" and dst.is_cuda:
        kind = 2
    else:
        raise RuntimeError("Invalid device combination")
    bm = block_mapping.to(torch.int64).contiguous()
    block_bytes = src.element_size() * src.stride(0)
    rc = _abi.load().mi355x_swap_blocks(_ptr(src), _ptr(dst), bm.data_ptr(), bm.size(0),
                                        block_bytes, kind, _stream())
    _abi.check(rc, "swap_blocks")


# ------------------------------------------------------------------- paged attention
def _check_blocksparse(blocksparse_vert_stride: int) -> None:
    if blocksparse_vert_stride > 1:
        raise RuntimeError("block-sparse paged attention is not supported by the MI355X plugin")


def paged_attention_v1(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                       max_seq_len: int, alibi_slopes: Optional[torch.Tensor],
                       kv_cache_dtype: str = "auto", k_scale: Optional[torch.Tensor] = None,
                       v_scale: Optional[torch.Tensor] = None, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> None:
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    _check_blocksparse(blocksparse_vert_stride)
    _dev(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes)
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("block_tables and seq_lens must be int32")
    rc = _abi.load().mi355x_paged_attention_v1(
        _ptr(out), _ptr(query), _ptr(key_cache), _ptr(value_cache), query.size(0),
        query.size(1), num_kv_heads, query.size(2), block_size, float(scale),
        _ptr(block_tables), _ptr(seq_lens), block_tables.size(1), max_seq_len,
        _ptr(alibi_slopes), query.stride(0), key_cache.stride(0), key_cache.stride(1),
        _dt(query), kvd, ks, vs, _stream())
    _abi.check(rc, "paged_attention_v1")


def paged_attention_v2(out: torch.Tensor, exp_sums: torch.Tensor, max_logits: torch.Tensor,
                       tmp_out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                       max_seq_len: int, alibi_slopes: Optional[torch.Tensor],
                       kv_cache_dtype: str = "auto", k_scale: Optional[torch.Tensor] = None,
                       v_scale: Optional[torch.Tensor] = None, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0,
                       partition_size: int = PARTITION_SIZE) -> None:
    """`partition_size` (beyond the reference's signature, default = its fixed 512): tokens per split-KV partition;
    exp_sums / max_logits / tmp_out must hold ceil(max_seq_len / partition_size) partitions."""
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    _check_blocksparse(blocksparse_vert_stride)
    _dev(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, block_tables,
         seq_lens, alibi_slopes)
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("block_tables and seq_lens must be int32")
    parts = (max_seq_len + partition_size - 1) // max(partition_size, 1)
    if partition_size <= 0 or exp_sums.numel() < query.size(0) * query.size(1) * parts \
            or tmp_out.numel() < query.size(0) * query.size(1) * parts * query.size(2):
        raise RuntimeError(f"paged_attention_v2: workspaces too small for {parts} partitions of {partition_size}")
    rc = _abi.load().mi355x_paged_attention_v2_ps(
        _ptr(out), _ptr(exp_sums), _ptr(max_logits), _ptr(tmp_out), _ptr(query),
        _ptr(key_cache), _ptr(value_cache), query.size(0), query.size(1), num_kv_heads,
        query.size(2), block_size, float(scale), _ptr(block_tables), _ptr(seq_lens),
        block_tables.size(1), max_seq_len, _ptr(alibi_slopes), query.stride(0),
        key_cache.stride(0), key_cache.stride(1), _dt(query), kvd, ks, vs, int(partition_size), _stream())
    _abi.check(rc, "paged_attention_v2")


def paged_attention_v1_max_seq_len(num_seqs: int, num_heads: int, num_kv_heads: int, head_size: int,
                                   block_size: int, dtype: torch.dtype) -> int:
    """Largest max_seq_len paged_attention_v1 takes for this geometry (its logits live in one
    workgroup's 160 KiB of LDS); longer contexts must go to paged_attention_v2."""
    v = _abi.load().mi355x_paged_attention_v1_max_seq_len(num_seqs, num_heads, num_kv_heads, head_size,
                                                         block_size, _DT[dtype])
    if v < 0:
        raise RuntimeError(_abi.last_error())
    return v


def paged_prefill_attention(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                            block_tables: torch.Tensor, seq_lens: torch.Tensor,
                            cu_seqlens_q: torch.Tensor, max_query_len: int,
                            block_size: int, kv_cache_dtype: str = "auto",
                            k_scale: Optional[torch.Tensor] = None,
                            v_scale: Optional[torch.Tensor] = None, sliding_window: Optional[int] = None,
                            softcap: Optional[float] = None, alibi_slopes: Optional[torch.Tensor] = None) -> None:
    """Varlen causal attention of the new tokens against the paged cache (the role of
    flash_attn_varlen_func(block_table=..., window_size, softcap, alibi_slopes) at the reference call site
    vllm_metax/v1/attention/backends/flash_attn.py:725-747)."""
    _dev(out, query, key_cache, value_cache, block_tables, seq_lens, cu_seqlens_q, alibi_slopes)
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    if alibi_slopes is not None:
        if alibi_slopes.dtype != torch.float32 or alibi_slopes.numel() != query.size(1) \
                or not alibi_slopes.is_contiguous():
            raise RuntimeError("paged_prefill_attention: alibi_slopes must be a contiguous float32 [num_heads]")
        rc = _abi.load().mi355x_paged_prefill_attention_alibi(
            _ptr(out), _ptr(query), _ptr(key_cache), _ptr(value_cache), seq_lens.size(0),
            query.size(1), num_kv_heads, query.size(2), block_size, float(scale),
            _ptr(block_tables), _ptr(seq_lens), _ptr(cu_seqlens_q), max_query_len,
            block_tables.size(1), query.stride(0), out.stride(0), key_cache.stride(0),
            key_cache.stride(1), _dt(query), kvd, ks, vs, int(sliding_window or 0), float(softcap or 0.0),
            _ptr(alibi_slopes), _stream())
        _abi.check(rc, "paged_prefill_attention")
        return
    rc = _abi.load().mi355x_paged_prefill_attention(
        _ptr(out), _ptr(query), _ptr(key_cache), _ptr(value_cache), seq_lens.size(0),
        query.size(1), num_kv_heads, query.size(2), block_size, float(scale),
        _ptr(block_tables), _ptr(seq_lens), _ptr(cu_seqlens_q), max_query_len,
        block_tables.size(1), query.stride(0), out.stride(0), key_cache.stride(0),
        key_cache.stride(1), _dt(query), kvd, ks, vs, int(sliding_window or 0), float(softcap or 0.0),
        _stream())
    _abi.check(rc, "paged_prefill_attention")


# ------------------------------------------------------------------------- layernorm
def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
             epsilon: float) -> None:
    _dev(out, input, weight)
    if not out.is_contiguous():
        raise RuntimeError("rms_norm: out must be contiguous")
    if input.stride(-1) != 1 or weight.stride(-1) != 1:
        raise RuntimeError("rms_norm: innermost stride must be 1")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden if hidden else 0
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    rc = _abi.load().mi355x_rms_norm(_ptr(out), _ptr(input), _ptr(weight), float(epsilon),
                                     num_tokens, hidden, in_stride, _dt(input), _stream())
    _abi.check(rc, "rms_norm")


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       epsilon: float) -> None:
    _dev(input, residual, weight)
    if not residual.is_contiguous() or not weight.is_contiguous():
        raise RuntimeError("fused_add_rms_norm: residual and weight must be contiguous")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden if hidden else 0
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    rc = _abi.load().mi355x_fused_add_rms_norm(_ptr(input), _ptr(residual), _ptr(weight),
                                               float(epsilon), num_tokens, hidden, in_stride,
                                               _dt(input), _stream())
    _abi.check(rc, "fused_add_rms_norm")


def fused_add_rms_norm_slabs(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                             slabs: Optional[torch.Tensor], sk: int, epsilon: float) -> None:
    """MI355X-side fusion: fused_add_rms_norm whose input is still the `sk` fp32 split-K slabs of
    the preceding decode GEMM (awq_gemm_deferred); sk == 0 -> plain fused_add_rms_norm.  `input`
    receives the normalised rows.  Bit-identical to awq_gemm + fused_add_rms_norm."""
    _dev(input, residual, weight)
    if not residual.is_contiguous() or not weight.is_contiguous():
        raise RuntimeError("fused_add_rms_norm_slabs: residual and weight must be contiguous")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden if hidden else 0
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    if sk > 0:
        _dev(slabs)
        if slabs.dtype != torch.float32 or slabs.numel() < sk * num_tokens * hidden:
            raise RuntimeError("fused_add_rms_norm_slabs: slabs must be float32 [sk, tokens, hidden]")
    rc = _abi.load().mi355x_fused_add_rms_norm_slabs(
        _ptr(input), _ptr(residual), _ptr(weight), _ptr(slabs) if sk > 0 else None, int(sk),
        float(epsilon), num_tokens, hidden, in_stride, _dt(input), _stream())
    _abi.check(rc, "fused_add_rms_norm_slabs")


def _check_fp8(out: torch.Tensor) -> None:
    if out.dtype != torch.float8_e4m3fn:
        raise RuntimeError(f"expected a float8_e4m3fn output, got {out.dtype}")


def rms_norm_static_fp8_quant(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
                              scale: torch.Tensor, epsilon: float) -> None:
    _dev(out, input, weight, scale)
    _check_fp8(out)
    if not out.is_contiguous():
        raise RuntimeError("out must be contiguous")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    rc = _abi.load().mi355x_rms_norm_static_fp8_quant(
        _ptr(out), _ptr(input), _ptr(weight), _ptr(scale), float(epsilon), num_tokens, hidden,
        in_stride, _dt(input), _stream())
    _abi.check(rc, "rms_norm_static_fp8_quant")


def fused_add_rms_norm_static_fp8_quant(out: torch.Tensor, input: torch.Tensor,
                                        residual: torch.Tensor, weight: torch.Tensor,
                                        scale: torch.Tensor, epsilon: float) -> None:
    _dev(out, input, residual, weight, scale)
    _check_fp8(out)
    if not out.is_contiguous() or not residual.is_contiguous():
        raise RuntimeError("out and residual must be contiguous")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    rc = _abi.load().mi355x_fused_add_rms_norm_static_fp8_quant(
        _ptr(out), _ptr(input), _ptr(residual), _ptr(weight), _ptr(scale), float(epsilon),
        num_tokens, hidden, in_stride, _dt(input), _stream())
    _abi.check(rc, "fused_add_rms_norm_static_fp8_quant")


def rms_norm_dynamic_per_token_quant(out: torch.Tensor, input: torch.Tensor,
                                     weight: torch.Tensor, scales: torch.Tensor,
                                     epsilon: float, scale_ub: Optional[torch.Tensor] = None,
                                     residual: Optional[torch.Tensor] = None) -> None:
    _dev(out, input, weight, scales, scale_ub, residual)
    _check_fp8(out)  # the int8 branch of the reference is out of scope
    if not (out.is_contiguous() and input.is_contiguous()):
        raise RuntimeError("out and input must be contiguous")
    if scales.dtype != torch.float32:
        raise RuntimeError("scales must be float32")
    hidden = input.size(-1)
    num_tokens = input.numel() // hidden
    rc = _abi.load().mi355x_rms_norm_dynamic_per_token_quant(
        _ptr(out), _ptr(input), _ptr(weight), _ptr(scales), float(epsilon), _ptr(scale_ub),
        _ptr(residual), num_tokens, hidden, _dt(input), _stream())
    _abi.check(rc, "rms_norm_dynamic_per_token_quant")


def rms_norm_dynamic_per_token_quant_slabs(out: torch.Tensor, slabs: torch.Tensor, sk: int, a_scales: torch.Tensor,
                                           b_scales: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor,
                                           epsilon: float, scale_ub: Optional[torch.Tensor] = None,
                                           residual: Optional[torch.Tensor] = None) -> None:
    """rms_norm_dynamic_per_token_quant whose input rows are the split-K slabs [sk, tokens, hidden] (float32) of an
    fp8 GEMM (scaled_mm_fp8_deferred; the GEMM's output type = weight.dtype): bit-identical to the GEMM's finish
    launch followed by that op."""
    _dev(out, slabs, a_scales, b_scales, weight, scales, scale_ub, residual)
    _check_fp8(out)
    hidden = out.size(-1)
    num_tokens = out.numel() // hidden
    if slabs.dtype != torch.float32 or slabs.numel() < sk * num_tokens * hidden or sk <= 0:
        raise RuntimeError("rms_norm_dynamic_per_token_quant_slabs: slabs must be float32 [sk, tokens, hidden]")
    if not out.is_contiguous() or scales.dtype != torch.float32:
        raise RuntimeError("rms_norm_dynamic_per_token_quant_slabs: contiguous out, float32 scales")
    rc = _abi.load().mi355x_rms_norm_dynamic_per_token_quant_slabs(
        _ptr(out), _ptr(slabs), int(sk), _ptr(a_scales), a_scales.numel(), _ptr(b_scales), b_scales.numel(),
        _ptr(weight), _ptr(scales), float(epsilon), _ptr(scale_ub), _ptr(residual), num_tokens, hidden,
        _dt(weight), _stream())
    _abi.check(rc, "rms_norm_dynamic_per_token_quant_slabs")


# ------------------------------------------------------------------------- fp8 quant
def _rows(t: torch.Tensor):
    hidden = t.size(-1)
    num_tokens = t.numel() // hidden if hidden else 0
    stride = t.stride(-2) if t.dim() >= 2 else hidden
    return num_tokens, hidden, stride


def static_scaled_fp8_quant(out: torch.Tensor, input: torch.Tensor,
                            scale: torch.Tensor) -> None:
    _dev(out, input, scale)
    _check_fp8(out)
    if input.stride(-1) != 1 or out.stride(-1) != 1:
        raise RuntimeError("last dimension must be contiguous")
    n, h, s_in = _rows(input)
    _, _, s_out = _rows(out)
    rc = _abi.load().mi355x_static_scaled_fp8_quant(_ptr(out), _ptr(input), _ptr(scale), n, h,
                                                    s_in, s_out, _dt(input), _stream())
    _abi.check(rc, "static_scaled_fp8_quant")


def dynamic_scaled_fp8_quant(out: torch.Tensor, input: torch.Tensor,
                             scale: torch.Tensor) -> None:
    """`scale` must be zero-initialised by the caller (as upstream's scaled_fp8_quant does)."""
    _dev(out, input, scale)
    _check_fp8(out)
    if input.stride(-1) != 1 or out.stride(-1) != 1:
        raise RuntimeError("last dimension must be contiguous")
    n, h, s_in = _rows(input)
    _, _, s_out = _rows(out)
    rc = _abi.load().mi355x_dynamic_scaled_fp8_quant(_ptr(out), _ptr(input), _ptr(scale), n, h,
                                                     s_in, s_out, _dt(input), _stream())
    _abi.check(rc, "dynamic_scaled_fp8_quant")


def dynamic_per_token_scaled_fp8_quant(out: torch.Tensor, input: torch.Tensor,
                                       scales: torch.Tensor,
                                       scale_ub: Optional[torch.Tensor] = None) -> None:
    _dev(out, input, scales, scale_ub)
    _check_fp8(out)
    if input.stride(-1) != 1 or out.stride(-1) != 1:
        raise RuntimeError("last dimension must be contiguous")
    n, h, s_in = _rows(input)
    _, _, s_out = _rows(out)
    rc = _abi.load().mi355x_dynamic_per_token_scaled_fp8_quant(
        _ptr(out), _ptr(input), _ptr(scales), _ptr(scale_ub), n, h, s_in, s_out, _dt(input),
        _stream())
    _abi.check(rc, "dynamic_per_token_scaled_fp8_quant")


# ---------------------------------------------------------------------------- rotary
def batched_rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: Optional[torch.Tensor],
                             head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool, rot_dim: int,
                             cos_sin_cache_offsets: torch.Tensor) -> None:
    """ref launcher: csrc/pos_encoding_kernels.cu:219-306 (row = position + offset of the token's LoRA)."""
    if cos_sin_cache_offsets.dtype != torch.int64 or cos_sin_cache_offsets.numel() != positions.numel():
        raise RuntimeError("positions must have the same num_tokens or batch_size as cos_sin_cache_offsets")
    if rot_dim != cos_sin_cache.size(1):
        raise RuntimeError("rot_dim must equal cos_sin_cache.size(1)")
    rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, cos_sin_cache_offsets)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor,
                     key: Optional[torch.Tensor], head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool,
                     _offsets: Optional[torch.Tensor] = None) -> None:
    """ref launcher: csrc/pos_encoding_kernels.cu:133-213 (shape / stride handling)."""
    _dev(positions, query, key, cos_sin_cache, _offsets)
    if positions.dtype != torch.int64:
        raise RuntimeError("positions must be int64")
    num_tokens = positions.numel()
    pdim = positions.dim()
    if pdim not in (1, 2):
        raise RuntimeError("positions must have shape [num_tokens] or [batch_size, seq_len]")
    if pdim == 1:
        if query.size(0) != positions.size(0) or (key is not None and key.size(0) != positions.size(0)):
            raise RuntimeError("query, key and positions must have the same number of tokens")
    else:
        if (query.size(0) != positions.size(0) or query.size(1) != positions.size(1)
                or (key is not None and (key.size(0) != positions.size(0)
                                         or key.size(1) != positions.size(1)))):
            raise RuntimeError("query, key and positions must have the same batch_size and seq_len")
    q_hidden = query.numel() // num_tokens
    k_hidden = key.numel() // num_tokens if key is not None else 0
    if q_hidden % head_size or k_hidden % head_size:
        raise RuntimeError("hidden size must be a multiple of head_size")
    num_heads = q_hidden // head_size
    num_kv_heads = k_hidden // head_size if key is not None else num_heads
    if num_heads % num_kv_heads:
        raise RuntimeError("num_heads must be a multiple of num_kv_heads")
    rot_dim = cos_sin_cache.size(1)
    seq_dim_idx = pdim - 1
    query_stride = query.stride(seq_dim_idx)
    key_stride = key.stride(seq_dim_idx) if key is not None else 0
    query_ndim = query.dim()
    head_stride = query.stride(-2) if query_ndim == pdim + 2 else head_size
    if _offsets is not None:
        rc = _abi.load().mi355x_batched_rotary_embedding(
            _ptr(positions), _ptr(query), _ptr(key), _ptr(cos_sin_cache), _ptr(_offsets), num_tokens,
            rot_dim, query_stride, key_stride, head_stride, num_heads, num_kv_heads, head_size,
            1 if is_neox else 0, _dt(query), _stream())
        _abi.check(rc, "batched_rotary_embedding")
        return
    rc = _abi.load().mi355x_rotary_embedding(
        _ptr(positions), _ptr(query), _ptr(key), _ptr(cos_sin_cache), num_tokens, rot_dim,
        query_stride, key_stride, head_stride, num_heads, num_kv_heads, head_size,
        1 if is_neox else 0, _dt(query), _stream())
    _abi.check(rc, "rotary_embedding")


# ------------------------------------------------------------------------ activation
def silu_and_mul_per_token_quant(input: torch.Tensor):
    """MI355X-side fusion: silu_and_mul(input) quantised per token to fp8 in the same launch -> (fp8 [T, d],
    scales [T, 1]), bit-identical to silu_and_mul + dynamic_per_token_scaled_fp8_quant; None when the fused form
    does not apply to these shapes."""
    _dev(input)
    if input.dim() != 2 or not input.is_contiguous() or input.dtype not in (torch.bfloat16, torch.float16):
        return None
    t, d2 = input.shape
    d = d2 // 2
    out = torch.empty((t, d), dtype=torch.float8_e4m3fn, device=input.device)
    scales = torch.empty((t, 1), dtype=torch.float32, device=input.device)
    rc = _abi.load().mi355x_silu_and_mul_per_token_quant(_ptr(out), _ptr(scales), _ptr(input), t, d, _dt(input),
                                                         _stream())
    if rc == 1:
        return None
    _abi.check(rc, "silu_and_mul_per_token_quant")
    return out, scales


def silu_and_mul_per_token_quant_slabs(slabs: torch.Tensor, sk: int, a_scales: torch.Tensor, b_scales: torch.Tensor,
                                       num_tokens: int, d: int, dtype: torch.dtype):
    """silu_and_mul_per_token_quant on the split-K slabs [sk, num_tokens, 2 d] (float32) of an fp8 gate_up GEMM
    (scaled_mm_fp8_deferred) whose output type would have been `dtype`: bit-identical to the GEMM's finish launch
    followed by that op.  None when not applicable."""
    _dev(slabs, a_scales, b_scales)
    if slabs.dtype != torch.float32 or slabs.numel() < sk * num_tokens * 2 * d or sk <= 0:
        raise RuntimeError("silu_and_mul_per_token_quant_slabs: slabs must be float32 [sk, tokens, 2 d]")
    out = torch.empty((num_tokens, d), dtype=torch.float8_e4m3fn, device=slabs.device)
    scales = torch.empty((num_tokens, 1), dtype=torch.float32, device=slabs.device)
    rc = _abi.load().mi355x_silu_and_mul_per_token_quant_slabs(
        _ptr(out), _ptr(scales), _ptr(slabs), int(sk), _ptr(a_scales), a_scales.numel(), _ptr(b_scales),
        b_scales.numel(), num_tokens, d, _DT[dtype], _stream())
    if rc == 1:
        return None
    _abi.check(rc, "silu_and_mul_per_token_quant_slabs")
    return out, scales


def silu_and_mul(out: torch.Tensor, input: torch.Tensor) -> None:
    _dev(out, input)
    if not (out.is_contiguous() and input.is_contiguous()):
        raise RuntimeError("silu_and_mul: tensors must be contiguous")
    d = input.size(-1) // 2
    num_tokens = input.numel() // input.size(-1) if d else 0
    rc = _abi.load().mi355x_silu_and_mul(_ptr(out), _ptr(input), num_tokens, d, _dt(input),
                                         _stream())
    _abi.check(rc, "silu_and_mul")


def silu_and_mul_quant(out: torch.Tensor, input: torch.Tensor, scale: torch.Tensor) -> None:
    """torch.ops._C.silu_and_mul_quant (csrc/quantization/activation_kernels.cu:117-127): out is
    fp8 e4m3 [..., d], input fp16 / bf16 [..., 2 d], scale one fp32 element."""
    _dev(out, input, scale)
    if out.dtype != torch.float8_e4m3fn:
        raise RuntimeError("silu_and_mul_quant: out must be torch.float8_e4m3fn")
    if input.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("silu_and_mul_quant: input must be float16 or bfloat16")
    if input.size(-1) % 2 != 0:
        raise RuntimeError("silu_and_mul_quant: last dimension of input must be even")
    if scale.dtype != torch.float32 or scale.numel() != 1:
        raise RuntimeError("silu_and_mul_quant: scale must be one float32 element")
    if not (out.is_contiguous() and input.is_contiguous()):
        raise RuntimeError("silu_and_mul_quant: tensors must be contiguous")
    d = input.size(-1) // 2
    num_tokens = input.numel() // input.size(-1) if d else 0
    rc = _abi.load().mi355x_silu_and_mul_quant(_ptr(out), _ptr(input), _ptr(scale), num_tokens, d,
                                               _dt(input), _stream())
    _abi.check(rc, "silu_and_mul_quant")


# ----------------------------------------------------------------- int4 weight-only
_DQ_SCRATCH: dict = {}


def _dq_scratch(m: int, n: int, k: int, device) -> Optional[torch.Tensor]:
    """Scratch for the dequantised weights of a prefill-sized GEMM (one buffer per device,
    grown to the largest n*k seen; the kernels of one stream run in order, so sharing is safe)."""
    m_pad = (m + 15) // 16 * 16
    if m <= 64:
        need = m_pad * k * 2                    # decode: packed activations only
    elif m < 1024:
        need = 8 * 128 * n * 4                  # 64 < m < 1024: fp32 split-K slabs of one 128-row pass
    else:
        # prefill: packed weights + packed activations + the partial tiles of a K split (few 256 x 256 tiles)
        need = ((n + m_pad) * k * 2 + 15) // 16 * 16 + 4 * int(_abi.load().mi355x_w4a16_prepacked_split_elems(m, n, k))
    return _grow(_DQ_SCRATCH, device, need, torch.uint8)


def awq_to_gptq_4bit(qweight: torch.Tensor) -> torch.Tensor:
    """ref: vllm_metax/_custom_ops.py:26-29; csrc/quantization/awq/gemm_kernels.cu:323-356.
    Returns a tensor DECLARED [N, K/8] whose memory is [K/8, N] (the reference's quirk)."""
    _dev(qweight)
    if qweight.dtype != torch.int32 or not qweight.is_contiguous():
        raise RuntimeError("awq_to_gptq_4bit: qweight must be a contiguous int32 tensor")
    k, n = qweight.size(0), qweight.size(1) * 8
    out = torch.zeros((n, (k + 7) // 8), dtype=qweight.dtype, device=qweight.device)
    rc = _abi.load().mi355x_awq_to_gptq_4bit(_ptr(out), _ptr(qweight), k, n, _stream())
    _abi.check(rc, "awq_to_gptq_4bit")
    return out


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor,
                   split_k_iters: int = 0, thx: int = 0, thy: int = 0) -> torch.Tensor:
    _dev(qweight, scales, zeros)
    k, n = qweight.size(0), qweight.size(1) * 8
    group = k // scales.size(0)
    out = torch.empty((k, n), dtype=scales.dtype, device=scales.device)
    rc = _abi.load().mi355x_awq_dequantize(_ptr(out), _ptr(qweight), _ptr(scales), _ptr(zeros),
                                           k, n, group, _dt(scales), _stream())
    _abi.check(rc, "awq_dequantize")
    return out


def awq_gemm(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
             scales: torch.Tensor, split_k_iters: int, temp_space: torch.Tensor,
             dtype_bf16: bool) -> torch.Tensor:
    """ref: vllm_metax/_custom_ops.py:7-22 (note the (qzeros, scales) order of the wrapper)
    -> torch.ops._C.awq_gemm(input, qweight, scales, qzeros, split_k_iters, temp_space, bf16)."""
    _dev(input, qweight, qzeros, scales)
    if dtype_bf16 != (input.dtype == torch.bfloat16):
        raise RuntimeError("awq_gemm: dtype_bf16 does not match the input dtype")
    if input.dim() != 2 or input.stride(1) != 1:
        raise RuntimeError("awq_gemm: input must be [M, K] with unit inner stride")
    m, k = input.shape
    n = qweight.size(0)  # declared [N, K/8]
    group = k // scales.size(0)
    out = torch.empty((m, n), dtype=input.dtype, device=input.device)
    ws = temp_space if (temp_space is not None and temp_space.is_cuda
                        and temp_space.dtype == torch.float32 and temp_space.numel() > 0) else None
    dq = _dq_scratch(m, n, k, input.device)
    rc = _abi.load().mi355x_awq_gemm(_ptr(out), _ptr(input), _ptr(qweight), _ptr(scales),
                                     _ptr(qzeros), _ptr(ws), ws.numel() if ws is not None else 0,
                                     _ptr(dq), dq.numel() if dq is not None else 0,
                                     m, n, k, group, input.stride(0), _dt(input), _stream())
    _abi.check(rc, "awq_gemm")
    return out


def merge_attn_states(output: torch.Tensor, prefix_output: torch.Tensor, prefix_lse: torch.Tensor,
                      suffix_output: torch.Tensor, suffix_lse: torch.Tensor,
                      output_lse: Optional[torch.Tensor] = None) -> None:
    """ref: vllm._custom_ops.merge_attn_states -> torch.ops._C.merge_attn_states(output, output_lse,
    prefix_output, prefix_lse, suffix_output, suffix_lse) (csrc/torch_bindings.cpp:74-82);
    checks as the launcher csrc/attention/merge_attn_states.cu:133-160."""
    _dev(output, prefix_output, prefix_lse, suffix_output, suffix_lse)
    if output.dim() != 3:
        raise RuntimeError("merge_attn_states: output must be [num_tokens, num_heads, head_size]")
    n, h, dsz = output.shape
    for name, t in (("output", output), ("prefix_output", prefix_output), ("suffix_output", suffix_output)):
        if t.shape != output.shape or t.dtype != output.dtype:
            raise RuntimeError(f"merge_attn_states: {name} must match output's shape and dtype")
        if t.stride(-2) != dsz or t.stride(-1) != 1 or t.stride(0) != h * dsz:
            raise RuntimeError(f"{name} heads must be contiguous in memory")
    for name, t in (("prefix_lse", prefix_lse), ("suffix_lse", suffix_lse), ("output_lse", output_lse)):
        if t is None:
            continue
        if t.dtype != torch.float32 or tuple(t.shape) != (h, n) or not t.is_contiguous():
            raise RuntimeError(f"merge_attn_states: {name} must be a contiguous float32 [num_heads, num_tokens]")
    if output_lse is not None:
        _dev(output_lse)
    rc = _abi.load().mi355x_merge_attn_states(
        _ptr(output), _ptr(output_lse), _ptr(prefix_output), _ptr(prefix_lse), _ptr(suffix_output),
        _ptr(suffix_lse), n, h, dsz, _dt(output), _stream())
    _abi.check(rc, "merge_attn_states")


def qkv_rope_cache(qkv: torch.Tensor, slabs: Optional[torch.Tensor], sk: int,
                   positions: torch.Tensor, cos_sin_cache: torch.Tensor, key_cache: torch.Tensor,
                   value_cache: torch.Tensor, slot_mapping: torch.Tensor, num_heads: int,
                   num_kv_heads: int, head_size: int) -> None:
    """MI355X-side decode fusion: [slab sum ->] NeoX rotary on q, k -> reshape_and_cache, one launch.
    qkv [T, (H + 2 KVH) * D] is updated in place (rotated q, k); caches as reshape_and_cache."""
    _dev(qkv, positions, cos_sin_cache, key_cache, value_cache, slot_mapping)
    if qkv.dim() != 2 or qkv.stride(1) != 1:
        raise RuntimeError("qkv_rope_cache: qkv must be [tokens, width] with unit inner stride")
    if key_cache.dim() != 5 or value_cache.dim() != 4:
        raise RuntimeError("qkv_rope_cache: caches must be in the x-split layout")
    if positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64:
        raise RuntimeError("qkv_rope_cache: positions and slot_mapping must be int64")
    if cos_sin_cache.dtype != qkv.dtype or cos_sin_cache.size(-1) != head_size:
        raise RuntimeError("qkv_rope_cache: cos_sin_cache must be [max_pos, head_size] in the qkv dtype")
    rc = _abi.load().mi355x_qkv_rope_cache(
        _ptr(qkv), qkv.stride(0), _ptr(slabs) if sk > 0 else None, int(sk), _ptr(positions),
        _ptr(cos_sin_cache), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping), qkv.size(0),
        num_heads, num_kv_heads, head_size, value_cache.size(3), key_cache.size(4),
        key_cache.stride(0), value_cache.stride(0), _dt(qkv), _stream())
    _abi.check(rc, "qkv_rope_cache")


def awq_gemm_silu_mul(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                      scales: torch.Tensor) -> Optional[torch.Tensor]:
    """MI355X-side prefill fusion: silu_and_mul(awq_gemm(input, W_gate_up)) -> [M, N/2] in one GEMM
    launch, bit-identical to the two ops (decode M <= 64: stripe kernel; prefill M >= 1024: packed
    GEMM).  Returns None when the fused path does not apply (64 < M < 1024, odd shapes): the caller
    then runs awq_gemm + silu_and_mul."""
    _dev(input, qweight, qzeros, scales)
    if input.dim() != 2 or input.stride(1) != 1:
        raise RuntimeError("awq_gemm_silu_mul: input must be [M, K] with unit inner stride")
    m, k = input.shape
    n = qweight.size(0)
    if input.dtype not in (torch.bfloat16, torch.float16):
        return None
    group = k // scales.size(0)
    if m <= 64:                       # decode: stripe kernel with the silu epilogue
        if n % 128 != 0 or k % 128 != 0 or not (group % 128 == 0 or group in (32, 64)):
            return None
        dq = None
    else:                             # prefill: packed GEMM with the silu epilogue
        if m < 1024 or n % 256 != 0 or k % 32 != 0:
            return None
        dq = _dq_scratch(m, n, k, input.device)
        if dq is None:
            return None
    out = torch.empty((m, n // 2), dtype=input.dtype, device=input.device)
    rc = _abi.load().mi355x_awq_gemm_silu_mul(
        _ptr(out), _ptr(input), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(dq),
        dq.numel() if dq is not None else 0, m, n, k, group, input.stride(0), _dt(input), _stream())
    _abi.check(rc, "awq_gemm_silu_mul")
    return out


def rotary_reshape_and_cache(positions: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                             key_cache: torch.Tensor, value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                             cos_sin_cache: torch.Tensor) -> bool:
    """MI355X-side prefill fusion: rotary_embedding (NeoX, rot_dim == head_size) on the key rows + reshape_and_cache
    in one launch; the rotated keys go only to the cache (`key` [T, KVH, D] is not modified).  False: not
    applicable, nothing launched."""
    _dev(positions, key, value, key_cache, value_cache, slot_mapping, cos_sin_cache)
    if key.dtype not in (torch.bfloat16, torch.float16) or key_cache.dtype != key.dtype or key_cache.dim() != 5 \
            or cos_sin_cache.dtype != key.dtype or cos_sin_cache.size(-1) != key.size(2) \
            or not cos_sin_cache.is_contiguous():
        return False
    if positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64:
        raise RuntimeError("rotary_reshape_and_cache: positions and slot_mapping must be int64")
    rc = _abi.load().mi355x_rotary_reshape_and_cache(
        _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping), _ptr(positions),
        _ptr(cos_sin_cache), slot_mapping.size(0), key.stride(0), value.stride(0), key.size(1), key.size(2),
        key_cache.size(3), key_cache.size(4), _dt(key), _stream())
    if rc == 1:
        return False
    _abi.check(rc, "rotary_reshape_and_cache")
    return True


def paged_prefill_attention_image(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                                  num_kv_heads: int, scale: float, block_tables: torch.Tensor,
                                  seq_lens: torch.Tensor, cu_seqlens_q: torch.Tensor, max_query_len: int,
                                  block_size: int, kv_cache_dtype: str = "auto",
                                  k_scale: Optional[torch.Tensor] = None,
                                  v_scale: Optional[torch.Tensor] = None,
                                  positions: Optional[torch.Tensor] = None,
                                  cos_sin_cache: Optional[torch.Tensor] = None) -> Optional["PackedOperand"]:
    """MI355X-side prefill fusion: paged_prefill_attention whose output [tokens, heads * head_size] is written as
    the activation operand image of the GEMM that consumes it (o_proj).  None: not applicable to these shapes.
    With positions + cos_sin_cache the query rows are taken UN-rotated and the NeoX rotary is applied while they
    are loaded (the caller rotates the key rows only)."""
    _dev(query, key_cache, value_cache, block_tables, seq_lens, cu_seqlens_q, positions, cos_sin_cache)
    if positions is not None:
        if positions.dtype != torch.int64 or cos_sin_cache is None or cos_sin_cache.dtype != query.dtype \
                or cos_sin_cache.size(-1) != query.size(-1) or not cos_sin_cache.is_contiguous():
            return None
    if query.dtype not in (torch.bfloat16, torch.float16) or query.dim() != 3:
        return None
    kvd, ks, vs = _kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale)
    t, h, d = query.shape
    mt = (t + 15) // 16
    image = torch.empty(mt * 16 * h * d, dtype=query.dtype, device=query.device)
    if t % 16:
        image[(mt - 1) * 16 * h * d:].zero_()        # rows past the last token of the last row tile
    rc = _abi.load().mi355x_paged_prefill_attention_image(
        _ptr(image), _ptr(query), _ptr(key_cache), _ptr(value_cache), seq_lens.size(0), h, num_kv_heads, d,
        block_size, float(scale), _ptr(block_tables), _ptr(seq_lens), _ptr(cu_seqlens_q), max_query_len,
        block_tables.size(1), query.stride(0), key_cache.stride(0), key_cache.stride(1), _dt(query), kvd, ks, vs,
        _ptr(positions) if positions is not None else None,
        _ptr(cos_sin_cache) if positions is not None else None, _stream())
    if rc == 1:
        return None
    _abi.check(rc, "paged_prefill_attention_image")
    return PackedOperand(image, t, h * d)


def _norm_image(name: str, x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                epsilon: float) -> Optional["PackedOperand"]:
    _dev(x, residual, weight)
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.bfloat16, torch.float16):
        return None
    m, hidden = x.shape
    if residual is not None and (not residual.is_contiguous() or residual.shape != x.shape):
        raise RuntimeError(f"{name}: residual must be contiguous and shaped like the input")
    image = torch.empty(((m + 15) // 16) * 16 * hidden, dtype=x.dtype, device=x.device)
    lib = _abi.load()
    if residual is None:
        rc = lib.mi355x_rms_norm_image(_ptr(image), _ptr(x), _ptr(weight), float(epsilon), m, hidden,
                                       x.stride(0), _dt(x), _stream())
    else:
        rc = lib.mi355x_fused_add_rms_norm_image(_ptr(image), _ptr(x), _ptr(residual), _ptr(weight),
                                                 float(epsilon), m, hidden, x.stride(0), _dt(x), _stream())
    if rc == 1:
        return None
    _abi.check(rc, name)
    return PackedOperand(image, m, hidden)


def rms_norm_image(x: torch.Tensor, weight: torch.Tensor, epsilon: float) -> Optional["PackedOperand"]:
    """MI355X-side prefill fusion: rms_norm(x) written directly as the prefill GEMM's activation operand
    image (None: not applicable to these shapes; bit-identical to rms_norm + the GEMM's own re-tiling)."""
    return _norm_image("rms_norm_image", x, None, weight, epsilon)


def fused_add_rms_norm_image(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                             epsilon: float) -> Optional["PackedOperand"]:
    """Like rms_norm_image for fused_add_rms_norm: residual += x in place (rounded), x itself is not modified."""
    return _norm_image("fused_add_rms_norm_image", x, residual, weight, epsilon)


def paged_attention_fused_qkv(out: torch.Tensor, exp_sums: Optional[torch.Tensor],
                              max_logits: Optional[torch.Tensor], tmp_out: Optional[torch.Tensor],
                              qkv: torch.Tensor, slabs: Optional[torch.Tensor], sk: int,
                              positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                              slot_mapping: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                              num_heads: int, num_kv_heads: int, scale: float, block_tables: torch.Tensor,
                              seq_lens: torch.Tensor, block_size: int, max_seq_len: int,
                              partitioned: bool, partition_size: int = PARTITION_SIZE,
                              slab_scales=None, quant_out=None) -> bool:
    """MI355X-side decode fusion: qkv_rope_cache folded into the paged-attention launch that follows it
    (include/mi355x_hotpath.h).  Returns False when the fused form does not apply to these shapes — the
    caller then runs qkv_rope_cache + paged_attention_v1 / _v2.  `out` [n, H, D]; the qkv buffer is not updated."""
    _dev(out, qkv, positions, cos_sin_cache, slot_mapping, key_cache, value_cache, block_tables, seq_lens)
    if qkv.dim() != 2 or qkv.stride(1) != 1 or key_cache.dim() != 5 or value_cache.dim() != 4:
        raise RuntimeError("paged_attention_fused_qkv: qkv [tokens, width], caches in the x-split layout")
    if (positions is None) != (cos_sin_cache is None):
        raise RuntimeError("paged_attention_fused_qkv: positions and cos_sin_cache go together (both None: no rotary)")
    if (positions is not None and positions.dtype != torch.int64) or slot_mapping.dtype != torch.int64:
        raise RuntimeError("paged_attention_fused_qkv: positions and slot_mapping must be int64")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention_fused_qkv: block_tables and seq_lens must be int32")
    head_size = key_cache.size(2) * key_cache.size(4)
    if qkv.dtype not in (torch.bfloat16, torch.float16) or key_cache.dtype != qkv.dtype:
        return False
    if cos_sin_cache is not None and (cos_sin_cache.dtype != qkv.dtype or cos_sin_cache.size(-1) != head_size):
        return False
    n = qkv.size(0)
    if not out.is_contiguous() or out.numel() != n * num_heads * head_size:
        raise RuntimeError("paged_attention_fused_qkv: out must be contiguous [tokens, heads, head_size]")
    if sk > 0:
        _dev(slabs)
        if slabs.dtype != torch.float32 or slabs.numel() < sk * n * qkv.size(1):
            raise RuntimeError("paged_attention_fused_qkv: slabs must be float32 [sk, tokens, width]")
    if partitioned:
        _dev(exp_sums, max_logits, tmp_out)
    if slab_scales is not None or quant_out is not None:
        # fp8 model (mi355x_paged_attention_fused_qkv_w8): `slab_scales` = (a_scales [n] or [1], b_scales [width] or
        # [1]) of the fp8 qkv GEMM whose slabs these are; `quant_out` = (fp8 [n, num_heads * head_size], float [n, 1]):
        # the reduce launch of the partitioned form quantises the attention output per token, `out` stays unwritten
        a_s, b_s = slab_scales if slab_scales is not None else (None, None)
        q8, qs = quant_out if quant_out is not None else (None, None)
        _dev(a_s, b_s, q8, qs)
        if quant_out is not None and (q8.dtype != torch.float8_e4m3fn or qs.dtype != torch.float32
                                      or q8.numel() != n * num_heads * head_size or qs.numel() != n
                                      or not q8.is_contiguous()):
            raise RuntimeError("paged_attention_fused_qkv: quant_out = (float8_e4m3fn [n, heads * d], float32 [n, 1])")
        if slab_scales is not None and (a_s.dtype != torch.float32 or b_s.dtype != torch.float32
                                        or not (a_s.is_contiguous() and b_s.is_contiguous())):
            raise RuntimeError("paged_attention_fused_qkv: slab scales must be contiguous float32")
        rc = _abi.load().mi355x_paged_attention_fused_qkv_w8(
            _ptr(out), _ptr(exp_sums) if partitioned else None, _ptr(max_logits) if partitioned else None,
            _ptr(tmp_out) if partitioned else None, _ptr(qkv), qkv.stride(0), _ptr(slabs) if sk > 0 else None,
            int(sk), _ptr(positions), _ptr(cos_sin_cache), _ptr(slot_mapping), _ptr(key_cache), _ptr(value_cache),
            n, num_heads, num_kv_heads, head_size, block_size, key_cache.size(4), float(scale), _ptr(block_tables),
            _ptr(seq_lens), block_tables.stride(0), max_seq_len, key_cache.stride(0), key_cache.stride(1),
            int(partition_size) if partitioned else 0, _dt(qkv), _ptr(a_s), a_s.numel() if a_s is not None else 0,
            _ptr(b_s), b_s.numel() if b_s is not None else 0, _ptr(q8), _ptr(qs), _stream())
        if rc == 1:
            return False
        _abi.check(rc, "paged_attention_fused_qkv_w8")
        return True
    rc = _abi.load().mi355x_paged_attention_fused_qkv(
        _ptr(out), _ptr(exp_sums) if partitioned else None, _ptr(max_logits) if partitioned else None,
        _ptr(tmp_out) if partitioned else None, _ptr(qkv), qkv.stride(0), _ptr(slabs) if sk > 0 else None,
        int(sk), _ptr(positions), _ptr(cos_sin_cache), _ptr(slot_mapping), _ptr(key_cache), _ptr(value_cache),
        n, num_heads, num_kv_heads, head_size, block_size, key_cache.size(4), float(scale), _ptr(block_tables),
        _ptr(seq_lens), block_tables.stride(0), max_seq_len, key_cache.stride(0), key_cache.stride(1),
        int(partition_size) if partitioned else 0, _dt(qkv), _stream())
    if rc == 1:
        return False
    _abi.check(rc, "paged_attention_fused_qkv")
    return True


def greedy_advance(logits: torch.Tensor, tokens: torch.Tensor, positions: torch.Tensor,
                   seq_lens: torch.Tensor, slot_mapping: torch.Tensor, block_tables: torch.Tensor,
                   block_size: int) -> None:
    """MI355X-side decode fusion (no reference op: upstream's sampler / model runner use torch ops): greedy
    token of every row of `logits` [n, vocab] into `tokens`, positions and seq_lens += 1, slot_mapping =
    the slot of the new position (block_tables [n, max_blocks] int32, row r = sequence r)."""
    _dev(logits, tokens, positions, seq_lens, slot_mapping, block_tables)
    n = logits.size(0)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("greedy_advance: logits must be [n, vocab] with unit inner stride")
    if tokens.dtype != torch.int64 or positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64 \
            or seq_lens.dtype != torch.int32 or block_tables.dtype != torch.int32:
        raise RuntimeError("greedy_advance: tokens / positions / slot_mapping int64, seq_lens / block_tables int32")
    for t in (tokens, positions, seq_lens, slot_mapping):
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError("greedy_advance: per-sequence tensors must be contiguous with one element per row")
    if block_tables.dim() != 2 or block_tables.size(0) < n or block_tables.stride(1) != 1:
        raise RuntimeError("greedy_advance: block_tables must be [>= n, max_blocks]")
    rc = _abi.load().mi355x_greedy_advance(
        _ptr(logits), logits.stride(0), n, logits.size(1), _ptr(tokens), _ptr(positions), _ptr(seq_lens),
        _ptr(slot_mapping), _ptr(block_tables), block_tables.stride(0), int(block_size), _dt(logits), _stream())
    _abi.check(rc, "greedy_advance")


class PackedOperand:
    """An activation matrix [m, k] held as the MFMA operand image the prefill GEMM reads (see
    mi355x_awq_gemm_silu_mul_packed in include/mi355x_hotpath.h): produced by
    awq_gemm_silu_mul_packed, consumed by awq_gemm_packed_a."""

    def __init__(self, data: torch.Tensor, m: int, k: int):
        self.data, self.m, self.k = data, m, k

    @property
    def shape(self):
        return (self.m, self.k)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device


def awq_gemm_silu_mul_packed(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                             scales: torch.Tensor) -> Optional[PackedOperand]:
    """Prefill (M >= 1024) gate_up GEMM + silu_and_mul whose result is written directly as the
    operand image of the following down_proj GEMM (no row-major act, no re-tiling launch).
    Returns None when the path does not apply; awq_gemm_packed_a(result, ...) consumes it."""
    _dev(input, qweight, qzeros, scales)
    if input.dim() != 2 or input.stride(1) != 1:
        raise RuntimeError("awq_gemm_silu_mul_packed: input must be [M, K] with unit inner stride")
    m, k = input.shape
    n = qweight.size(0)
    if input.dtype not in (torch.bfloat16, torch.float16) or m < 1024 or n % 256 != 0 or k % 32 != 0:
        return None
    group = k // scales.size(0)
    dq = _dq_scratch(m, n, k, input.device)
    m_pad = (m + 15) // 16 * 16
    out = torch.empty(m_pad * (n // 2), dtype=input.dtype, device=input.device)
    rc = _abi.load().mi355x_awq_gemm_silu_mul_packed(
        _ptr(out), _ptr(input), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(dq), dq.numel(),
        m, n, k, group, input.stride(0), _dt(input), _stream())
    _abi.check(rc, "awq_gemm_silu_mul_packed")
    return PackedOperand(out, m, n // 2)


def awq_gemm_packed_a(a: PackedOperand, qweight: torch.Tensor, qzeros: torch.Tensor,
                      scales: torch.Tensor) -> torch.Tensor:
    """awq_gemm whose activations already are an operand image (M >= 1024)."""
    _dev(a.data, qweight, qzeros, scales)
    m, k = a.m, a.k
    n = qweight.size(0)
    if qweight.size(1) * 8 != k:
        raise RuntimeError(f"awq_gemm_packed_a: operand has k = {k}, weights have k = {qweight.size(1) * 8}")
    group = k // scales.size(0)
    dq = _dq_scratch(m, n, k, a.data.device)
    if dq is None:
        raise RuntimeError("awq_gemm_packed_a: needs M >= 1024")
    out = torch.empty((m, n), dtype=a.data.dtype, device=a.data.device)
    rc = _abi.load().mi355x_awq_gemm_packed_a(
        _ptr(out), _ptr(a.data), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(dq), dq.numel(),
        m, n, k, group, _dt(a.data), _stream())
    _abi.check(rc, "awq_gemm_packed_a")
    return out


# ---- weight-load-time image of a w4a16 layer for the prefill GEMM (MI355X-side, no reference op) ----
PREPACKED_SILU, PREPACKED_OUT_IMAGE, PREPACKED_A_IMAGE = 1, 2, 4


def w4a16_prepack(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                  gptq_zeros: bool = False) -> torch.Tensor:
    """The operand image the prefill GEMM multiplies (T(fma(q, s, -z s)) in MFMA piece order), computed
    ONCE instead of on every awq_gemm / gptq_gemm call with m >= 1024.  qweight: exllama layout (declared
    [N, K/8] from awq_to_gptq_4bit, or [K/8, N] from gptq_shuffle).  Returns a flat tensor of N * K
    elements of scales.dtype (N * K * 2 bytes)."""
    _dev(qweight, qzeros, scales)
    n = scales.size(1)
    k = qweight.numel() * 8 // n
    group = k // scales.size(0)
    image = torch.empty(n * k, dtype=scales.dtype, device=scales.device)
    rc = _abi.load().mi355x_w4a16_prepack(_ptr(image), _ptr(qweight), _ptr(scales), _ptr(qzeros), n, k,
                                          group, 1 if gptq_zeros else 0, _dt(scales), _stream())
    _abi.check(rc, "w4a16_prepack")
    return image


W4_PREPACKED_MIN_M = 384     # smallest m mi355x_w4a16_gemm_prepacked takes (csrc/w4a16.cuh kW4PrepackedMinM)


def w4a16_gemm_prepacked(a, image: torch.Tensor, n: int, k: int, silu: bool = False,
                         out_image: bool = False):
    """a [m, k] (row-major tensor or PackedOperand) times a prepacked weight image (m >= W4_PREPACKED_MIN_M).
    silu: gate_up projection, returns silu_and_mul(a . W) [m, n/2]; out_image (with silu): returns it
    as a PackedOperand for the next GEMM.  Bit-identical to awq_gemm (+ silu_and_mul) at the same m."""
    packed_in = isinstance(a, PackedOperand)
    data = a.data if packed_in else a
    _dev(data, image)
    m = a.m if packed_in else a.shape[0]
    if (a.k if packed_in else a.shape[1]) != k:
        raise RuntimeError("w4a16_gemm_prepacked: operand k does not match the image")
    mode = (PREPACKED_SILU if silu else 0) | (PREPACKED_OUT_IMAGE if out_image else 0) | \
        (PREPACKED_A_IMAGE if packed_in else 0)
    m_pad = (m + 15) // 16 * 16
    ws = None
    # partial tiles of a K split behind the activation image (shapes with few 256 x 256 tiles)
    split_bytes = 4 * int(_abi.load().mi355x_w4a16_prepacked_split_elems(m, n, k))
    if not packed_in:
        if a.dim() != 2 or a.stride(1) != 1:
            raise RuntimeError("w4a16_gemm_prepacked: input must be [M, K] with unit inner stride")
        ws = _a_scratch(m_pad * k * 2 + split_bytes, data.device)
    elif split_bytes:
        ws = _a_scratch(split_bytes, data.device)
    if out_image:
        out = torch.empty(m_pad * (n // 2), dtype=data.dtype, device=data.device)
    else:
        out = torch.empty((m, n // 2 if silu else n), dtype=data.dtype, device=data.device)
    rc = _abi.load().mi355x_w4a16_gemm_prepacked(
        _ptr(out), _ptr(data), _ptr(image), _ptr(ws), ws.numel() if ws is not None else 0, m, n, k,
        k if packed_in else a.stride(0), mode, _dt(data), _stream())
    _abi.check(rc, "w4a16_gemm_prepacked")
    return PackedOperand(out, m, n // 2) if out_image else out


_A_SCRATCH: dict = {}


def _a_scratch(nbytes: int, device) -> torch.Tensor:
    return _grow(_A_SCRATCH, device, nbytes, torch.uint8)


def awq_gemm_deferred(input: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor,
                      scales: torch.Tensor, temp_space: torch.Tensor):
    """awq_gemm that may leave its split-K reduction to the consumer: returns (out, sk).  sk >= 2:
    the result is still `sk` fp32 slabs [sk, M, N] at the start of temp_space and `out` is
    unwritten (pass both to fused_add_rms_norm_slabs); sk == 0: `out` holds the result."""
    _dev(input, qweight, qzeros, scales, temp_space)
    if input.dim() != 2 or input.stride(1) != 1:
        raise RuntimeError("awq_gemm_deferred: input must be [M, K] with unit inner stride")
    m, k = input.shape
    n = qweight.size(0)
    group = k // scales.size(0)
    out = torch.empty((m, n), dtype=input.dtype, device=input.device)
    dq = _dq_scratch(m, n, k, input.device)
    sk = ctypes.c_int(0)
    rc = _abi.load().mi355x_awq_gemm_deferred(
        _ptr(out), _ptr(input), _ptr(qweight), _ptr(scales), _ptr(qzeros), _ptr(temp_space),
        temp_space.numel(), _ptr(dq), dq.numel() if dq is not None else 0, m, n, k, group,
        input.stride(0), _dt(input), ctypes.byref(sk), _stream())
    _abi.check(rc, "awq_gemm_deferred")
    return out, int(sk.value)


def gptq_shuffle(q_weight: torch.Tensor, q_perm: torch.Tensor, bit: int) -> None:
    """ref: vllm_metax/_custom_ops.py:61-62; csrc/quantization/gptq/q_gemm.cu:2415-2423."""
    _dev(q_weight)
    perm = None
    if q_perm is not None and q_perm.device.type != "meta" and q_perm.numel() > 0:
        _dev(q_perm)
        perm = q_perm.to(torch.int32).contiguous()
    k = q_weight.size(0) * 32 // bit
    n = q_weight.size(1)
    scratch = torch.empty_like(q_weight) if perm is not None else None
    rc = _abi.load().mi355x_gptq_shuffle(_ptr(q_weight), _ptr(perm), _ptr(scratch), k, n, bit,
                                         _stream())
    _abi.check(rc, "gptq_shuffle")


def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor,
              b_gptq_scales: torch.Tensor, b_g_idx: torch.Tensor, use_exllama: bool, bit: int,
              group_size: int, perm_space: torch.Tensor, temp_space: torch.Tensor,
              dtype_bf16: bool) -> torch.Tensor:
    """ref: vllm_metax/_custom_ops.py:33-58; csrc/quantization/gptq/q_gemm.cu:2373-2413."""
    _dev(a, b_q_weight, b_gptq_qzeros, b_gptq_scales)
    if dtype_bf16 != (a.dtype == torch.bfloat16):
        raise RuntimeError("gptq_gemm: dtype_bf16 does not match the input dtype")
    if not use_exllama:
        raise RuntimeError("gptq_gemm: only the exllama (shuffled) weight layout is supported")
    if not a.is_contiguous():
        raise RuntimeError("gptq_gemm: a must be contiguous")
    m, k = a.shape
    n = b_q_weight.size(1)
    if group_size == -1:          # per-channel GPTQ: one group spanning K
        group_size = k
    g_idx = None
    if b_g_idx is not None and b_g_idx.device.type != "meta" and b_g_idx.numel() > 0:
        g_idx = b_g_idx.to(torch.int32).contiguous()
    pspace = None
    if g_idx is not None:
        if perm_space is not None and perm_space.is_cuda and perm_space.numel() >= m * k \
                and perm_space.element_size() == 2:
            pspace = perm_space
        else:
            pspace = torch.empty((m, k), dtype=a.dtype, device=a.device)
    ws = temp_space if (temp_space is not None and temp_space.is_cuda
                        and temp_space.dtype == torch.float32 and temp_space.numel() > 0) else None
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    dq = _dq_scratch(m, n, k, a.device)
    rc = _abi.load().mi355x_gptq_gemm(_ptr(out), _ptr(a), _ptr(b_q_weight), _ptr(b_gptq_qzeros),
                                      _ptr(b_gptq_scales), _ptr(g_idx), _ptr(pspace), _ptr(ws),
                                      ws.numel() if ws is not None else 0, _ptr(dq),
                                      dq.numel() if dq is not None else 0, m, n, k, bit,
                                      group_size, _dt(a), _stream())
    _abi.check(rc, "gptq_gemm")
    return out


# -------------------------------------------------------------------------- fp8 GEMM
def scaled_int8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None,
                      azp: Optional[torch.Tensor] = None, symmetric: bool = True):
    """ref: vllm._custom_ops.scaled_int8_quant -> torch.ops._C.static_scaled_int8_quant /
    dynamic_scaled_int8_quant (csrc/quantization/compressed_tensors/int8_quant_kernels.cu).
    Returns (int8 tensor, scales, None).  static when `scale` is given (one fp32 value), else
    dynamic per token.  Symmetric only."""
    if azp is not None or not symmetric:
        raise NotImplementedError("scaled_int8_quant: the asymmetric (azp) variants are not implemented")
    _dev(input)
    if input.stride(-1) != 1:
        raise RuntimeError("scaled_int8_quant: input must have unit inner stride")
    hidden = input.size(-1)
    tokens = input.numel() // hidden if hidden else 0
    in_stride = input.stride(-2) if input.dim() >= 2 else hidden
    out = torch.empty(input.shape, dtype=torch.int8, device=input.device)
    if scale is not None:
        _dev(scale)
        if scale.numel() != 1 or scale.dtype != torch.float32:
            raise RuntimeError("scaled_int8_quant: static scale must be one float32 value")
        rc = _abi.load().mi355x_static_scaled_int8_quant(_ptr(out), _ptr(input), _ptr(scale), tokens,
                                                         hidden, in_stride, _dt(input), _stream())
        _abi.check(rc, "static_scaled_int8_quant")
        return out, scale, None
    scales = torch.empty((tokens, 1), dtype=torch.float32, device=input.device)
    rc = _abi.load().mi355x_dynamic_scaled_int8_quant(_ptr(out), _ptr(input), _ptr(scales), tokens,
                                                      hidden, in_stride, _dt(input), _stream())
    _abi.check(rc, "dynamic_scaled_int8_quant")
    return out, scales, None


_F32_SCRATCH = {}


def _scratch_f32(elems: int, device) -> torch.Tensor:
    """Per-device reusable fp32 scratch (stream-ordered use), grown on demand."""
    return _grow(_F32_SCRATCH, device, elems, torch.float32)


def _scaled_mm_in_place(a: torch.Tensor, ldb: int, b_ptr: int):
    """Which operands of the m > 320 8-bit GEMM are read in place, without an operand image (fp8_gemm.hip run_fp8:
    fp8, k % 128 == 0, rows 16-byte aligned; MI355X_F8_ROWMAJOR bit 0 activations, bit 1 weights, for A/B runs)."""
    bits = int(os.environ.get("MI355X_F8_ROWMAJOR", "3"))
    wide = a.dtype == torch.float8_e4m3fn and a.size(1) % 128 == 0
    a_in_place = wide and bool(bits & 1) and a.stride(0) % 16 == 0 and a.data_ptr() % 16 == 0
    b_in_place = a_in_place and bool(bits & 2) and ldb % 16 == 0 and b_ptr % 16 == 0
    return a_in_place, b_in_place


def scaled_mm_prepack(b: torch.Tensor, force: bool = False) -> Optional[torch.Tensor]:
    """Load-time operand image of an fp8 / int8 weight `b` [k, n] (column-major, as cutlass_scaled_mm takes it) for
    the packed path of the 8-bit GEMM (m > 320): uint8 [n * k], or None when the shape has none (n % 64, k % 64) or
    needs none — fp8 weights with k % 128 == 0 and 16-byte aligned rows are read in place at the image's speed
    (profiles/r03_fp8_operands_in_place.txt), so no second copy of them is kept unless `force`."""
    _dev(b)
    if b.dim() != 2 or b.stride(0) != 1 or b.dtype not in (torch.float8_e4m3fn, torch.int8):
        raise RuntimeError("scaled_mm_prepack: b must be column-major float8_e4m3fn or int8 [k, n]")
    k, n = b.shape
    if not force and b.dtype == torch.float8_e4m3fn and k % 128 == 0 and b.stride(1) % 16 == 0 \
            and b.data_ptr() % 16 == 0 and int(os.environ.get("MI355X_F8_ROWMAJOR", "3")) == 3:
        return None
    image = torch.empty(n * k, dtype=torch.uint8, device=b.device)
    rc = _abi.load().mi355x_scaled_mm_prepack(_ptr(image), _ptr(b), n, k, b.stride(1), _stream())
    if rc == 1:
        return None
    _abi.check(rc, "scaled_mm_prepack")
    return image


def scaled_mm_prepacked(out: torch.Tensor, a: torch.Tensor, b_image: torch.Tensor, n: int, a_scales: torch.Tensor,
                        b_scales: torch.Tensor, bias: Optional[torch.Tensor] = None) -> None:
    """cutlass_scaled_mm(out, a, b, ...) with b given as scaled_mm_prepack(b) (m > 320): bit-identical, without the
    per-call re-tiling of the weights."""
    _dev(out, a, b_image, a_scales, b_scales, bias)
    m, k = a.shape
    if a.stride(1) != 1 or out.stride(1) != 1 or out.shape != (m, n) or b_image.numel() != n * k \
            or a.dtype not in (torch.float8_e4m3fn, torch.int8):
        raise RuntimeError("scaled_mm_prepacked: shape / layout mismatch")
    if a_scales.numel() not in (1, m) or b_scales.numel() not in (1, n) or a_scales.dtype != torch.float32 \
            or b_scales.dtype != torch.float32 or not (a_scales.is_contiguous() and b_scales.is_contiguous()):
        raise RuntimeError("scaled_mm_prepacked: float32 contiguous scales, per-tensor or per-row / per-column")
    if bias is not None and (bias.numel() != n or not bias.is_contiguous() or bias.dtype != out.dtype):
        raise RuntimeError("scaled_mm_prepacked: bad bias")
    elems = (0 if _scaled_mm_in_place(a, 0, 0)[0] else (m + 15) // 16 * 16 * k // 4) \
        + int(_abi.load().mi355x_scaled_mm_split_elems(m, n, k))
    ws = _scratch_f32(elems, a.device) if elems else None
    rc = _abi.load().mi355x_scaled_mm_prepacked(
        _ptr(out), _ptr(a), _ptr(b_image), _ptr(a_scales), a_scales.numel(), _ptr(b_scales), b_scales.numel(),
        _ptr(bias), _ptr(ws), ws.numel() if ws is not None else 0, m, n, k, a.stride(0), out.stride(0), _dt(out),
        1 if a.dtype == torch.int8 else 0, _stream())
    _abi.check(rc, "scaled_mm_prepacked")


def scaled_mm_fp8_deferred(out: torch.Tensor, a: torch.Tensor, b: torch.Tensor, a_scales: torch.Tensor,
                           b_scales: torch.Tensor, workspace: torch.Tensor) -> int:
    """MI355X-side decode fusion: cutlass_scaled_mm (fp8, no bias) that may leave its K split as float32 partial slabs
    `workspace[:sk * m * n]` ([sk, m, n]) for the consumer — returns sk (> 1), or 0 when `out` is final.  The
    consumers (paged_attention_fused_qkv slab_scales=, silu_and_mul_per_token_quant_slabs,
    rms_norm_dynamic_per_token_quant_slabs) produce the bits the GEMM's own finish launch would have stored."""
    _dev(out, a, b, a_scales, b_scales, workspace)
    if a.dim() != 2 or b.dim() != 2 or out.dim() != 2 or out.size(0) != a.size(0) or a.size(1) != b.size(0) \
            or b.size(1) != out.size(1):
        raise RuntimeError("scaled_mm_fp8_deferred: shape mismatch")
    if a.stride(1) != 1 or out.stride(1) != 1 or b.stride(0) != 1:
        raise RuntimeError("scaled_mm_fp8_deferred: a / out row-major, b column-major")
    if a.dtype != torch.float8_e4m3fn or b.dtype != torch.float8_e4m3fn:
        raise RuntimeError("scaled_mm_fp8_deferred: a and b must be float8_e4m3fn")
    m, k = a.shape
    n = b.size(1)
    if a_scales.numel() not in (1, m) or b_scales.numel() not in (1, n) or a_scales.dtype != torch.float32 \
            or b_scales.dtype != torch.float32 or not (a_scales.is_contiguous() and b_scales.is_contiguous()):
        raise RuntimeError("scaled_mm_fp8_deferred: float32 contiguous scales, per-tensor or per-row / per-column")
    if workspace.dtype != torch.float32:
        raise RuntimeError("scaled_mm_fp8_deferred: workspace must be float32")
    sk = ctypes.c_int(0)
    rc = _abi.load().mi355x_scaled_mm_fp8_deferred(
        _ptr(out), _ptr(a), _ptr(b), _ptr(a_scales), a_scales.numel(), _ptr(b_scales), b_scales.numel(),
        _ptr(workspace), workspace.numel(), m, n, k, a.stride(0), b.stride(1), out.stride(0), _dt(out),
        ctypes.byref(sk), _stream())
    _abi.check(rc, "scaled_mm_fp8_deferred")
    return sk.value


def cutlass_scaled_mm_supports_fp8(cuda_device_capability: int) -> bool:
    """The reference returns False (scaled_mm_entry.cu:22-24); gfx950 has OCP-fp8 MFMA."""
    return True


def cutlass_scaled_mm(out: torch.Tensor, a: torch.Tensor, b: torch.Tensor,
                      a_scales: torch.Tensor, b_scales: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> None:
    """Schema csrc/torch_bindings.cpp:251-256; checks follow scaled_mm_entry.cu:84-140."""
    _dev(out, a, b, a_scales, b_scales, bias)
    if a.dim() != 2 or b.dim() != 2 or out.dim() != 2:
        raise RuntimeError("cutlass_scaled_mm: a, b, out must be 2-D")
    if out.size(0) != a.size(0) or a.size(1) != b.size(0) or b.size(1) != out.size(1):
        raise RuntimeError("cutlass_scaled_mm: shape mismatch")
    if a.stride(1) != 1 or out.stride(1) != 1:
        raise RuntimeError("cutlass_scaled_mm: a and out must be row-major")
    if b.stride(0) != 1:
        raise RuntimeError("cutlass_scaled_mm: b must be column-major")
    if out.stride(0) % 16 or b.stride(1) % 16:
        raise RuntimeError("cutlass_scaled_mm: 16-byte alignment required")
    if a.dtype != b.dtype or a.dtype not in (torch.float8_e4m3fn, torch.int8):
        raise RuntimeError("cutlass_scaled_mm: a and b must both be float8_e4m3fn or both int8")
    m, k = a.shape
    n = b.size(1)
    if a_scales.numel() not in (1, m) or b_scales.numel() not in (1, n):
        raise RuntimeError("cutlass_scaled_mm: scales must be per-tensor or per-row/column")
    if a_scales.dtype != torch.float32 or b_scales.dtype != torch.float32:
        raise RuntimeError("cutlass_scaled_mm: scales must be float32")
    if not (a_scales.is_contiguous() and b_scales.is_contiguous()):
        raise RuntimeError("cutlass_scaled_mm: scales must be contiguous")
    if bias is not None and (bias.numel() != n or not bias.is_contiguous() or bias.dtype != out.dtype):
        raise RuntimeError("cutlass_scaled_mm: bad bias")
    # small-M (decode) shapes split K across up to 8 workgroups, one fp32 / int32 partial slab [m, n] each
    if m <= 320:       # (64 < m <= 320: passes of 64 rows through the decode kernel, one pass's slabs at a time)
        ws = torch.empty((8, min(m, 64), n), dtype=torch.float32, device=a.device)
    elif m >= int(os.environ.get("MI355X_F8_PACKED_MIN_M", "321")) and k % 64 == 0:
        # prefill: scratch for the re-tiled operands, (roundup(m,16) + roundup(n,16)) * k bytes — none for the operands
        # the GEMM reads in place
        a_in_place, b_in_place = _scaled_mm_in_place(a, b.stride(1), b.data_ptr())
        need = (0 if a_in_place else (m + 15) // 16 * 16 * k) + (0 if b_in_place else (n + 15) // 16 * 16 * k)
        # + the partial tiles of a K split (shapes with few 256 x 256 tiles)
        elems = (need + 15) // 16 * 4 + int(_abi.load().mi355x_scaled_mm_split_elems(m, n, k))
        ws = _scratch_f32(elems, a.device) if elems else None
    else:
        ws = None
    fn = _abi.load().mi355x_scaled_mm_int8 if a.dtype == torch.int8 else _abi.load().mi355x_scaled_mm_fp8
    rc = fn(
        _ptr(out), _ptr(a), _ptr(b), _ptr(a_scales), a_scales.numel(), _ptr(b_scales),
        b_scales.numel(), _ptr(bias), _ptr(ws), ws.numel() if ws is not None else 0, m, n, k,
        a.stride(0), b.stride(1), out.stride(0), _dt(out), _stream())
    _abi.check(rc, "cutlass_scaled_mm")
