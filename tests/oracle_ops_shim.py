"""TEST INFRASTRUCTURE: the op surface of vllm_metax_amd._custom_ops that harness.py uses, implemented
with the CPU oracle (oracle/ref_ops.py).  It lets CPU tests execute harness.HotPathModel itself —
its sharding, collectives, slot arithmetic and op sequence — without a GPU (tests/test_cpu_tp_gloo.py
installs it over `harness.ops` / `backend.ops`).  Never imported by the product package.
"""
from typing import Optional

import torch

from oracle import ref_ops as R

PARTITION_SIZE = 512
W4_PREPACKED_MIN_M = 384     # (_custom_ops.W4_PREPACKED_MIN_M; the shim has no weight images: QLinear.prepack is off on CPU)


class PackedOperand:          # the prefill-only operand image type of the real module (never produced here)
    pass


def rms_norm(out, x, w, eps):
    out.copy_(R.rms_norm(x, w, eps))


def fused_add_rms_norm(x, residual, w, eps):
    o, z = R.fused_add_rms_norm(x, residual, w, eps)
    x.copy_(o)
    residual.copy_(z)


def fused_add_rms_norm_slabs(x, residual, w, slabs, sk, eps):
    assert sk == 0
    fused_add_rms_norm(x, residual, w, eps)


def rotary_embedding(positions, q, k, head_size, cos_sin, is_neox):
    nq, nk = R.rotary_embedding(positions, q, k, head_size, cos_sin, is_neox)
    q.copy_(nq)
    if k is not None:
        k.copy_(nk)


def reshape_and_cache(key, value, kc, vc, slots, kv_cache_dtype="auto", k_scale=None, v_scale=None):
    if kv_cache_dtype == "auto":
        R.reshape_and_cache(key, value, kc, vc, slots)
    else:
        R.reshape_and_cache_fp8(key, value, kc, vc, slots, float(k_scale), float(v_scale))


def qkv_rope_cache(qkv, slabs, sk, positions, cos_sin, kc, vc, slots, num_heads, num_kv_heads, head_size):
    assert sk == 0
    qs, ks = num_heads * head_size, num_kv_heads * head_size
    q, k, v = qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]
    rotary_embedding(positions, q, k, head_size, cos_sin, True)
    R.reshape_and_cache(k.reshape(-1, num_kv_heads, head_size), v.reshape(-1, num_kv_heads, head_size), kc, vc, slots)


def _caches(kc, vc, kv_cache_dtype, k_scale, v_scale, dtype):
    if kv_cache_dtype == "auto":
        return kc, vc
    return R.fp8_dequant(kc, float(k_scale), dtype), R.fp8_dequant(vc, float(v_scale), dtype)


def paged_attention_v1(out, q, kc, vc, kvh, scale, bt, sl, bs, max_len, alibi, kv_cache_dtype="auto",
                       k_scale=None, v_scale=None, *a):
    kc, vc = _caches(kc, vc, kv_cache_dtype, k_scale, v_scale, q.dtype)
    out.copy_(R.paged_attention_v1(q, kc, vc, kvh, scale, bt, sl, alibi))


def paged_attention_v2(out, es, ml, tmp, q, kc, vc, kvh, scale, bt, sl, bs, max_len, alibi,
                       kv_cache_dtype="auto", k_scale=None, v_scale=None, *a, partition_size=PARTITION_SIZE):
    kc, vc = _caches(kc, vc, kv_cache_dtype, k_scale, v_scale, q.dtype)
    out.copy_(R.paged_attention_v2(q, kc, vc, kvh, scale, bt, sl, max_len, alibi, partition_size)[0])


def paged_attention_v1_max_seq_len(num_seqs, num_heads, num_kv_heads, head_size, block_size, dtype):
    return 1 << 20


def paged_prefill_attention(out, q, kc, vc, kvh, scale, bt, sl, cu, max_q, bs, kv_cache_dtype="auto",
                            k_scale=None, v_scale=None):
    kc, vc = _caches(kc, vc, kv_cache_dtype, k_scale, v_scale, q.dtype)
    out.copy_(R.paged_prefill_attention(q, kc, vc, kvh, scale, bt, sl, cu))


def silu_and_mul(out, x):
    out.copy_(R.silu_and_mul(x))


def awq_to_gptq_4bit(qw):
    return R.awq_to_gptq_4bit(qw)


def awq_gemm(x, qweight, qzeros, scales, split_k_iters, temp_space, dtype_bf16):
    return R.awq_gemm(x, qweight, scales, qzeros)


def awq_gemm_deferred(x, qweight, qzeros, scales, temp_space):
    return awq_gemm(x, qweight, qzeros, scales, 0, None, True), 0


def awq_gemm_silu_mul(x, qweight, qzeros, scales) -> Optional[torch.Tensor]:
    return None                                   # the caller then runs awq_gemm + silu_and_mul


def awq_gemm_silu_mul_packed(x, qweight, qzeros, scales):
    return None


def greedy_advance(logits, tokens, positions, seq_lens, slot_mapping, block_tables, block_size):
    tokens.copy_(logits.float().argmax(dim=-1))
    positions.add_(1)
    seq_lens.add_(1)
    bi = (positions // block_size).clamp(max=block_tables.shape[1] - 1)
    blk = block_tables[torch.arange(positions.numel()), bi].long()
    slot_mapping.copy_(blk * block_size + positions % block_size)


def paged_attention_fused_qkv(*a, **k):
    return False          # the CPU shim has no fused form: the harness falls back to qkv_rope_cache + attention


def rms_norm_image(x, weight, eps):
    return None           # no image form on the CPU shim: the harness runs the row-major op


def fused_add_rms_norm_image(x, residual, weight, eps):
    return None


def paged_prefill_attention_image(*a, **k):
    return None


def rotary_reshape_and_cache(*a, **k):
    return False
