"""CPU oracle for the MI355X hot path — TEST INFRASTRUCTURE ONLY.

A plain torch-CPU / numpy restatement of the reference's algorithms for every op on
the hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package (vllm_metax_amd/) never
does, and fails loudly when its HIP library is missing.

Pinning status ("parity pinned by"): the reference ships no stored known-answer
vectors for any of these kernels and neither its CUDA-dialect csrc nor
its Python (which imports vllm) can run in the build container, so every function
below is a restatement of the cited reference source, cross-checked against the
independent in-file PyTorch references of the reference's own tests where one exists
(cited per function) and frozen into tests/golden/*.npz by tests/golden/make_golden.py.
Ops without such a test-side reference say "parity unpinned" in their docstring.

Rounding points are spelled out with explicit .float() / .to(dtype) so that the oracle
follows the kernels' order of operations, not just the mathematical definition.
All paths are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0
FP8_MIN_SCALE = 1.0 / (448.0 * 512.0)
PARTITION_SIZE = 512


# =============================================================================== cache
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping) -> None:
    """csrc/cache_kernels.cu:203-255 (kernel index math :235-242); slot < 0 skipped
    (:217-220).  Cross-check: tests/kernels/attention/test_cache.py:190-213.
    key/value [T, H, D]; key_cache [nb, H, D/x, bs, x]; value_cache [nb, H, D, bs]."""
    nb, H, Dx, bs, x = key_cache.shape
    T = slot_mapping.shape[0]
    for t in range(T):
        slot = int(slot_mapping[t])
        if slot < 0:
            continue
        blk, off = slot // bs, slot % bs
        key_cache[blk, :, :, off, :] = key[t].reshape(H, Dx, x)
        value_cache[blk, :, :, off] = value[t]


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping) -> None:
    """csrc/cache_kernels.cu:271-344; num_tokens = slot_mapping.size(0) (:459-469).
    key_cache/value_cache indexed [blk, off, head, :] (NHD) — an HND cache is passed
    as a permuted view.  Cross-check: tests/kernels/attention/test_cache.py:324-360."""
    bs = key_cache.shape[1]
    for t in range(slot_mapping.shape[0]):
        slot = int(slot_mapping[t])
        if slot < 0:
            continue
        blk, off = slot // bs, slot % bs
        key_cache[blk, off] = key[t]
        value_cache[blk, off] = value[t]


def copy_blocks(key_caches: Sequence[torch.Tensor], value_caches: Sequence[torch.Tensor],
                block_mapping) -> None:
    """csrc/cache_kernels.cu:65-91,116-163: pairs applied in order, per layer.
    Cross-check: tests/kernels/attention/test_cache.py:106-118."""
    for src, dst in block_mapping.tolist():
        for kc, vc in zip(key_caches, value_caches):
            kc[dst].copy_(kc[src])
            vc[dst].copy_(vc[src])


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping) -> None:
    """csrc/cache_kernels.cu:18-60: dst[d] = src[s] for each (s, d) pair."""
    for s, d in block_mapping.tolist():
        dst[d].copy_(src[s])


# ====================================================================== fp8 (e4m3fn) KV cache
# SURVEY §8f-3.  The reference carries the hook (cache_kernels.cu:245-253 scaled_convert in the
# cache write, :258-269 CopyWithScaleOp, :544-612 convert_fp8, attention_kernels.cuh:266-277 /
# :398-407 on the read side) but its dispatch rejects everything except "auto"
# (quantization/fp8/metax/quant_utils.cuh:29-42), so the arithmetic restated here is upstream
# vLLM's: byte = sat_e4m3(float(x) / scale) (RNE), value = T(float(byte) * scale).
# Layout: x = 16 / sizeof(cache_t) = 16: key_cache [nb, H, D/16, bs, 16] bytes, value_cache
# [nb, H, D, bs] bytes.  parity unpinned by a reference run (rejected there); the read side is
# pinned through the reference's own test procedure (dequantise, then its torch attention
# reference: tests/kernels/attention/test_attention.py:303-328, fixture ref_paged_attention_fp8kv).
# "fp8_e5m2" (upstream vLLM's second 8-bit cache format; csrc/attention/dtype_fp8.cuh:9-13 lists kFp8E5M2 beside
# kFp8E4M3): the same arithmetic with e5m2 bytes — byte = sat_e5m2(float(x) / scale), saturating at 57344, RNE.
BF8 = torch.float8_e5m2
BF8_MAX = 57344.0


def fp8_quant(x: torch.Tensor, scale: float, fmt: str = "e4m3") -> torch.Tensor:
    """scalar_t -> e4m3 (or e5m2) bytes (uint8 tensor)."""
    y = x.float() / np.float32(scale)
    if fmt == "e5m2":
        return y.clamp(-BF8_MAX, BF8_MAX).to(BF8).view(torch.uint8)
    return _to_fp8_sat(y).view(torch.uint8)


def fp8_dequant(b: torch.Tensor, scale: float, dtype, fmt: str = "e4m3") -> torch.Tensor:
    """e4m3 (or e5m2) bytes -> T(float(byte) * scale)."""
    return (b.view(BF8 if fmt == "e5m2" else FP8).float() * np.float32(scale)).to(dtype)


def reshape_and_cache_fp8(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale, fmt="e4m3") -> None:
    """reshape_and_cache with kv_cache_dtype "fp8" / "fp8_e5m2": caches are uint8, x = 16."""
    reshape_and_cache(fp8_quant(key, k_scale, fmt), fp8_quant(value, v_scale, fmt), key_cache, value_cache,
                      slot_mapping)


def reshape_and_cache_flash_fp8(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale) -> None:
    reshape_and_cache_flash(fp8_quant(key, k_scale), fp8_quant(value, v_scale), key_cache, value_cache,
                            slot_mapping)


# =========================================================================== attention
def _gather_kv(key_cache, value_cache, block_table, seq_len, kv_head):
    """K [L, D], V [L, D] of one sequence / kv head from the x-split paged layout."""
    nb, H, Dx, bs, x = key_cache.shape
    nblk = (seq_len + bs - 1) // bs
    blocks = block_table[:nblk].long()
    k = key_cache[blocks, kv_head]                      # [nblk, D/x, bs, x]
    k = k.permute(0, 2, 1, 3).reshape(nblk * bs, Dx * x)[:seq_len]
    v = value_cache[blocks, kv_head]                    # [nblk, D, bs]
    v = v.permute(0, 2, 1).reshape(nblk * bs, -1)[:seq_len]
    return k, v


def _softmax_pv(q, k, v, scale, alibi_bias, dtype):
    """One (sequence-or-partition, head): attention_kernels.cuh:283-295 (fp32 logits,
    alibi), :321-334 (exp, 1/(sum+1e-6)), :387-389 (probs -> scalar_t), :420 (fp32 PV)."""
    logits = scale * (k.float() @ q.float())            # [L]
    if alibi_bias is not None:
        logits = logits + alibi_bias
    m = logits.max()
    e = torch.exp(logits - m)
    s = e.sum()
    p = (e * (1.0 / (s + 1e-6))).to(dtype)              # rounding point
    o = p.float() @ v.float()                           # [D], fp32 accumulate
    return o, m, s


def paged_attention_v1(query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, alibi_slopes=None) -> torch.Tensor:
    """csrc/attention/attention_kernels.cuh:75-485 via paged_attention_v1.cu:43-125.
    Cross-check: tests/kernels/attention/test_attention.py:50-118
    (ref_single_query_cached_kv_attention), atol 1e-3 there."""
    S, H, D = query.shape
    out = torch.zeros(S, H, D, dtype=query.dtype)
    G = H // num_kv_heads
    for s in range(S):
        L = int(seq_lens[s])
        if L == 0:
            continue
        for h in range(H):
            if h % G == 0:      # the G query heads of a kv head share one gathered (fp32) K / V
                k, v = _gather_kv(key_cache, value_cache, block_tables[s], L, h // G)
                k, v = k.float(), v.float()
            bias = None
            if alibi_slopes is not None and float(alibi_slopes[h]) != 0.0:
                pos = torch.arange(L, dtype=torch.float32)
                bias = float(alibi_slopes[h]) * (pos - L + 1)
            o, _, _ = _softmax_pv(query[s, h], k, v, scale, bias, query.dtype)
            out[s, h] = o.to(query.dtype)
    return out


def paged_attention_v2(query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, max_seq_len, alibi_slopes=None, partition_size=None):
    """attention_kernels.cuh:519-551 (partitions of 512 tokens, per-partition max/sum
    :338-346, tmp_out in scalar_t :481) + reduce :553-658.  Returns (out, exp_sums,
    max_logits, tmp_out); entries of partitions past seq_len are left at zero.
    `partition_size`: the same arithmetic with another partition length (the reference fixes 512,
    paged_attention_v2.cu:45; the MI355X launcher may split finer, mi355x_paged_attention_v2_ps)."""
    S, H, D = query.shape
    PARTITION_SIZE = partition_size or globals()["PARTITION_SIZE"]
    P = max((max_seq_len + PARTITION_SIZE - 1) // PARTITION_SIZE, 1)
    dt = query.dtype
    out = torch.zeros(S, H, D, dtype=dt)
    exp_sums = torch.zeros(S, H, P, dtype=torch.float32)
    max_logits = torch.zeros(S, H, P, dtype=torch.float32)
    tmp_out = torch.zeros(S, H, P, D, dtype=dt)
    G = H // num_kv_heads
    for s in range(S):
        L = int(seq_lens[s])
        np_ = (L + PARTITION_SIZE - 1) // PARTITION_SIZE
        for h in range(H):
            if L == 0:
                continue
            if h % G == 0:
                k, v = _gather_kv(key_cache, value_cache, block_tables[s], L, h // G)
                k, v = k.float(), v.float()
            for p in range(np_):
                lo, hi = p * PARTITION_SIZE, min((p + 1) * PARTITION_SIZE, L)
                bias = None
                if alibi_slopes is not None and float(alibi_slopes[h]) != 0.0:
                    pos = torch.arange(lo, hi, dtype=torch.float32)
                    bias = float(alibi_slopes[h]) * (pos - L + 1)
                o, m, sm = _softmax_pv(query[s, h], k[lo:hi], v[lo:hi], scale, bias, dt)
                tmp_out[s, h, p] = o.to(dt)
                exp_sums[s, h, p] = sm
                max_logits[s, h, p] = m
            if np_ == 1:
                out[s, h] = tmp_out[s, h, 0]            # :571-583 copy-through
            else:
                ml = max_logits[s, h, :np_]
                w = exp_sums[s, h, :np_] * torch.exp(ml - ml.max())
                inv = 1.0 / (w.sum() + 1e-6)
                acc = (tmp_out[s, h, :np_].float() * (w * inv)[:, None]).sum(0)
                out[s, h] = acc.to(dt)
    return out, exp_sums, max_logits, tmp_out


def paged_prefill_attention(query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                            seq_lens, cu_seqlens_q, sliding_window=None, softcap=None,
                            alibi_slopes=None) -> torch.Tensor:
    """Varlen causal (bottom-right aligned) GQA attention of the new tokens against the
    paged cache.  The arithmetic at the reference call site
    (vllm_metax/v1/attention/backends/flash_attn.py:725-747) is inside the closed
    flash_attn wheel; this follows the reference's own test oracle for that boundary,
    tests/kernels/attention/test_flash_attn.py:27-80 (ref_paged_attn: softmax of fp32
    scores, probabilities cast to v.dtype, then PV) — parity pinned by that oracle only."""
    T, H, D = query.shape
    G = H // num_kv_heads
    out = torch.zeros(T, H, D, dtype=query.dtype)
    for s in range(seq_lens.shape[0]):
        q0, q1 = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        ql, L = q1 - q0, int(seq_lens[s])
        if ql == 0:
            continue
        ctx = L - ql
        for h in range(H):
            if h % G == 0:      # the G query heads of a kv head share one gathered K / V
                k, v = _gather_kv(key_cache, value_cache, block_tables[s], L, h // G)
            sc = scale * (query[q0:q1, h].float() @ k.float().T)          # [ql, L]
            qi = torch.arange(ql)[:, None]
            ki = torch.arange(L)[None, :]
            if softcap:                                     # test_flash_attn.py:66-67: before the mask
                sc = softcap * torch.tanh(sc / softcap)
            if alibi_slopes is not None:
                # slope * (key position - query position): the decode kernel's bias (attention_kernels.cuh:286,
                # slope * (tok - seq_len + 1) for the one query at seq_len - 1) at every query position; the
                # reference's prefill oracle has no ALiBi case: parity unpinned
                sc = sc + float(alibi_slopes[h]) * (ki - (qi + ctx)).float()
            sc = sc.masked_fill(ki > (qi + ctx), float("-inf"))
            if sliding_window:                              # :60-65: keys pos - W + 1 .. pos stay visible
                sc = sc.masked_fill(ki < (qi + ctx - sliding_window + 1), float("-inf"))
            p = torch.softmax(sc, dim=-1).to(v.dtype)
            out[q0:q1, h] = (p.float() @ v.float()).to(query.dtype)
    return out


# =========================================================================== layernorm
def _rms(x32: torch.Tensor, eps: float) -> torch.Tensor:
    var = (x32 * x32).sum(dim=-1, keepdim=True) / x32.shape[-1]
    return torch.rsqrt(var + eps)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """csrc/layernorm_kernels.cu:12-41: out = ((T)(x*rsqrt(mean(x^2)+eps))) * w with the
    T*T product rounded to T.  Cross-check: tests/kernels/core/test_layernorm.py:40-64."""
    x32 = x.float()
    n = (x32 * _rms(x32, eps)).to(x.dtype)
    return (n.float() * weight.float()).to(x.dtype)


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """csrc/layernorm_kernels.cu:104-137 (and the equivalent vec8 path :47-99):
    z = T(x + residual) first; returns (normed -> written into input, z -> residual)."""
    z = (x.float() + residual.float()).to(x.dtype)
    return rms_norm(z, weight, eps), z


def _to_fp8_sat(y32: torch.Tensor) -> torch.Tensor:
    """csrc/quantization/fp8/common.cuh:25-38: clamp to +-448, then RNE cast."""
    return y32.clamp(-FP8_MAX, FP8_MAX).to(FP8)


def rms_norm_static_fp8_quant(x, weight, scale, eps) -> torch.Tensor:
    """csrc/layernorm_quant_kernels.cu:20-54: fp8(norm * (1/scale))."""
    o = rms_norm(x, weight, eps).float()
    inv = (1.0 / scale.float()).item() if scale.numel() == 1 else 1.0 / scale.float()
    return _to_fp8_sat(o * np.float32(inv))


def fused_add_rms_norm_static_fp8_quant(x, residual, weight, scale, eps):
    """csrc/layernorm_quant_kernels.cu:60-164: returns (fp8 out, new residual)."""
    o, z = fused_add_rms_norm(x, residual, weight, eps)
    inv = np.float32((1.0 / scale.float()).item())
    return _to_fp8_sat(o.float() * inv), z


def rms_norm_dynamic_per_token_quant(x, weight, eps, scale_ub=None, residual=None):
    """csrc/quantization/fused_kernels/layernorm_utils.cuh:17-115 and
    fused_layernorm_dynamic_per_token_quant.cu:46-85: the residual sum is NOT rounded
    before the norm; per-token scale = max(min(absmax, ub)/448, 1/(448*512)); the fp8
    branch DIVIDES by the scale.  Returns (fp8 out, scales [T,1], new residual|None).
    Cross-check: tests/kernels/core/test_fused_quant_layernorm.py:40-74."""
    x32 = x.float()
    new_res = None
    if residual is not None:
        x32 = x32 + residual.float()
        new_res = x32.to(x.dtype)
    n = (x32 * _rms(x32, eps)).to(x.dtype)
    o = (n.float() * weight.float()).to(x.dtype).float()
    amax = o.abs().amax(dim=-1, keepdim=True)
    if scale_ub is not None:
        amax = torch.minimum(amax, scale_ub.float().reshape(()))
    scales = torch.clamp(amax / FP8_MAX, min=FP8_MIN_SCALE)
    return _to_fp8_sat(o / scales), scales.reshape(-1, 1), new_res


# =========================================================================== fp8 quant
def static_scaled_fp8_quant(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """csrc/quantization/fp8/common.cu:11-32: x * (1/scale).
    Cross-check: tests/kernels/quant_utils.py:22-97."""
    inv = np.float32(1.0) / np.float32(scale.float().item())
    return _to_fp8_sat(x.float() * inv)


def dynamic_scaled_fp8_quant(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """csrc/quantization/fp8/common.cu:34-89: scale = absmax/448 (atomic max over rows
    into a zeroed scalar), then x * (1/scale)."""
    scale = (x.float().abs().max() / FP8_MAX).reshape(1)
    inv = np.float32(1.0) / np.float32(scale.item())
    return _to_fp8_sat(x.float() * inv), scale


def dynamic_per_token_scaled_fp8_quant(x, scale_ub=None):
    """csrc/quantization/fp8/common.cu:91-133: per-token scale, fp8(x / scale)."""
    x32 = x.float()
    amax = x32.abs().amax(dim=-1, keepdim=True)
    if scale_ub is not None:
        amax = torch.minimum(amax, scale_ub.float().reshape(()))
    scales = torch.clamp(amax / FP8_MAX, min=FP8_MIN_SCALE)
    return _to_fp8_sat(x32 / scales), scales


# ============================================================================== rotary
def rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox):
    """csrc/pos_encoding_kernels.cu:10-34,37-100: in scalar_t, rounding after each
    multiply and after the add/sub.  query [T, H*hs] or [T, H, hs]; returns new (q, k).
    Cross-check: tests/kernels/core/test_pos_encoding.py:49-113 (upstream forward_native)."""
    dt = query.dtype
    rot = cos_sin_cache.shape[1]
    emb = rot // 2
    T = positions.numel()
    cs = cos_sin_cache[positions.reshape(-1).long()]       # [T, rot]
    cos, sin = cs[:, None, :emb], cs[:, None, emb:]         # [T, 1, emb]

    def apply(x):
        if x is None:
            return None
        shape = x.shape
        xv = x.reshape(T, -1, head_size).clone()
        r = xv[..., :rot]
        if is_neox:
            a, b = r[..., :emb], r[..., emb:]
        else:
            a, b = r[..., 0::2], r[..., 1::2]

        def mul(u, v):
            return (u.float() * v.float()).to(dt)
        na = (mul(a, cos).float() - mul(b, sin).float()).to(dt)
        nb = (mul(b, cos).float() + mul(a, sin).float()).to(dt)
        if is_neox:
            xv[..., :emb], xv[..., emb:rot] = na, nb
        else:
            xv[..., 0:rot:2], xv[..., 1:rot:2] = na, nb
        return xv.reshape(shape)

    return apply(query), apply(key)


def batched_rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox, cos_sin_cache_offsets):
    """csrc/pos_encoding_kernels.cu:102-129: cache row = positions + cos_sin_cache_offsets (several LoRA
    rope tables stacked in one cache).  Cross-check: tests/kernels/core/test_pos_encoding.py:126-192."""
    return rotary_embedding(positions.reshape(-1) + cos_sin_cache_offsets.reshape(-1), query, key, head_size,
                            cos_sin_cache, is_neox)


# ========================================================================== activation
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """csrc/activation_kernels.cu:14-36,142-147: T(x/(1+exp(-x))) * y, product rounded."""
    d = x.shape[-1] // 2
    a, b = x[..., :d].float(), x[..., d:]
    s = (a / (1.0 + torch.exp(-a))).to(x.dtype)
    return (s.float() * b.float()).to(x.dtype)


def silu_and_mul_quant(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """csrc/quantization/activation_kernels.cu:21-90: the T-rounded silu(x) * y (as silu_and_mul),
    times inverted_scale = 1 / *scale (:57), clamped to +-448 and cast RNE to e4m3
    (scaled_fp8_conversion<true>, csrc/quantization/fp8/common.cuh:25-38)."""
    inv = np.float32(1.0) / np.float32(scale.float().item())
    return _to_fp8_sat(silu_and_mul(x).float() * inv)


# ================================================================ merge_attn_states
def merge_attn_states(prefix_output: torch.Tensor, prefix_lse: torch.Tensor,
                      suffix_output: torch.Tensor, suffix_lse: torch.Tensor):
    """csrc/attention/merge_attn_states.cu:43-87 restated in fp32 with its rounding points: lse ==
    +inf -> -inf (:45-46), m = max, se = exp(lse - m), scale = se / (p_se + s_se), out =
    fma(p_out, p_scale, fl32(s_out * s_scale)) rounded once to scalar_t (:73-76), out_lse =
    log(p_se + s_se) + m (:84-85).  Returns (output [T,H,D], output_lse [H,T]).
    Cross-check: the in-test torch reference tests/kernels/attention/test_merge_attn_states.py:16-45
    (same formula, without the fma)."""
    p_lse = prefix_lse.float().clone()
    s_lse = suffix_lse.float().clone()
    p_lse[torch.isinf(p_lse)] = -float("inf")
    s_lse[torch.isinf(s_lse)] = -float("inf")
    m = torch.maximum(p_lse, s_lse)
    p_se = torch.exp(p_lse - m)
    s_se = torch.exp(s_lse - m)
    out_se = p_se + s_se
    out_lse = torch.log(out_se) + m
    p_scale = (p_se / out_se).t().unsqueeze(-1)          # [T, H, 1] fp32
    s_scale = (s_se / out_se).t().unsqueeze(-1)
    t = (suffix_output.float() * s_scale)                # fp32-rounded product
    # fma(p, p_scale, t): p * p_scale is exact in float64 (24 + 24 bits), one rounding at the end
    out = (prefix_output.double() * p_scale.double() + t.double()).float()
    return out.to(prefix_output.dtype), out_lse


# ============================================================== int4 weight-only (AWQ)
AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]       # nibble i of an AWQ word holds column AWQ_ORDER[i]
EXL_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]       # nibble p of a shuffled word holds k-row EXL_ORDER[p]


def _unpack_nibbles(words: np.ndarray) -> np.ndarray:
    """[..., W] uint32 -> [..., W, 8] nibble p = (w >> 4p) & 15."""
    w = words.astype(np.uint32)[..., None]
    return ((w >> (4 * np.arange(8, dtype=np.uint32))) & 0xF).astype(np.uint8)


def _pack_nibbles(nib: np.ndarray) -> np.ndarray:
    """[..., 8] -> uint32 with nibble p = nib[..., p]."""
    out = np.zeros(nib.shape[:-1], dtype=np.uint32)
    for p in range(8):
        out |= nib[..., p].astype(np.uint32) << np.uint32(4 * p)
    return out


def awq_unpack(qweight: torch.Tensor) -> np.ndarray:
    """AWQ int32 [R, N/8] -> uint8 [R, N] in natural column order.
    Layout per tests/kernels/quantization/test_awq_triton.py:17-32 (reverse_awq_order)."""
    nib = _unpack_nibbles(qweight.numpy().view(np.uint32))        # [R, N/8, 8]
    nat = np.empty_like(nib)
    for i, col in enumerate(AWQ_ORDER):
        nat[..., col] = nib[..., i]
    return nat.reshape(nib.shape[0], -1)


def awq_pack(w: np.ndarray) -> torch.Tensor:
    """uint8 [R, N] -> AWQ-packed int32 [R, N/8] (inverse of awq_unpack)."""
    r, n = w.shape
    nat = w.reshape(r, n // 8, 8)
    nib = np.empty_like(nat)
    for i, col in enumerate(AWQ_ORDER):
        nib[..., i] = nat[..., col]
    return torch.from_numpy(_pack_nibbles(nib).view(np.int32))


def awq_to_gptq_4bit(qweight: torch.Tensor) -> torch.Tensor:
    """csrc/quantization/awq/gemm_kernels.cu:127-184: output memory [K/8, N] words,
    nibble gptq_shift[j] of word (kk, n) = W[8kk + j, n], gptq_shift = {0,4,1,5,2,6,3,7},
    i.e. nibble p holds k-row EXL_ORDER[p]; the tensor is DECLARED [N, K/8] (:347-348).
    parity unpinned by any reference test (test_awq.py:11-47 is opcheck-only)."""
    w = awq_unpack(qweight)                                        # [K, N]
    k, n = w.shape
    blk = w.reshape(k // 8, 8, n)                                  # [kk, j, n]
    nib = np.empty((k // 8, n, 8), dtype=np.uint8)
    for p, j in enumerate(EXL_ORDER):
        nib[..., p] = blk[:, j, :]
    words = _pack_nibbles(nib)                                     # [K/8, N]
    return torch.from_numpy(words.view(np.int32).reshape(n, k // 8).copy())


def exllama_unpack(qweight_mem: np.ndarray) -> np.ndarray:
    """shuffled words [K/8, N] -> uint8 [K, N]."""
    nib = _unpack_nibbles(qweight_mem.view(np.uint32))             # [K/8, N, 8]
    kk, n, _ = nib.shape
    w = np.empty((kk, 8, n), dtype=np.uint8)
    for p, j in enumerate(EXL_ORDER):
        w[:, j, :] = nib[..., p]
    return w.reshape(kk * 8, n)


def w4_dequant(w: np.ndarray, zeros: np.ndarray, scales: torch.Tensor, group: int) -> torch.Tensor:
    """w[K,N], zeros[K/g,N] integer, scales [K/g,N] of dtype T ->
    T( float(q - z) * float(s) ).  The reference computes fma(q, s, (-z)*s) in fp32
    (hgemm_gptq.h:487-570, 869-905) and then rounds to T (:187-196); both products are
    exact in fp32 for 4-bit q,z and a <=11-bit-mantissa s, so this is the same value."""
    k, n = w.shape
    z = np.repeat(zeros.astype(np.int32), group, axis=0)[:k]
    s = scales.float().repeat_interleave(group, dim=0)[:k]
    d = torch.from_numpy(w.astype(np.int32) - z).float() * s
    return d.to(scales.dtype)


def awq_dequantize(qweight, scales, qzeros) -> torch.Tensor:
    """csrc/quantization/awq/gemm_kernels.cu:96-125 + hgemm_gptq.h:373-485:
    (q - z) * s on the ORIGINAL AWQ layout -> [K, N].
    Cross-check: tests/kernels/quantization/test_awq_triton.py:36-62."""
    w = awq_unpack(qweight)
    z = awq_unpack(qzeros)
    return w4_dequant(w, z, scales, w.shape[0] // scales.shape[0])


def awq_gemm(x, qweight_gptq_layout, scales, qzeros) -> torch.Tensor:
    """csrc/quantization/awq/gemm_kernels.cu:410-463 -> hgemm_gptq.h:2165-2259:
    Y = X . T((q - z) * s), fp32 accumulation, one rounding of Y to T.
    qweight is the awq_to_gptq_4bit output (declared [N, K/8], memory [K/8, N])."""
    n, k8 = qweight_gptq_layout.shape
    mem = qweight_gptq_layout.numpy().reshape(k8, n)
    w = exllama_unpack(mem)
    z = awq_unpack(qzeros)
    wd = w4_dequant(w, z, scales, w.shape[0] // scales.shape[0])
    return (x.double() @ wd.double()).to(x.dtype)


# ============================================================= int4 weight-only (GPTQ)
def gptq_unpack_rows(qweight: torch.Tensor) -> np.ndarray:
    """checkpoint GPTQ qweight [K/8, N]: nibble p of word (kk, n) = W[8kk + p, n]."""
    nib = _unpack_nibbles(qweight.numpy().view(np.uint32))         # [K/8, N, 8]
    return nib.transpose(0, 2, 1).reshape(-1, nib.shape[1])


def gptq_pack_rows(w: np.ndarray) -> torch.Tensor:
    k, n = w.shape
    nib = w.reshape(k // 8, 8, n).transpose(0, 2, 1)
    return torch.from_numpy(_pack_nibbles(nib).view(np.int32))


def gptq_unpack_zeros(qzeros: torch.Tensor) -> np.ndarray:
    """GPTQ qzeros [K/g, N/8]: nibble i of word j = zero of column 8j + i (natural)."""
    nib = _unpack_nibbles(qzeros.numpy().view(np.uint32))
    return nib.reshape(nib.shape[0], -1)


def gptq_shuffle(qweight: torch.Tensor, q_perm: Optional[torch.Tensor]) -> torch.Tensor:
    """csrc/quantization/gptq/q_gemm.cu:2321-2368: optional make_sequential (:2145-2174,
    new row i = old row q_perm[i]) then shuffle_4bit_8 (qdq_4.cuh:16-29).  Returns the
    shuffled words [K/8, N].  parity unpinned (test_gptq.py:10-32 is opcheck-only)."""
    w = gptq_unpack_rows(qweight)
    if q_perm is not None and q_perm.numel() > 0:
        w = w[q_perm.long().numpy()]
    k, n = w.shape
    blk = w.reshape(k // 8, 8, n)
    nib = np.empty((k // 8, n, 8), dtype=np.uint8)
    for p, j in enumerate(EXL_ORDER):
        nib[..., p] = blk[:, j, :]
    return torch.from_numpy(_pack_nibbles(nib).view(np.int32))


def gptq_gemm(x, qweight_shuffled, qzeros, scales, g_idx_perm, group_size) -> torch.Tensor:
    """csrc/quantization/gptq/q_gemm.cu:2373-2413: Y = X[:, perm] . T((q - (z+1)) * s).
    The reference's fast path hard-codes zero = 8 (hgemm_gptq.h:870-877); its generic
    exllama path uses qzeros + 1 (q_gemm.cu:247-250).  They agree on symmetric
    checkpoints (stored zero 7); this oracle — like the build — uses qzeros + 1."""
    w = exllama_unpack(qweight_shuffled.numpy())
    z = gptq_unpack_zeros(qzeros).astype(np.int32) + 1
    wd = w4_dequant(w, z, scales, group_size)
    xx = x
    if g_idx_perm is not None and g_idx_perm.numel() > 0:
        xx = x[:, g_idx_perm.long()]
    return (xx.double() @ wd.double()).to(x.dtype)


def gptq8_unpack_rows(qweight: torch.Tensor) -> np.ndarray:
    """8-bit GPTQ qweight [K/4, N]: byte i of word (kk, n) = W[4kk + i, n] (the exllama 8-bit shuffle is
    the identity: csrc/quantization/gptq/qdq_8.cuh:14)."""
    w = qweight.numpy().view(np.uint32)[..., None]
    b = ((w >> (8 * np.arange(4, dtype=np.uint32))) & 0xFF).astype(np.uint8)        # [K/4, N, 4]
    return b.transpose(0, 2, 1).reshape(-1, b.shape[1])


def gptq8_pack_rows(w: np.ndarray) -> torch.Tensor:
    k, n = w.shape
    b = w.reshape(k // 4, 4, n).astype(np.uint32)
    out = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16) | (b[:, 3] << 24)
    return torch.from_numpy(out.view(np.int32))


def gptq8_gemm(x, qweight, qzeros, scales, group_size) -> torch.Tensor:
    """gptq_gemm with bit = 8 (csrc/quantization/gptq/q_gemm.cu:1998-2003, hgemm_gptq.h:892-899): the
    reference's fast path fixes the zero at 128 (kU8B128: zs = -128 * s); the generic exllama path uses
    qzeros + 1 (q_gemm.cu:688).  They agree on symmetric checkpoints (stored zero 127); like the 4-bit
    path this oracle — and the build — uses qzeros + 1.  qzeros [K/g, N/4], byte i of word j = column
    4j + i.  parity unpinned (no numeric reference test exists for gptq_gemm)."""
    w = gptq8_unpack_rows(qweight).astype(np.int32)
    zb = qzeros.numpy().view(np.uint32)[..., None]
    z = ((zb >> (8 * np.arange(4, dtype=np.uint32))) & 0xFF).astype(np.int32).reshape(qzeros.shape[0], -1) + 1
    zz = np.repeat(z, group_size, axis=0)[:w.shape[0]]
    s = scales.float().repeat_interleave(group_size, dim=0)[:w.shape[0]]
    wd = (torch.from_numpy(w - zz).float() * s).to(scales.dtype)
    return (x.double() @ wd.double()).to(x.dtype)


# ============================================================================ fp8 GEMM
def scaled_mm_fp8(a, b, a_scales, b_scales, out_dtype, bias=None) -> torch.Tensor:
    """Behind the schema cutlass_scaled_mm (csrc/torch_bindings.cpp:251-256).  The
    reference has no fp8 implementation (scaled_mm_entry.cu:22-24), so the oracle is the
    reference's test-side definition tests/kernels/utils.py:1231-1270 (baseline_scaled_mm):
    out = (a_scales * a) @ (b_scales * b) + bias, computed in fp32 and cast once."""
    o = (a_scales.float().reshape(-1, 1) * a.float()).double() @ \
        (b.float() * b_scales.float().reshape(1, -1)).double()
    if bias is not None:
        o = o + bias.double()
    return o.to(out_dtype)


# ======================================================================= int8 W8A8 (§8f-4)
def scaled_int8_quant(x: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """csrc/quantization/compressed_tensors/int8_quant_kernels.cu:12-22 (rn: round-half-even, clamp
    to [-127, 127]), :50-68 (static: x / scale), :94-135 (dynamic per token: scale = absmax / 127,
    q = rn(x * (127 / absmax)), 0 when absmax == 0).  Returns (int8, scales [T, 1] or the scale)."""
    xf = x.float()
    if scale is not None:
        q = torch.round(xf / scale.float().reshape(()))
        return q.clamp(-127, 127).to(torch.int8), scale
    absmax = xf.abs().amax(dim=-1, keepdim=True)
    # (tensor / tensor: `127.0 / t` is evaluated by torch as t.reciprocal() * 127 — two roundings)
    inv = torch.where(absmax == 0, torch.zeros_like(absmax), torch.full_like(absmax, 127.0) / absmax)
    q = torch.round(xf * inv).clamp(-127, 127).to(torch.int8)
    return q, (absmax / 127.0).reshape(-1, 1)


def scaled_mm_int8(a, b, a_scales, b_scales, out_dtype, bias=None) -> torch.Tensor:
    """The int8 branch of cutlass_scaled_mm (scaled_mm_entry.cu:34-39, :84-140): exact int32
    accumulation; epilogue in fp32 a_s * (b_s * acc) (+ bias), one rounding (the order of the
    reference's test-side definition tests/kernels/utils.py baseline_scaled_mm)."""
    acc = (a.to(torch.int64) @ b.to(torch.int64)).to(torch.float32)       # exact: |acc| < 2^24 here
    o = a_scales.float().reshape(-1, 1) * (b_scales.float().reshape(1, -1) * acc)
    if bias is not None:
        o = o + bias.float()
    return o.to(out_dtype)
