"""Attention backend of the MI355X plugin: paged attention over the x-split KV cache.

Structure follows the reference's smallest complete backend, triton_attn.py (metadata
dataclass :38-65, builder with full-graph support :68-146, backend class :149-203, impl that
writes the KV cache and then runs attention on query[:num_actual_tokens] :267-373), with the
decode / prefill split of its default backend (flash_attn.py:343-379, split computed in build())
— but WITHOUT that backend's per-layer host syncs (flash_attn.py:726-730 `.tolist()` +
cumsum): every index tensor the kernels need is derived once per step in `build_metadata`.

KV cache: shape (2, num_blocks, block_size * num_kv_heads * head_size); plane 0 is viewed as
[nb, kvh, d/x, bs, x] (keys), plane 1 as [nb, kvh, d, bs] (values) — the layout
reshape_and_cache / paged_attention_v1/v2 are defined on (csrc/cache_kernels.cu:203-255,
attention_kernels.cuh:86-89).

The core (`build_metadata`, `paged_attention_forward`) has no vLLM dependency; the vLLM-facing
classes at the bottom are thin and only defined when vLLM imports.
"""
from __future__ import annotations

import dataclasses
import functools
import math
from typing import Optional, Tuple

import torch

from .. import _custom_ops as ops

PARTITION_SIZE = ops.PARTITION_SIZE


@dataclasses.dataclass
class Mi355xPagedMetadata:
    # |---------------- seq_len ----------------|
    # |------ context_len ------|-- query_len --|
    num_actual_tokens: int          # tokens excluding graph padding
    max_query_len: int
    max_seq_len: int
    query_start_loc: torch.Tensor   # int32 [R+1]
    seq_lens: torch.Tensor          # int32 [R]
    block_table: torch.Tensor       # int32 [R, max_blocks]
    slot_mapping: torch.Tensor      # int64 [T]
    # decode-first split (requests with query_len == 1 come first)
    num_decodes: int = 0
    num_decode_tokens: int = 0
    num_prefills: int = 0
    num_prefill_tokens: int = 0
    max_decode_seq_len: int = 0
    max_prefill_query_len: int = 0
    prefill_query_start_loc: Optional[torch.Tensor] = None  # int32 [Rp+1], rebased to 0
    # split-KV workspaces for the decode kernel (shared by all layers; views of the builder's
    # persistent buffers when a DecodeWorkspace is given, so that a captured graph never writes to
    # freed memory)
    exp_sums: Optional[torch.Tensor] = None
    max_logits: Optional[torch.Tensor] = None
    tmp_out: Optional[torch.Tensor] = None
    # v1 / v2 decided once per step from the launcher's own LDS budget (None: decide per call)
    use_v1: Optional[bool] = None
    # tokens per split-KV partition of the v2 launch (decode_partition_size)
    partition_size: int = PARTITION_SIZE


MIN_PARTITION_SIZE = 128     # finest split decode_partition_size chooses
TARGET_DECODE_WGS = 256      # one workgroup per CU


def decode_heads_per_workgroup(num_heads: int, num_kv_heads: int) -> int:
    """Query heads of one kv head that one decode workgroup serves (csrc/paged_attention.hip::pa_plan)."""
    q_per_kv = num_heads // num_kv_heads
    return 8 if q_per_kv >= 5 else (4 if q_per_kv >= 3 else q_per_kv)


def decode_partition_size(num_seqs: int, num_heads: int, num_kv_heads: int, max_seq_len: int,
                          block_size: int) -> int:
    """Tokens per split-KV partition for paged_attention_v2.  The reference's launcher fixes 512
    (csrc/attention/paged_attention_v2.cu:45) and its grid is (heads, seqs, partitions); here one workgroup
    serves all (up to 8) query heads of a kv head, so a TP = 8 rank of a 70B model at batch 64 (8 q / 1 kv
    head) has 64 x ceil(L / 512) workgroups for 256 CUs.  When 512-token partitions give fewer than one
    workgroup per CU the partitions are cut finer — never below 128 tokens — so that about one workgroup per CU
    streams the cache (measured: profiles/r03 notes).  A multiple of the block size and of 16."""
    gt = decode_heads_per_workgroup(num_heads, num_kv_heads)
    base = num_seqs * num_kv_heads * (-(-(num_heads // num_kv_heads) // gt))
    if base <= 0 or max_seq_len <= MIN_PARTITION_SIZE:
        return PARTITION_SIZE
    if base * (-(-max_seq_len // PARTITION_SIZE)) >= TARGET_DECODE_WGS:
        return PARTITION_SIZE
    want = -(-TARGET_DECODE_WGS // base)                  # partitions per sequence
    unit = block_size * 16 // math.gcd(block_size, 16)
    ps = -(-max_seq_len // want)
    ps = -(-ps // unit) * unit
    return max(MIN_PARTITION_SIZE, min(PARTITION_SIZE, ps))


class DecodeWorkspace:
    """Split-KV buffers of paged_attention_v2, allocated ONCE for (max_num_seqs, ceil(max_model_len /
    512)) — plus room for the finer partitions decode_partition_size picks for small batches — and handed out as views.  build() used to torch.empty() them per step: under a full HIP-graph
    capture the graph keeps the capture-time addresses, which the allocator may have reused by replay
    time.  Invariant: every pointer a captured decode launch sees belongs to a buffer that
    outlives the graph."""

    def __init__(self, max_num_seqs: int, num_heads: int, head_size: int, max_model_len: int,
                 dtype: torch.dtype, device):
        self.max_num_seqs = max_num_seqs
        self.max_parts = max((max_model_len + PARTITION_SIZE - 1) // PARTITION_SIZE, 1)
        self.max_model_len = max_model_len
        # (sequence, partition) pairs: 512-token partitions for a full batch; a finer split is only chosen while
        # seqs * partitions stays below ~TARGET_DECODE_WGS + seqs (decode_partition_size)
        self.max_pairs = max_num_seqs * (self.max_parts + 1) + TARGET_DECODE_WGS
        self.exp_sums = torch.empty(self.max_pairs, num_heads, dtype=torch.float32, device=device)
        self.max_logits = torch.empty_like(self.exp_sums)
        self.tmp_out = torch.empty(self.max_pairs, num_heads, head_size, dtype=dtype, device=device)

    def views(self, num_seqs: int, parts: int):
        if num_seqs > self.max_num_seqs or num_seqs * parts > self.max_pairs:
            raise RuntimeError(f"decode workspace too small: {num_seqs} seqs x {parts} partitions "
                               f"(sized for {self.max_pairs} (sequence, partition) pairs)")
        n = num_seqs
        h, d = self.exp_sums.shape[-1], self.tmp_out.shape[-1]
        # contiguous [n, h, parts(, d)] views of the flat storage: the kernels index with `parts` as the
        # partition stride, so the views must be dense in the shape they are launched with
        es = self.exp_sums.view(-1)[:n * h * parts].view(n, h, parts)
        ml = self.max_logits.view(-1)[:n * h * parts].view(n, h, parts)
        to = self.tmp_out.view(-1)[:n * h * parts * d].view(n, h, parts, d)
        return es, ml, to


def split_decodes_and_prefills(query_lens_cpu, decode_threshold: int = 1) -> Tuple[int, int, int, int]:
    """Same contract as upstream's helper used at flash_attn.py:307-314: the batch is already
    ordered decode-first; returns (num_decodes, num_prefills, num_decode_tokens,
    num_prefill_tokens) where decodes are the leading requests with query_len <= threshold."""
    q = [int(v) for v in query_lens_cpu]
    n = len(q)
    first_prefill = n
    for i, v in enumerate(q):
        if v > decode_threshold:
            first_prefill = i
            break
    num_decodes = first_prefill
    num_decode_tokens = sum(q[:first_prefill])
    return num_decodes, n - num_decodes, num_decode_tokens, sum(q) - num_decode_tokens


def build_metadata(query_start_loc: torch.Tensor, query_start_loc_cpu, seq_lens: torch.Tensor,
                   seq_lens_cpu, block_table: torch.Tensor, slot_mapping: torch.Tensor,
                   num_actual_tokens: int, max_query_len: int, max_seq_len: int, num_heads: int,
                   head_size: int, dtype: torch.dtype, workspace: Optional[DecodeWorkspace] = None,
                   num_kv_heads: Optional[int] = None, block_size: int = 16,
                   fixed_decode_len: Optional[int] = None) -> Mi355xPagedMetadata:
    """Everything forward() needs, computed once per step (cf. flash_attn.py:286-526).
    `*_cpu` are host copies (lists / CPU tensors) that vLLM's CommonAttentionMetadata already
    carries, so no device->host sync happens here either."""
    qsl = [int(v) for v in query_start_loc_cpu]
    q_lens = [qsl[i + 1] - qsl[i] for i in range(len(qsl) - 1)]
    nd, npf, ndt, npt = split_decodes_and_prefills(q_lens)
    sl_cpu = [int(v) for v in seq_lens_cpu]
    md = Mi355xPagedMetadata(
        num_actual_tokens=num_actual_tokens, max_query_len=max_query_len, max_seq_len=max_seq_len,
        query_start_loc=query_start_loc, seq_lens=seq_lens, block_table=block_table,
        slot_mapping=slot_mapping, num_decodes=nd, num_decode_tokens=ndt, num_prefills=npf,
        num_prefill_tokens=npt)
    dev = seq_lens.device
    if nd > 0:
        # `fixed_decode_len` (HIP-graph capture / replay): every host-side decision of the decode
        # launch — v1 vs v2, the partition count = grid.z, the LDS size — is taken from this fixed
        # value instead of the batch's current maximum, so a replay with longer sequences stays inside
        # what was captured (the kernels read the true lengths from seq_lens on the device)
        md.max_decode_seq_len = fixed_decode_len if fixed_decode_len else (max(sl_cpu[:nd]) if sl_cpu else 0)
        if num_kv_heads is not None:
            md.partition_size = decode_partition_size(nd, num_heads, num_kv_heads, md.max_decode_seq_len, block_size)
        parts = max((md.max_decode_seq_len + md.partition_size - 1) // md.partition_size, 1)
        if workspace is not None:
            md.exp_sums, md.max_logits, md.tmp_out = workspace.views(nd, parts)
        else:
            md.exp_sums = torch.empty(nd, num_heads, parts, dtype=torch.float32, device=dev)
            md.max_logits = torch.empty_like(md.exp_sums)
            md.tmp_out = torch.empty(nd, num_heads, parts, head_size, dtype=dtype, device=dev)
        if num_kv_heads is not None:
            md.use_v1 = use_paged_attention_v1(nd, num_heads, md.max_decode_seq_len, num_kv_heads, head_size,
                                               block_size, dtype)
    if npf > 0:
        md.max_prefill_query_len = max(q_lens[nd:])
        md.prefill_query_start_loc = (query_start_loc[nd:] - query_start_loc[nd:nd + 1]).to(torch.int32)
    return md


def kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> Tuple[int, ...]:
    return (2, num_blocks, block_size * num_kv_heads * head_size)


def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int, head_size: int):
    """(2, nb, bs*kvh*d) -> key_cache [nb, kvh, d/x, bs, x], value_cache [nb, kvh, d, bs]."""
    x = 16 // kv_cache.element_size()
    nb = kv_cache.shape[1]
    key_cache = kv_cache[0].view(nb, num_kv_heads, head_size // x, -1, x)
    value_cache = kv_cache[1].view(nb, num_kv_heads, head_size, -1)
    return key_cache, value_cache


@functools.lru_cache(maxsize=None)
def v1_max_seq_len(num_heads: int, num_kv_heads: int, head_size: int, block_size: int,
                   dtype: torch.dtype) -> int:
    """Largest context paged_attention_v1 takes for this head geometry: the launcher's own LDS budget
    (mi355x_paged_attention_v1_max_seq_len; the reference sizes its LDS the same way,
    csrc/attention/paged_attention_v1.cu:77-87) — 6656 tokens for 32/8 heads in bf16, 4928 in fp32."""
    return ops.paged_attention_v1_max_seq_len(1, num_heads, num_kv_heads, head_size, block_size, dtype)


def v1_v2_rule(num_seqs: int, num_heads: int, max_seq_len: int, v1_limit: int) -> bool:
    """The v1 / v2 choice of the upstream caller (vllm/attention/ops/paged_attn.py,
    PagedAttention.forward_decode: v1 when max_seq_len <= 8192 and either a single 512-token partition
    covers it or there are already > 512 (sequence, head) pairs to spread) with its fixed 8192 replaced
    by what one workgroup's LDS really holds for this geometry (`v1_limit`): beyond it only v2 runs."""
    max_parts = (max_seq_len + 511) // 512
    return max_seq_len <= min(8192, v1_limit) and (max_parts == 1 or num_seqs * num_heads > 512)


def use_paged_attention_v1(num_seqs: int, num_heads: int, max_seq_len: int, num_kv_heads: int,
                           head_size: int, block_size: int, dtype: torch.dtype) -> bool:
    return v1_v2_rule(num_seqs, num_heads, max_seq_len,
                      v1_max_seq_len(num_heads, num_kv_heads, head_size, block_size, dtype))


def decode_attention(out: torch.Tensor, exp_sums: torch.Tensor, max_logits: torch.Tensor,
                     tmp_out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                     value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                     block_table: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                     max_seq_len: int, alibi_slopes: Optional[torch.Tensor] = None,
                     kv_cache_dtype: str = "auto", k_scale: Optional[torch.Tensor] = None,
                     v_scale: Optional[torch.Tensor] = None, use_v1: Optional[bool] = None,
                     partition_size: int = PARTITION_SIZE) -> None:
    """paged_attention_v1 or _v2, chosen like the upstream caller the reference plugs into
    (vllm/attention/ops/paged_attn.py, PagedAttention.forward_decode): v1 when the context is
    short enough for one workgroup's LDS and there is already enough parallelism without
    partitioning (num_seqs * num_heads > 512), else the 512-token-partition kernel + reduce.
    Measured at 64 seqs x 32 heads, ctx 1088: v1 63.9 us, v2 67.1 us (scripts/bench_attn.py)."""
    if use_v1 is None:
        use_v1 = use_paged_attention_v1(query.shape[0], query.shape[1], max_seq_len, num_kv_heads,
                                        query.shape[2], block_size, query.dtype)
    if use_v1:
        ops.paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_table,
                               seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                               v_scale)
    else:
        ops.paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                               num_kv_heads, scale, block_table, seq_lens, block_size, max_seq_len,
                               alibi_slopes, kv_cache_dtype, k_scale, v_scale, partition_size=partition_size)


def decode_attention_fused(out: torch.Tensor, exp_sums: torch.Tensor, max_logits: torch.Tensor,
                           tmp_out: torch.Tensor, qkv: torch.Tensor, slabs: Optional[torch.Tensor], sk: int,
                           positions: torch.Tensor, cos_sin_cache: torch.Tensor, slot_mapping: torch.Tensor,
                           key_cache: torch.Tensor, value_cache: torch.Tensor, num_heads: int, num_kv_heads: int,
                           scale: float, block_table: torch.Tensor, seq_lens: torch.Tensor, block_size: int,
                           max_seq_len: int, use_v1: Optional[bool] = None,
                           partition_size: int = PARTITION_SIZE, slab_scales=None, quant_out=None) -> bool:
    """qkv_rope_cache + decode_attention in one launch (MI355X-side fusion, include/mi355x_hotpath.h), with the
    same v1 / v2 choice as decode_attention.  False: not applicable to these shapes, nothing was launched."""
    head_size = key_cache.shape[2] * key_cache.shape[4]
    if use_v1 is None:
        use_v1 = use_paged_attention_v1(qkv.shape[0], num_heads, max_seq_len, num_kv_heads, head_size,
                                        block_size, qkv.dtype)
    return ops.paged_attention_fused_qkv(out, exp_sums, max_logits, tmp_out, qkv, slabs, sk, positions,
                                         cos_sin_cache, slot_mapping, key_cache, value_cache, num_heads,
                                         num_kv_heads, scale, block_table, seq_lens, block_size, max_seq_len,
                                         not use_v1, partition_size, slab_scales=slab_scales,
                                         quant_out=quant_out if not use_v1 else None)


def _qkv_rows(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor) -> Optional[torch.Tensor]:
    """[tokens, width] view of the buffer q | k | v are column ranges of (upstream's attention layers hand the backend
    the three `qkv.split(...)` views of ONE qkv_proj output, rotated in place), or None when they are separate tensors."""
    try:
        if query.dim() != 3 or key.dim() != 3 or value.dim() != 3 or query.shape[0] == 0:
            return None
        T, H, d = query.shape
        kvh = key.shape[1]
        es = query.element_size()
        if key.shape != (T, kvh, d) or value.shape != (T, kvh, d) or query.dtype != key.dtype or key.dtype != value.dtype:
            return None
        w = query.stride(0)
        if query.stride() != (w, d, 1) or key.stride() != (w, d, 1) or value.stride() != (w, d, 1):
            return None
        if w < (H + 2 * kvh) * d or w % 8 or query.data_ptr() % 16:
            return None
        if key.data_ptr() != query.data_ptr() + H * d * es or value.data_ptr() != key.data_ptr() + kvh * d * es:
            return None
        if query.untyped_storage().data_ptr() != key.untyped_storage().data_ptr() or \
                key.untyped_storage().data_ptr() != value.untyped_storage().data_ptr():
            return None
        return torch.as_strided(query, (T, (H + 2 * kvh) * d), (w, 1), query.storage_offset())
    except (RuntimeError, AttributeError):
        return None


FUSE_DECODE_CACHE_WRITE = True   # (tests / A-B: False runs reshape_and_cache + decode attention as two launches)


def paged_attention_forward(query: torch.Tensor, key: Optional[torch.Tensor],
                            value: Optional[torch.Tensor], kv_cache: torch.Tensor,
                            md: Mi355xPagedMetadata, output: torch.Tensor, num_kv_heads: int,
                            scale: float, alibi_slopes: Optional[torch.Tensor] = None,
                            kv_cache_dtype: str = "auto", k_scale: Optional[torch.Tensor] = None,
                            v_scale: Optional[torch.Tensor] = None, sliding_window: Optional[int] = None,
                            softcap: Optional[float] = None) -> torch.Tensor:
    """query [T, H, d], key/value [T, KVH, d] (may be padded past num_actual_tokens),
    output [T, H, d] caller-provided (accept_output_buffer, flash_attn.py:56)."""
    head_size = query.shape[-1]
    key_cache, value_cache = split_kv_cache(kv_cache, num_kv_heads, head_size)
    block_size = key_cache.shape[3]
    n = md.num_actual_tokens
    nd, ndt = md.num_decodes, md.num_decode_tokens
    if nd > 0 and (sliding_window or softcap):
        # paged_attention_v1/v2 have no such arguments in the reference either (schema
        # csrc/torch_bindings.cpp:45-69); the prefill kernel takes them (flash_attn.py:725-747)
        raise RuntimeError("sliding window / soft-cap are supported on the prefill path only")
    # Decode tokens whose q | k | v are the split views of one qkv buffer (what upstream's attention layers pass):
    # the cache write of the new token runs in the prologue of the decode attention launch
    # (mi355x_paged_attention_fused_qkv without rotary: q and k are already rotated) — one launch less per layer and
    # step, bit-identical to reshape_and_cache + paged_attention_v1 / _v2.
    fused_decode = False
    if FUSE_DECODE_CACHE_WRITE and nd > 0 and ndt == nd and key is not None and value is not None \
            and kv_cache_dtype == "auto" and alibi_slopes is None:
        rows = _qkv_rows(query[:ndt], key[:ndt], value[:ndt])
        if rows is not None:
            use_v1 = md.use_v1
            if use_v1 is None:
                use_v1 = use_paged_attention_v1(nd, query.shape[1], md.max_decode_seq_len, num_kv_heads, head_size,
                                                block_size, query.dtype)
            fused_decode = ops.paged_attention_fused_qkv(
                output[:ndt], md.exp_sums, md.max_logits, md.tmp_out, rows, None, 0, None, None, md.slot_mapping[:ndt],
                key_cache, value_cache, query.shape[1], num_kv_heads, scale, md.block_table[:nd], md.seq_lens[:nd],
                block_size, md.max_decode_seq_len, not use_v1, md.partition_size)
    if key is not None and value is not None:
        # slot_mapping.size(0) is the number of real tokens (cache_kernels.cu:459-469)
        lo = ndt if fused_decode else 0
        if n > lo:
            ops.reshape_and_cache(key[lo:n], value[lo:n], key_cache, value_cache, md.slot_mapping[lo:n], kv_cache_dtype,
                                  k_scale, v_scale)
    if nd > 0 and not fused_decode:
        decode_attention(output[:ndt], md.exp_sums, md.max_logits, md.tmp_out, query[:ndt],
                         key_cache, value_cache, num_kv_heads, scale, md.block_table[:nd],
                         md.seq_lens[:nd], block_size, md.max_decode_seq_len, alibi_slopes,
                         kv_cache_dtype, k_scale, v_scale, md.use_v1, md.partition_size)
    if md.num_prefills > 0:
        ops.paged_prefill_attention(output[ndt:n], query[ndt:n], key_cache, value_cache,
                                    num_kv_heads, scale, md.block_table[nd:], md.seq_lens[nd:],
                                    md.prefill_query_start_loc, md.max_prefill_query_len, block_size,
                                    kv_cache_dtype, k_scale, v_scale, sliding_window, softcap, alibi_slopes)
    return output


# --------------------------------------------------------------------------- vLLM-facing
try:  # pragma: no cover - needs upstream vLLM
    from vllm.attention.backends.abstract import AttentionBackend, AttentionImpl, AttentionType
    from vllm.v1.attention.backends.utils import (AttentionCGSupport, AttentionMetadataBuilder,
                                                  CommonAttentionMetadata)

    class Mi355xPagedMetadataBuilder(AttentionMetadataBuilder[Mi355xPagedMetadata]):
        cudagraph_support = AttentionCGSupport.ALWAYS      # cf. triton_attn.py:69
        reorder_batch_threshold = 1                         # decode-first batches

        def __init__(self, kv_cache_spec, layer_names, vllm_config, device):
            super().__init__(kv_cache_spec, layer_names, vllm_config, device)
            mc = vllm_config.model_config
            self.num_heads = mc.get_num_attention_heads(vllm_config.parallel_config)
            self.num_kv_heads = mc.get_num_kv_heads(vllm_config.parallel_config)
            self.head_size = mc.get_head_size()
            self.dtype = mc.dtype
            self.block_size = kv_cache_spec.block_size
            self.max_model_len = mc.max_model_len
            # persistent split-KV workspaces (never re-allocated: captured graphs hold their addresses)
            self.workspace = DecodeWorkspace(vllm_config.scheduler_config.max_num_seqs, self.num_heads,
                                             self.head_size, self.max_model_len, self.dtype, device)
            self._capturing = False

        def build_for_cudagraph_capture(self, common_attn_metadata):
            # Full-graph decode: every host-side launch decision is frozen at max_model_len (see
            # build_metadata: fixed_decode_len), so a replay with longer sequences than the capture
            # batch runs the same grid / LDS size / kernel; seq_lens = 1 keeps the capture run cheap.
            self._capturing = True
            try:
                md = self.build(0, common_attn_metadata)
            finally:
                self._capturing = False
            md.seq_lens.fill_(1)                            # cf. triton_attn.py:93-99
            return md

        def build(self, common_prefix_len, common_attn_metadata: "CommonAttentionMetadata",
                  fast_build: bool = False):
            c = common_attn_metadata
            # decode-only batches may be replayed from a full graph: keep them on the frozen geometry
            decode_only = c.max_query_len == 1
            fixed = self.max_model_len if (self._capturing or (decode_only and self._full_graphs())) else None
            return build_metadata(c.query_start_loc, c.query_start_loc_cpu, c.seq_lens,
                                  c.seq_lens_cpu, c.block_table_tensor, c.slot_mapping,
                                  c.num_actual_tokens, c.max_query_len, c.max_seq_len,
                                  self.num_heads, self.head_size, self.dtype, self.workspace,
                                  self.num_kv_heads, self.block_size, fixed)

        def _full_graphs(self) -> bool:
            cc = getattr(self.vllm_config, "compilation_config", None)
            mode = getattr(cc, "cudagraph_mode", None)
            return bool(mode is not None and getattr(mode, "has_full_cudagraphs", lambda: False)())

    class Mi355xPagedAttentionImpl(AttentionImpl):
        def __init__(self, num_heads, head_size, scale, num_kv_heads, alibi_slopes, sliding_window,
                     kv_cache_dtype, logits_soft_cap=None, attn_type=AttentionType.DECODER,
                     kv_sharing_target_layer_name=None, **kwargs):
            if kv_cache_dtype not in ("auto", "fp8", "fp8_e4m3"):
                raise ValueError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
            self.kv_cache_dtype = kv_cache_dtype
            if sliding_window is not None or logits_soft_cap:
                raise NotImplementedError("sliding window / soft-cap are out of scope")
            if attn_type != AttentionType.DECODER:
                raise NotImplementedError("only decoder self-attention is supported")
            self.num_heads, self.head_size, self.scale = num_heads, head_size, float(scale)
            self.num_kv_heads = num_kv_heads
            self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32)
                                 if alibi_slopes is not None else None)

        def forward(self, layer, query, key, value, kv_cache, attn_metadata, output=None,
                    output_scale=None, output_block_scale=None):
            assert output is not None, "Output tensor must be provided."
            if attn_metadata is None:                      # profiling run (flash_attn.py:630-632)
                return output.fill_(0)
            slopes = self.alibi_slopes
            if slopes is not None and slopes.device != query.device:
                slopes = self.alibi_slopes = slopes.to(query.device)
            q3 = query.view(-1, self.num_heads, self.head_size)
            k3 = key.view(-1, self.num_kv_heads, self.head_size) if key is not None else None
            v3 = value.view(-1, self.num_kv_heads, self.head_size) if value is not None else None
            fp8 = self.kv_cache_dtype != "auto"
            if fp8 and kv_cache.dtype != torch.uint8:
                kv_cache = kv_cache.view(torch.uint8)
            paged_attention_forward(q3, k3, v3, kv_cache, attn_metadata,
                                    output.view(-1, self.num_heads, self.head_size),
                                    self.num_kv_heads, self.scale, slopes, self.kv_cache_dtype,
                                    layer._k_scale if fp8 else None, layer._v_scale if fp8 else None)
            return output

    class Mi355xPagedAttentionBackend(AttentionBackend):
        accept_output_buffer: bool = True

        @classmethod
        def get_supported_dtypes(cls):
            return [torch.float16, torch.bfloat16, torch.float32]

        @staticmethod
        def get_supported_head_sizes():
            return [32, 64, 80, 96, 112, 120, 128, 192, 256]   # paged_attention_v1.cu:90-124

        @staticmethod
        def get_name() -> str:
            return "MI355X_PAGED_ATTN"

        @staticmethod
        def get_impl_cls():
            return Mi355xPagedAttentionImpl

        @staticmethod
        def get_metadata_cls():
            return Mi355xPagedMetadata

        @staticmethod
        def get_builder_cls():
            return Mi355xPagedMetadataBuilder

        @staticmethod
        def get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size,
                               cache_dtype_str: str = "auto"):
            if block_size not in (8, 16, 32):
                raise ValueError("Block size must be 8, 16 or 32.")
            return kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

        @staticmethod
        def use_cascade_attention(*args, **kwargs) -> bool:
            return False                                    # platform.py:219-221
except Exception:  # noqa: BLE001 - vLLM absent: only the core above is available
    pass
