"""Synthetic Llama-shaped caller of the hot path — the measurement harness's model.

Stands in for the upstream-vLLM call stack that drives the reference's kernels
(LlamaDecoderLayer.forward -> RMSNorm / QKV linear / rotary / cache write /
attention / o_proj / RMSNorm / gate_up / SiluAndMul / down): the same sequence of ops, on
synthetic weights of the named architecture (no checkpoints exist offline),
with greedy sampling.  Every op on the hot path goes through the C-ABI
(vllm_metax_amd._custom_ops); torch is used for allocation, the embedding gather and the
un-quantised lm_head GEMM (library GEMM).

Decode steps are captured into a HIP graph (hipGraph via torch.cuda.CUDAGraph): all per-step
index arithmetic (positions, slot mapping, sequence lengths) happens on the device inside
the graph, so a replay is one host call.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import List, Optional

import torch

from . import _custom_ops as ops
from .attention.backend import (decode_attention, decode_attention_fused, decode_partition_size,
                                use_paged_attention_v1)


@dataclasses.dataclass
class ModelConfig:
    name: str = "llama-3-8b"
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    ffn: int = 14336
    vocab: int = 128256
    rope_theta: float = 500000.0
    eps: float = 1e-5
    quant: str = "awq"          # "awq" | "gptq" | "fp8" | "int8" | "none"
    group_size: int = 128
    tp: int = 1                 # tensor-parallel degree (heads / ffn sharded, all-reduce after o/down)
    tp_rank: int = 0            # this process's shard
    kv_cache_dtype: str = "auto"   # "auto" | "fp8" (e4m3 KV cache) | "fp8_e5m2"

    @staticmethod
    def llama3_8b(quant="awq", tp=1):
        return ModelConfig(quant=quant, tp=tp)

    @staticmethod
    def llama3_70b(quant="fp8", tp=8):
        return ModelConfig(name="llama-3-70b", hidden=8192, layers=80, heads=64, kv_heads=8,
                           ffn=28672, quant=quant, tp=tp)

    @staticmethod
    def qwen2_72b(quant="gptq", tp=8):
        return ModelConfig(name="qwen2-72b", hidden=8192, layers=80, heads=64, kv_heads=8,
                           ffn=29696, vocab=152064, rope_theta=1e6, eps=1e-6, quant=quant, tp=tp)

    @staticmethod
    def llama_geometry(quant="awq", layers=2, vocab=8192):
        """A few layers at Llama-3-8B WIDTH (hidden 4096, 32 / 8 heads x 128, FFN 14336): what the fused decode /
        image paths and the split-K GEMMs need to be selected — for end-to-end tests, not for the bench."""
        return ModelConfig(name=f"llama-geometry-{layers}l", layers=layers, vocab=vocab, quant=quant)

    @staticmethod
    def tiny(quant="awq"):
        return ModelConfig(name="tiny", hidden=512, layers=2, heads=8, kv_heads=2, head_dim=64,
                           ffn=1024, vocab=1024, quant=quant)


class QuantAct:
    """An activation matrix already quantised per token (fp8 bytes + one float32 scale per row): what the fused
    norm + quant op hands to the W8A8 linear (rms_norm_dynamic_per_token_quant)."""

    def __init__(self, data: torch.Tensor, scales: torch.Tensor, dtype):
        self.data, self.scales, self.dtype = data, scales, dtype
        self.shape, self.device = data.shape, data.device


class QLinear:
    """One (column- or row-parallel shard of a) linear layer with synthetic weights.

    The FULL [k, n] layer is always generated (same generator consumption on every rank and for
    every tp), then the shard is cut out: `cols` = list of (start, length) column segments kept
    (column-parallel: qkv, gate_up), `rows` = (start, length) of the K range kept (row-parallel: o,
    down) — so a TP=N model is a true sharding of the TP=1 model and produces the same tokens.
    Segment bounds are multiples of 8 columns (packed words) / of the group size (rows)."""

    def __init__(self, k: int, n: int, cfg: ModelConfig, dtype, device, gen, cols=None, rows=None):
        self.quant, self.group = cfg.quant, cfg.group_size
        self.dtype = dtype
        g = self.group

        def ccut(t, unit=1):          # keep the column segments of the last dim (in units of `unit` cols)
            if cols is None:
                return t
            return torch.cat([t[..., a // unit:(a + l) // unit] for a, l in cols], dim=-1).contiguous()

        def rcut(t, unit=1):          # keep the row range of dim 0 (in units of `unit` rows)
            if rows is None:
                return t
            return t[rows[0] // unit:(rows[0] + rows[1]) // unit].contiguous()

        if cfg.quant == "awq":
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k, n // 8), dtype=torch.int32, device=device,
                               generator=gen)
            qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // g, n // 8), dtype=torch.int32, device=device,
                               generator=gen)
            sc = (torch.rand(k // g, n, device=device, generator=gen) * 4e-3 + 1e-3).to(dtype)
            self.qweight = ops.awq_to_gptq_4bit(rcut(ccut(qw, 8)))        # exllama layout
            self.qzeros = rcut(ccut(qz, 8), g)
            self.scales = rcut(ccut(sc), g)
        elif cfg.quant == "gptq":
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), dtype=torch.int32, device=device,
                               generator=gen)
            sc = (torch.rand(k // g, n, device=device, generator=gen) * 4e-3 + 1e-3).to(dtype)
            self.qweight = rcut(ccut(qw), 8)
            ops.gptq_shuffle(self.qweight, torch.empty(0, dtype=torch.int32), 4)
            self.scales = rcut(ccut(sc), g)
            self.qzeros = torch.full((self.scales.shape[0], self.scales.shape[1] // 8), 0x77777777,
                                     dtype=torch.int32, device=device)    # symmetric (stored zero 7)
            self.g_idx = torch.empty(0, dtype=torch.int32, device=device)
        elif cfg.quant == "fp8":
            w = torch.randn(n, k, device=device, generator=gen).clamp_(-448, 448)      # [N, K]
            ws = (torch.rand(1, n, device=device, generator=gen) * 4e-3 + 1e-3)
            self.weight = rcut(ccut(w.t())).t().contiguous().to(torch.float8_e4m3fn).t()   # [K, N] column-major
            self.w_scale = ccut(ws)
        elif cfg.quant == "int8":
            # W8A8 (compressed-tensors style): per-channel int8 weights, dynamic per-token activations
            w = torch.randint(-127, 128, (n, k), dtype=torch.int32, device=device, generator=gen)
            ws = (torch.rand(1, n, device=device, generator=gen) * 4e-5 + 1e-5)
            self.weight = rcut(ccut(w.t())).t().contiguous().to(torch.int8).t()          # [K, N] column-major
            self.w_scale = ccut(ws)
        else:
            w = (torch.randn(k, n, device=device, generator=gen) * 0.02).to(dtype)
            self.weight = rcut(ccut(w))
        self.k = rows[1] if rows is not None else k
        self.n = sum(l for _, l in cols) if cols is not None else n
        self._ws = {}

    _shared_ws = {}   # device -> fp32 split-K scratch shared by every layer (stream-ordered use)
    _retired = []     # outgrown scratch buffers, kept alive for graphs captured before the growth
    packed_silu = os.environ.get("MI355X_PACKED_SILU", "1") != "0"   # A/B switch for the bench
    # keep the prefill GEMM's weight operand image (dequantised once at load time, n*k*2 bytes per
    # layer: 14 GB for Llama-3-8B of the 288 GB) instead of re-deriving it from the int4 words on every
    # prefill call; decode still streams the int4 words.  MI355X_PREPACK=0 switches it off.
    prepack = os.environ.get("MI355X_PREPACK", "1") != "0"

    def image(self):
        """The prefill GEMM's weight image (None when not applicable), built on first use."""
        if not self.prepack or self.quant not in ("awq", "gptq") or self.dtype == torch.float32 \
                or self.n % 64 or self.k % 32:
            return None
        if getattr(self, "_image", None) is None:
            self._image = ops.w4a16_prepack(self.qweight, self.qzeros, self.scales, self.quant == "gptq")
        return self._image

    def _workspace(self, m: int, device):
        """fp32 scratch for the split-K partial slabs of the decode GEMM: up to 8 slabs of
        [m, n] (the reference passes a fresh torch.zeros temp_space per call, awq.py:140-147;
        here one buffer per device is reused, its contents are scratch)."""
        need = 8 * m * self.n
        ws = QLinear._shared_ws.get(device)
        if ws is None or ws.numel() < need:
            if ws is not None:
                QLinear._retired.append(ws)      # a captured decode graph may still write to it
            ws = torch.zeros(need, dtype=torch.float32, device=device)
            QLinear._shared_ws[device] = ws
        return ws

    def __call__(self, x) -> torch.Tensor:
        m = x.shape[0]
        # (with the weights' image at hand the image GEMM — K split for the narrow projections — beats the 128-row
        #  stripe passes from 384 rows on; without it 1024, where deriving the image per call pays)
        if m >= ops.W4_PREPACKED_MIN_M and self.image() is not None:
            return ops.w4a16_gemm_prepacked(x, self.image(), self.n, self.k)
        if isinstance(x, ops.PackedOperand):     # prefill: activations already re-tiled by the producer
            return ops.awq_gemm_packed_a(x, self.qweight, self.qzeros, self.scales)
        if self.quant == "awq":
            ws = self._workspace(m, x.device) if m <= 64 else torch.empty(0)
            return ops.awq_gemm(x, self.qweight, self.qzeros, self.scales, 8, ws,
                                x.dtype == torch.bfloat16)
        if self.quant == "gptq":
            ws = self._workspace(m, x.device) if m <= 64 else torch.empty(0)
            return ops.gptq_gemm(x, self.qweight, self.qzeros, self.scales, self.g_idx, True, 4,
                                 self.group, torch.empty(0), ws, x.dtype == torch.bfloat16)
        if self.quant == "fp8":
            if isinstance(x, QuantAct):      # quantised by the producer (fused norm + quant)
                xq, xs, odt = x.data, x.scales, x.dtype
            else:
                xq = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
                xs = torch.empty(m, 1, dtype=torch.float32, device=x.device)
                ops.dynamic_per_token_scaled_fp8_quant(xq, x, xs, None)
                odt = x.dtype
            out = torch.empty(m, self.n, dtype=odt, device=xq.device)
            self._scaled_mm(out, xq, xs)
            return out
        if self.quant == "int8":
            xq, xs, _ = ops.scaled_int8_quant(x)
            out = torch.empty(m, self.n, dtype=x.dtype, device=x.device)
            self._scaled_mm(out, xq, xs)
            return out
        return torch.matmul(x, self.weight)

    def _scaled_mm(self, out, xq, xs):
        """cutlass_scaled_mm on this layer's 8-bit weights; prefill-sized calls (m > 320) on weights the GEMM re-tiles
        per call (int8; fp8 with k % 128 != 0) multiply by the weights' operand image built once at first use
        (mi355x_scaled_mm_prepack: n * k more bytes per layer) — MI355X_PREPACK=0, a shape without an image or fp8
        weights read in place (scaled_mm_prepack returns None): the plain op."""
        if xq.shape[0] > 320 and self.prepack and hasattr(ops, "scaled_mm_prepack"):
            if getattr(self, "_w8_image", None) is None:
                img = ops.scaled_mm_prepack(self.weight)
                self._w8_image = img if img is not None else False
            if self._w8_image is not False:
                ops.scaled_mm_prepacked(out, xq, self._w8_image, self.n, xs, self.w_scale, None)
                return
        ops.cutlass_scaled_mm(out, xq, self.weight, xs, self.w_scale, None)

    def silu_mul(self, x: torch.Tensor):
        """silu_and_mul(self(x)) in one launch where the library supports it (AWQ, M >= 1024), else None."""
        if x.shape[0] >= ops.W4_PREPACKED_MIN_M and self.n % 256 == 0 and self.image() is not None:
            return ops.w4a16_gemm_prepacked(x, self.image(), self.n, self.k, silu=True,
                                            out_image=self.packed_silu)
        if self.quant == "awq":
            if x.shape[0] >= 1024 and self.packed_silu:
                # the result goes straight into the operand image of the next (down_proj) GEMM
                out = ops.awq_gemm_silu_mul_packed(x, self.qweight, self.qzeros, self.scales)
                if out is not None:
                    return out
            return ops.awq_gemm_silu_mul(x, self.qweight, self.qzeros, self.scales)
        return None

    # fp8 decode: leave a K split of the scaled GEMM to its consumer (mi355x_scaled_mm_fp8_deferred) instead of a
    # finish launch per GEMM; "0": every GEMM finishes itself (A/B switch for the bench)
    fp8_defer = os.environ.get("MI355X_FP8_DEFER", "1") != "0"
    # ... also by the per-token consumers (silu + quant of gate_up, norm + quant of o / down at tp == 1): built, bit-
    # identical, and measured SLOWER than the GEMM's own finish launch (one workgroup per token reads the fp32 slabs
    # with 64 CUs; the finish kernel spreads them over the chip: gate_up 8B 40.4 vs 46.9 us, 70B rank 24.4 vs 25.8,
    # o / down +-0.5 us, scripts/bench_fp8_consumers.py) -> off; "1" switches it on (tests do)
    fp8_defer_rows = os.environ.get("MI355X_FP8_DEFER_ROWS", "0") != "0"

    def deferred(self, x, allow_scaled: bool = False):
        """(out, slabs, sk, scales): like __call__, but a decode-sized GEMM may leave its split-K slabs unreduced
        for its consumer (sk == 0: `out` is final).  AWQ: scales is None, the consumer rounds the slab sum
        (ops.fused_add_rms_norm_slabs, qkv_rope_cache, paged_attention_fused_qkv).  fp8 (`allow_scaled`: the
        caller's consumer takes them): scales = (activation scales [m, 1], weight scales [1, n]) that the consumer
        still has to apply — the bits of the GEMM's own finish launch."""
        m = x.shape[0]
        if self.quant == "awq" and m <= 64:
            ws = self._workspace(m, x.device)
            out, sk = ops.awq_gemm_deferred(x, self.qweight, self.qzeros, self.scales, ws)
            return out, ws, sk, None
        if self.quant == "fp8" and m <= 64 and allow_scaled and self.fp8_defer:
            if isinstance(x, QuantAct):
                xq, xs, odt = x.data, x.scales, x.dtype
            else:
                xq = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
                xs = torch.empty(m, 1, dtype=torch.float32, device=x.device)
                ops.dynamic_per_token_scaled_fp8_quant(xq, x, xs, None)
                odt = x.dtype
            out = torch.empty(m, self.n, dtype=odt, device=xq.device)
            ws = self._workspace(m, xq.device)
            sk = ops.scaled_mm_fp8_deferred(out, xq, self.weight, xs, self.w_scale, ws)
            return out, ws, sk, ((xs, self.w_scale) if sk > 0 else None)
        return self(x), None, 0, None

    def weight_bytes(self) -> int:
        if self.quant in ("awq", "gptq"):
            return self.k * self.n // 2 + (self.k // self.group) * self.n * 2 \
                + (self.k // self.group) * self.n // 2
        if self.quant in ("fp8", "int8"):
            return self.k * self.n + self.n * 4
        return self.k * self.n * 2


class Layer:
    """One decoder layer, sharded Megatron-style for tp > 1: qkv / gate_up column-
    parallel (rank r owns q heads [r H/tp, ..), kv heads [r KVH/tp, ..), ffn columns [r F/tp, ..)),
    o_proj / down_proj row-parallel over the matching K range (their outputs are all-reduced)."""

    def __init__(self, cfg: ModelConfig, dtype, device, gen):
        h, d = cfg.hidden, cfg.head_dim
        tp, r = cfg.tp, cfg.tp_rank
        if cfg.heads % tp or cfg.kv_heads % tp or cfg.ffn % tp or (cfg.ffn // tp) % cfg.group_size:
            raise ValueError(f"tp={tp} must divide heads {cfg.heads}, kv_heads {cfg.kv_heads} and ffn {cfg.ffn} "
                             f"(ffn/tp a multiple of the group size)")
        self.q_heads = cfg.heads // tp
        self.kv_heads = cfg.kv_heads // tp
        self.q_size, self.kv_size = self.q_heads * d, self.kv_heads * d
        ffn = cfg.ffn // tp
        Q, KV = cfg.heads * d, cfg.kv_heads * d
        if tp == 1:
            qkv_cols = gu_cols = o_rows = down_rows = None
        else:
            qkv_cols = [(r * self.q_size, self.q_size), (Q + r * self.kv_size, self.kv_size),
                        (Q + KV + r * self.kv_size, self.kv_size)]
            gu_cols = [(r * ffn, ffn), (cfg.ffn + r * ffn, ffn)]
            o_rows = (r * self.q_size, self.q_size)
            down_rows = (r * ffn, ffn)
        self.qkv = QLinear(h, Q + 2 * KV, cfg, dtype, device, gen, cols=qkv_cols)
        self.o = QLinear(Q, h, cfg, dtype, device, gen, rows=o_rows)
        self.gate_up = QLinear(h, 2 * cfg.ffn, cfg, dtype, device, gen, cols=gu_cols)
        self.down = QLinear(cfg.ffn, h, cfg, dtype, device, gen, rows=down_rows)
        self.ln1 = (torch.rand(h, device=device, generator=gen) * 0.2 + 0.9).to(dtype)
        self.ln2 = (torch.rand(h, device=device, generator=gen) * 0.2 + 0.9).to(dtype)
        self.ffn = ffn


class HotPathModel:
    """Weights + paged KV cache + the forward passes (prefill chunk / decode step)."""

    BLOCK = 16

    def __init__(self, cfg: ModelConfig, max_seqs: int, max_len: int, device="cuda:0",
                 dtype=torch.bfloat16, seed: int = 0, tp_group=None):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.tp_group = tp_group
        gen = torch.Generator(device=self.device).manual_seed(seed)
        self.layers: List[Layer] = [Layer(cfg, dtype, self.device, gen) for _ in range(cfg.layers)]
        self.embed = (torch.randn(cfg.vocab, cfg.hidden, device=self.device, generator=gen)
                      * 0.02).to(dtype)
        vshard = cfg.vocab // cfg.tp            # vocab-parallel lm_head: a column slice of the full matrix
        lm_full = (torch.randn(cfg.hidden, cfg.vocab, device=self.device, generator=gen) * 0.02).to(dtype)
        self.lm_head = lm_full[:, cfg.tp_rank * vshard:(cfg.tp_rank + 1) * vshard].contiguous()
        del lm_full
        self.final_norm = torch.ones(cfg.hidden, dtype=dtype, device=self.device)
        d = cfg.head_dim
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, device=self.device).float() / d))
        t = torch.arange(max_len + 1, device=self.device).float()
        fr = torch.outer(t, inv_freq)
        self.cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).to(dtype)
        # paged KV cache, x-split layout: one (K, V) pair per layer
        self.max_seqs, self.max_len = max_seqs, max_len
        self.blocks_per_seq = (max_len + self.BLOCK - 1) // self.BLOCK
        nb = max_seqs * self.blocks_per_seq
        kvh = self.layers[0].kv_heads
        self.kv_dtype = cfg.kv_cache_dtype
        cache_dt = dtype if self.kv_dtype == "auto" else torch.uint8      # e4m3 bytes, x = 16
        x = 16 // torch.tensor([], dtype=cache_dt).element_size()
        self.k_cache = [torch.zeros(nb, kvh, d // x, self.BLOCK, x, dtype=cache_dt, device=self.device)
                        for _ in range(cfg.layers)]
        self.v_cache = [torch.zeros(nb, kvh, d, self.BLOCK, dtype=cache_dt, device=self.device)
                        for _ in range(cfg.layers)]
        # one float32 scale per tensor, as the schema's k_scale / v_scale (1.0: the reference tests' value)
        self.k_scale = torch.ones(1, dtype=torch.float32, device=self.device) if self.kv_dtype != "auto" else None
        self.v_scale = torch.ones(1, dtype=torch.float32, device=self.device) if self.kv_dtype != "auto" else None
        # block tables: a fixed random permutation of the pool (as the reference's tests do)
        perm = torch.randperm(nb, device=self.device, generator=gen).to(torch.int32)
        self.block_tables = perm.reshape(max_seqs, self.blocks_per_seq).contiguous()
        self.scale = 1.0 / math.sqrt(d)
        self._graph = None
        self.graph_error = None
        self.collectives_always = False
        # prefill through the decode fusion kernel (rotary + cache write in one launch): measured SLOWER
        # at 8192 tokens (96 us vs 29.9 + 19.1 us: its V scatter is per token, reshape_and_cache's is a
        # 16-token tile transposed through LDS) -> off; bench 5585 vs 5639 tokens/s (profiles/r02 notes)
        self.fuse_prefill_rope = os.environ.get("MI355X_PREFILL_ROPE_FUSION", "0") != "0"
        # greedy sampling + position / slot bookkeeping between two decode steps in one launch
        # (mi355x_greedy_advance) instead of ~13 torch launches (~100 us per step); "0": the torch ops
        self.fuse_greedy = os.environ.get("MI355X_FUSE_GREEDY", "1") != "0"
        # decode: slab sum + rotary + cache write of the new token inside the attention launch
        # (mi355x_paged_attention_fused_qkv) instead of a qkv_rope_cache launch in front of it; "0": two launches
        self.fuse_attn_qkv = os.environ.get("MI355X_FUSE_ATTN_QKV", "1") != "0"
        # prefill: the norms in front of the qkv / gate_up GEMMs write the GEMM's activation operand image
        # (mi355x_*rms_norm_image) instead of a row-major tensor the GEMM re-tiles; "0": row-major
        self.norm_image = os.environ.get("MI355X_NORM_IMAGE", "1") != "0"
        # fp8 weights: the norms in front of qkv / gate_up quantise their output per token themselves
        # (rms_norm_dynamic_per_token_quant, the reference's fused op csrc/quantization/fused_kernels/): one launch
        # and one pass over the activations less per GEMM; "0": rms_norm, then dynamic_per_token_scaled_fp8_quant
        self.fuse_norm_quant = os.environ.get("MI355X_FUSE_NORM_QUANT", "1") != "0"

    # ---------------------------------------------------------------- helpers
    def _collectives(self) -> bool:
        # `collectives_always`: rehearsal switch — issue the collectives on a 1-rank group too, so that
        # RCCL calls inside a captured decode graph can be exercised on a single GPU
        return self.tp_group is not None and (self.cfg.tp > 1 or self.collectives_always)

    def _all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self._collectives():
            torch.distributed.all_reduce(x, group=self.tp_group)
        return x

    def _norm_quant(self, x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                    pending=(None, 0, None)) -> QuantAct:
        """rms_norm (fused add when `residual` is given: it then receives x + residual) + dynamic per-token fp8
        quantisation in one launch.  `pending` = (slabs, sk, scales): x is still the unreduced output of an fp8
        GEMM — the norm adds the slabs and applies the GEMM's scales on its way in."""
        q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
        sc = torch.empty(x.shape[0], 1, dtype=torch.float32, device=x.device)
        if pending[1] > 0:
            ops.rms_norm_dynamic_per_token_quant_slabs(q, pending[0], pending[1], pending[2][0], pending[2][1], weight,
                                                       sc, self.cfg.eps, None, residual)
        else:
            ops.rms_norm_dynamic_per_token_quant(q, x.contiguous(), weight, sc, self.cfg.eps, None, residual)
        return QuantAct(q, sc, x.dtype)

    def _fused_qkv_geometry(self, L) -> bool:
        """Shapes mi355x_paged_attention_fused_qkv takes (include/mi355x_hotpath.h): decided up front, because an
        fp8 qkv GEMM may only leave its slabs unreduced when that launch will consume them."""
        return self.cfg.head_dim == 128 and self.BLOCK == 16 and self.kv_dtype == "auto" \
            and self.dtype in (torch.bfloat16, torch.float16) and L.q_heads // L.kv_heads in (4, 8) \
            and L.q_heads % L.kv_heads == 0

    def _slots(self, seq_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        blk = self.block_tables[seq_ids.long(), (positions // self.BLOCK).long()].long()
        return blk * self.BLOCK + (positions % self.BLOCK)

    def _layer(self, i: int, x: torch.Tensor, residual: Optional[torch.Tensor],
               positions: torch.Tensor, slots: torch.Tensor, attn_fn, pending=(None, 0, None),
               defer: bool = False, attn_fused_fn=None):
        """`pending` = (slabs, sk, scales) when x is still the unreduced output of the previous layer's
        down_proj (tp == 1 decode): the fused norm adds the slabs itself (scales: an fp8 GEMM's, see
        QLinear.deferred)."""
        L = self.layers[i]
        cfg = self.cfg
        if len(pending) == 2:
            pending = (pending[0], pending[1], None)
        fuse = defer and self.cfg.tp == 1   # with tp > 1 the all-reduce sits between GEMM and norm
        use_img = (not defer) and self.norm_image and x.shape[0] >= ops.W4_PREPACKED_MIN_M and pending[1] == 0 \
            and L.qkv.image() is not None and L.gate_up.image() is not None
        nq = cfg.quant == "fp8" and self.fuse_norm_quant and x.dtype != torch.float32 \
            and (pending[1] == 0 or pending[2] is not None)
        # fp8 decode: the K splits of the qkv / gate_up (and at tp == 1 o / down) GEMMs are reduced by their consumers
        w8_defer = defer and nq and QLinear.fp8_defer and hasattr(ops, "scaled_mm_fp8_deferred")
        if residual is None:
            residual = x.clone()
            h = ops.rms_norm_image(x, L.ln1, cfg.eps) if use_img else None
            if h is None and nq:
                h = self._norm_quant(x, None, L.ln1)
            if h is None:
                h = torch.empty_like(x)
                ops.rms_norm(h, x, L.ln1, cfg.eps)
        else:
            h = ops.fused_add_rms_norm_image(x, residual, L.ln1, cfg.eps) if use_img else None
            if h is None and nq:
                h = self._norm_quant(x, residual, L.ln1, pending)
            if h is None:
                ops.fused_add_rms_norm_slabs(x, residual, L.ln1, pending[0], pending[1], cfg.eps)
                h = x
        attn = None
        if defer and x.dtype != torch.float32 and self.kv_dtype == "auto":
            # decode: slab sum + rotary + cache write in one launch (column-parallel GEMM: no
            # collective between it and the rotary, so the fusion also holds under tp)
            fq_w8 = w8_defer and attn_fused_fn is not None and self._fused_qkv_geometry(L)
            qkv, slabs, sk, scl = L.qkv.deferred(h, allow_scaled=fq_w8)
            attn = attn_fused_fn(i, qkv, slabs, sk, scl, fq_w8) if attn_fused_fn is not None else None
            if attn is None:
                if scl is not None:
                    raise RuntimeError("fp8 qkv slabs left without a consumer (paged_attention_fused_qkv refused)")
                ops.qkv_rope_cache(qkv, slabs, sk, positions, self.cos_sin, self.k_cache[i],
                                   self.v_cache[i], slots, L.q_heads, L.kv_heads, cfg.head_dim)
            q = qkv[:, :L.q_size]
        elif self.fuse_prefill_rope and x.dtype != torch.float32 and self.kv_dtype == "auto":
            # prefill: rotary + cache write in one launch as well (the decode kernel, no slabs)
            qkv = L.qkv(h)
            ops.qkv_rope_cache(qkv, None, 0, positions, self.cos_sin, self.k_cache[i], self.v_cache[i],
                               slots, L.q_heads, L.kv_heads, cfg.head_dim)
            q = qkv[:, :L.q_size]
        else:
            qkv = L.qkv(h)
            q = qkv[:, :L.q_size]
            k = qkv[:, L.q_size:L.q_size + L.kv_size]
            v = qkv[:, L.q_size + L.kv_size:]
            cached = False
            if getattr(self, "_pf_rope_q_in_attn", False) and not defer:
                # prefill: the attention kernel rotates the query rows while it loads them; the key rows are
                # rotated on their way into the cache (one launch), or in place when that form does not apply
                cached = self.kv_dtype == "auto" and ops.rotary_reshape_and_cache(
                    positions, k.view(-1, L.kv_heads, cfg.head_dim), v.view(-1, L.kv_heads, cfg.head_dim),
                    self.k_cache[i], self.v_cache[i], slots, self.cos_sin)
                if not cached:
                    ops.rotary_embedding(positions, k, None, cfg.head_dim, self.cos_sin, True)
            else:
                ops.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin, True)
            if not cached:
                ops.reshape_and_cache(k.view(-1, L.kv_heads, cfg.head_dim), v.view(-1, L.kv_heads, cfg.head_dim),
                                      self.k_cache[i], self.v_cache[i], slots, self.kv_dtype, self.k_scale,
                                      self.v_scale)
        if attn is None:
            attn = attn_fn(i, q.view(-1, L.q_heads, cfg.head_dim))
        if fuse and not nq:
            o, slabs, sk, _ = L.o.deferred(attn.view(-1, L.q_size))
            ops.fused_add_rms_norm_slabs(o, residual, L.ln2, slabs, sk, cfg.eps)
        elif fuse and w8_defer and QLinear.fp8_defer_rows:
            o, slabs, sk, scl = L.o.deferred(attn if isinstance(attn, QuantAct) else attn.view(-1, L.q_size),
                                             allow_scaled=True)
            o = self._norm_quant(o, residual, L.ln2, (slabs, sk, scl))
        else:
            o = self._all_reduce(L.o(attn if isinstance(attn, (ops.PackedOperand, QuantAct))
                                     else attn.view(-1, L.q_size)))
            o_img = ops.fused_add_rms_norm_image(o, residual, L.ln2, cfg.eps) if use_img else None
            if o_img is not None:
                o = o_img
            elif nq:
                o = self._norm_quant(o, residual, L.ln2)
            else:
                ops.fused_add_rms_norm(o, residual, L.ln2, cfg.eps)
        act = L.gate_up.silu_mul(o)            # prefill-sized AWQ: fused into the GEMM epilogue
        if act is None:
            # (fp8 decode: the K split of gate_up is reduced by the silu + quant launch)
            gu, slabs, sk, scl = L.gate_up.deferred(o, allow_scaled=L.ffn % 8 == 0 and L.ffn <= 16384) \
                if w8_defer and QLinear.fp8_defer_rows else (L.gate_up(o), None, 0, None)
            if nq:                             # fp8: silu_and_mul + per-token quant of down_proj's input in one launch
                if scl is not None:
                    qa = ops.silu_and_mul_per_token_quant_slabs(slabs, sk, scl[0], scl[1], gu.shape[0], L.ffn, gu.dtype)
                    if qa is None:
                        raise RuntimeError("fp8 gate_up slabs left without a consumer")
                else:
                    qa = ops.silu_and_mul_per_token_quant(gu)
                if qa is not None:
                    act = QuantAct(qa[0], qa[1], gu.dtype)
        if act is None:
            act = torch.empty(gu.shape[0], L.ffn, dtype=gu.dtype, device=gu.device)
            ops.silu_and_mul(act, gu)
        if fuse and (cfg.quant != "fp8" or (w8_defer and QLinear.fp8_defer_rows and i + 1 < cfg.layers)):
            # (fp8: the next layer's norm + quant applies the scales; the last layer finishes itself for _logits)
            out, slabs, sk, scl = L.down.deferred(act, allow_scaled=True)
            return out, residual, (slabs, sk, scl)
        out = self._all_reduce(L.down(act))
        return out, residual, (None, 0, None)

    def _logits_argmax(self, x: torch.Tensor, residual: torch.Tensor, pending=(None, 0, None)) -> torch.Tensor:
        return self._logits(x, residual, pending).argmax(dim=-1)

    def _logits(self, x: torch.Tensor, residual: torch.Tensor, pending=(None, 0, None)) -> torch.Tensor:
        ops.fused_add_rms_norm_slabs(x, residual, self.final_norm, pending[0], pending[1], self.cfg.eps)
        logits = torch.matmul(x, self.lm_head)
        if self._collectives():
            # (a 1-rank rehearsal group of a tp > 1 shard gathers only its own vocabulary slice)
            ws = torch.distributed.get_world_size(self.tp_group)
            gathered = torch.empty(ws * logits.shape[0], logits.shape[1], dtype=logits.dtype, device=logits.device)
            torch.distributed.all_gather_into_tensor(gathered, logits.contiguous(), group=self.tp_group)
            # [ws * M, V/tp] -> [M, V]: rank r owns vocabulary slice r
            logits = gathered.view(ws, logits.shape[0], logits.shape[1]).permute(1, 0, 2) \
                             .reshape(logits.shape[0], -1)
        return logits

    # ---------------------------------------------------------------- prefill
    def prefill(self, token_ids: torch.Tensor, seq_ids: List[int], context_len: int = 0):
        """One prefill chunk: `token_ids` [n_seqs, q_len] new tokens for the sequences `seq_ids`,
        each with `context_len` tokens already cached.  Returns the next token per sequence."""
        n, q_len = token_ids.shape
        dev = self.device
        sid = torch.tensor(seq_ids, dtype=torch.int64, device=dev)
        pos = (torch.arange(q_len, device=dev) + context_len).repeat(n)
        seq_of_tok = sid.repeat_interleave(q_len)
        slots = self._slots(seq_of_tok, pos)
        cu = (torch.arange(n + 1, device=dev, dtype=torch.int32) * q_len)
        seq_lens = torch.full((n,), context_len + q_len, dtype=torch.int32, device=dev)
        bt = self.block_tables[sid]
        x = self.embed[token_ids.reshape(-1)]
        residual = None

        # q rotary inside the attention kernel (with the image output): decided once per chunk
        self._pf_rope_q_in_attn = bool(self.norm_image and n * q_len >= ops.W4_PREPACKED_MIN_M and self.cfg.head_dim == 128
                                       and self.BLOCK == 16 and self.dtype != torch.float32
                                       and not self.fuse_prefill_rope
                                       and all(L.o.image() is not None for L in self.layers[:1]))

        def attn_fn(i, q3):
            rope = self._pf_rope_q_in_attn
            if self.norm_image and q3.shape[0] >= ops.W4_PREPACKED_MIN_M and self.layers[i].o.image() is not None:
                # the attention output goes straight into o_proj's activation operand image
                img = ops.paged_prefill_attention_image(q3, self.k_cache[i], self.v_cache[i],
                                                        self.layers[i].kv_heads, self.scale, bt, seq_lens, cu,
                                                        q_len, self.BLOCK, self.kv_dtype, self.k_scale, self.v_scale,
                                                        pos if rope else None, self.cos_sin if rope else None)
                if img is not None:
                    return img
            if rope:      # the image form did not apply after all: rotate q now
                ops.rotary_embedding(pos, q3.view(q3.shape[0], -1), None, self.cfg.head_dim, self.cos_sin, True)
            out = torch.empty_like(q3)
            ops.paged_prefill_attention(out, q3, self.k_cache[i], self.v_cache[i],
                                        self.layers[i].kv_heads, self.scale, bt, seq_lens, cu,
                                        q_len, self.BLOCK, self.kv_dtype, self.k_scale, self.v_scale)
            return out

        for i in range(self.cfg.layers):
            x, residual, _ = self._layer(i, x, residual, pos, slots, attn_fn)
        last = (cu[1:] - 1).long()
        self.last_prefill_logits = self._logits(x[last].contiguous(), residual[last].contiguous())
        return self.last_prefill_logits.argmax(dim=-1)

    # ----------------------------------------------------------------- decode
    def setup_decode(self, num_seqs: int, start_len: int, max_seq_len: int):
        """Allocate the static decode state (all sequences active, equal length)."""
        dev = self.device
        self.d_tokens = torch.zeros(num_seqs, dtype=torch.int64, device=dev)
        self.d_positions = torch.full((num_seqs,), start_len, dtype=torch.int64, device=dev)
        self.d_seq_lens = torch.full((num_seqs,), start_len + 1, dtype=torch.int32, device=dev)
        self.d_seq_ids = torch.arange(num_seqs, dtype=torch.int64, device=dev)
        self.d_bt = self.block_tables[:num_seqs].contiguous()
        self.d_slots = self._slots(self.d_seq_ids, self.d_positions)   # slot the next decode step writes
        self.d_max_seq_len = max_seq_len
        H = self.layers[0].q_heads
        # split-KV partition size of the v2 launch: 512 as in the reference, finer when (sequences x kv heads) alone
        # leaves CUs without a workgroup (a TP = 8 rank: 1 kv head) — attention/backend.py::decode_partition_size
        self.d_partition = decode_partition_size(num_seqs, H, self.layers[0].kv_heads, max_seq_len, self.BLOCK)
        P = (max_seq_len + self.d_partition - 1) // self.d_partition
        self.d_use_v1 = use_paged_attention_v1(num_seqs, H, max_seq_len, self.layers[0].kv_heads, self.cfg.head_dim,
                                               self.BLOCK, self.dtype) if self.device.type == "cuda" else True
        self.d_tmp = torch.empty(num_seqs, H, P, self.cfg.head_dim, dtype=self.dtype, device=dev)
        self.d_es = torch.empty(num_seqs, H, P, dtype=torch.float32, device=dev)
        self.d_ml = torch.empty_like(self.d_es)
        self._graph = None

    def set_decode_lengths(self, lengths: torch.Tensor):
        """positions[s] = lengths[s] (index of the token being decoded), seq_lens = lengths + 1."""
        self.d_positions.copy_(lengths.to(torch.int64))
        self.d_seq_lens.copy_((lengths + 1).to(torch.int32))
        self.d_slots.copy_(self._slots(self.d_seq_ids, self.d_positions))

    def _decode_body(self):
        slots = self.d_slots
        x = self.embed[self.d_tokens]
        residual = None

        def attn_fn(i, q3):
            out = torch.empty_like(q3)
            decode_attention(out, self.d_es, self.d_ml, self.d_tmp, q3, self.k_cache[i],
                             self.v_cache[i], self.layers[i].kv_heads, self.scale, self.d_bt,
                             self.d_seq_lens, self.BLOCK, self.d_max_seq_len, None, self.kv_dtype,
                             self.k_scale, self.v_scale, partition_size=self.d_partition)
            return out

        def attn_fused_fn(i, qkv, slabs, sk, scl=None, w8=False):
            L = self.layers[i]
            n = qkv.shape[0]
            out = torch.empty(n, L.q_heads, self.cfg.head_dim, dtype=qkv.dtype, device=qkv.device)
            # fp8 model, partitioned launch: its reduce kernel also quantises the output per token (o_proj's input)
            quant = None
            if w8 and not self.d_use_v1 and L.q_heads <= 16 \
                    and -(-self.d_max_seq_len // self.d_partition) <= 64:
                quant = (torch.empty(n, L.q_size, dtype=torch.float8_e4m3fn, device=qkv.device),
                         torch.empty(n, 1, dtype=torch.float32, device=qkv.device))
            ok = decode_attention_fused(out, self.d_es, self.d_ml, self.d_tmp, qkv, slabs, sk, self.d_positions,
                                        self.cos_sin, slots, self.k_cache[i], self.v_cache[i], L.q_heads,
                                        L.kv_heads, self.scale, self.d_bt, self.d_seq_lens, self.BLOCK,
                                        self.d_max_seq_len, use_v1=self.d_use_v1, partition_size=self.d_partition,
                                        slab_scales=scl, quant_out=quant)
            if not ok:
                return None
            return QuantAct(quant[0], quant[1], qkv.dtype) if quant is not None else out

        pending = (None, 0, None)
        for i in range(self.cfg.layers):
            x, residual, pending = self._layer(i, x, residual, self.d_positions, slots, attn_fn,
                                               pending, defer=True,
                                               attn_fused_fn=attn_fused_fn if self.fuse_attn_qkv else None)
        logits = self._logits(x, residual, pending)
        self.last_logits = logits      # (inside a captured graph: a tensor of the graph's pool, rewritten by every replay)
        if self.fuse_greedy:
            # argmax + positions / seq_lens += 1 + the next step's slots: one launch instead of ~13 torch ones
            ops.greedy_advance(logits.contiguous(), self.d_tokens, self.d_positions, self.d_seq_lens, self.d_slots,
                               self.d_bt, self.BLOCK)
        else:
            self.d_tokens.copy_(logits.argmax(dim=-1))
            self.d_positions.add_(1)
            self.d_seq_lens.add_(1)
            last = self.d_bt.shape[1] * self.BLOCK - 1      # (a full sequence has no next slot: clamped, unused)
            self.d_slots.copy_(self._slots(self.d_seq_ids, self.d_positions.clamp(max=last)))

    def decode_step(self, use_graph: bool = True):
        """One decode step for all sequences.  With use_graph the step is replayed from a HIP graph
        captured on first use — under tp > 1 the RCCL all-reduces / all-gather are captured in it too
        (torch.distributed's NCCL backend records them on the capturing stream).  If the capture
        fails (e.g. a backend that cannot be captured, such as gloo), the error is kept in
        `graph_error` and the step runs eagerly from then on."""
        if not use_graph or self._graph is False:
            self._decode_body()
            return
        if self._graph is None and self._collectives() and \
                torch.distributed.get_backend(self.tp_group) != "nccl":
            # only RCCL ("nccl") collectives can be recorded into a HIP graph; a host-staged backend
            # (gloo rehearsals on one GPU) would invalidate the capture
            self.graph_error = f"backend {torch.distributed.get_backend(self.tp_group)!r} cannot be captured"
            self._graph = False
            self._decode_body()
            return
        if self._graph is None:
            # warm up on a side stream (also initialises the communicator), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            saved = (self.d_tokens.clone(), self.d_positions.clone(), self.d_seq_lens.clone(), self.d_slots.clone())

            def restore():
                self.d_tokens.copy_(saved[0]); self.d_positions.copy_(saved[1]); self.d_seq_lens.copy_(saved[2])
                self.d_slots.copy_(saved[3])
            with torch.cuda.stream(s):
                self._decode_body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            restore()
            g = torch.cuda.CUDAGraph()
            try:
                # thread_local: the process group's watchdog thread may touch the runtime meanwhile
                mode = {"capture_error_mode": "thread_local"} if self.cfg.tp > 1 else {}
                with torch.cuda.graph(g, **mode):
                    self._decode_body()
                self._graph = g
            except Exception as e:  # noqa: BLE001 - keep serving eagerly, report why
                self.graph_error = repr(e)
                self._graph = False
                torch.cuda.synchronize()
                restore()
                self._decode_body()
                return
            restore()
        self._graph.replay()

    # -------------------------------------------------------------- accounting
    def decode_step_bytes(self, mean_len: float, num_seqs: int) -> dict:
        """Algorithmic HBM bytes of one decode step."""
        L = self.layers[0]
        w = sum(l.weight_bytes() for l in (L.qkv, L.o, L.gate_up, L.down)) * self.cfg.layers
        kv_elt = 2 if self.kv_dtype == "auto" else 1
        kv = num_seqs * mean_len * L.kv_heads * self.cfg.head_dim * 2 * kv_elt * self.cfg.layers
        head = self.lm_head.numel() * 2
        return {"weights": w, "kv": kv, "lm_head": head, "total": w + kv + head}


class PluginSurfaceModel(HotPathModel):
    """The SAME model driven only through what upstream vLLM reaches when this plugin is dropped in unchanged
    (north_star): `quant_config.linear.apply_awq / apply_gptq` (= torch.ops.vllm._apply_awq / _apply_gptq),
    `attention.backend.build_metadata + paged_attention_forward` (= Mi355xPagedAttentionImpl.forward), and
    `torch.ops._C.{rms_norm, fused_add_rms_norm, rotary_embedding, silu_and_mul}` (vLLM's RMSNorm / RotaryEmbedding
    / SiluAndMul.forward_cuda, ops/__init__.py) in the order of upstream's LlamaDecoderLayer.forward.
    None of the cross-op fusions of HotPathModel (operand images, SILU epilogues, slab-consuming norms, the qkv
    prologue of the decode attention, greedy_advance) is reachable from here — this is the measured cost of the
    plain op surface; `patch/fused_layers.py` (register_patch) is what brings the fusions back under vLLM.
    Sampling is torch.argmax + torch index arithmetic, as upstream's sampler / model runner do it."""

    def __init__(self, *a, prepack_weights: bool = False, patched: bool = False, **kw):
        """`patched`: the decoder layer as register_patch() rewires it under vLLM (patch/llama.py ->
        patch/fused_layers.py: input_layernorm + qkv_proj and post_attention_layernorm + MLP through the fused entry
        points — operand-image norms, SILU epilogues, image hand-over to down_proj); attention, rotary, cache write and
        sampling stay the plain ops."""
        super().__init__(*a, **kw)
        self.patched = patched
        import vllm_metax_amd._C  # noqa: F401  (registers torch.ops._C*)
        from .attention import backend as B
        from .quant_config import linear
        self.B, self.linear = B, linear
        cfg, d = self.cfg, self.cfg.head_dim
        kvh = self.layers[0].kv_heads
        nb = self.k_cache[0].shape[0]
        # the backend's KV cache shape (2, nb, bs * kvh * d); k_cache / v_cache become views of it
        self.kv_cache = [torch.zeros(B.kv_cache_shape(nb, self.BLOCK, kvh, d), dtype=self.k_cache[0].dtype,
                                     device=self.device) for _ in range(cfg.layers)]
        for i in range(cfg.layers):
            self.k_cache[i], self.v_cache[i] = B.split_kv_cache(self.kv_cache[i], kvh, d)
        self.workspace = B.DecodeWorkspace(self.max_seqs, self.layers[0].q_heads, d, self.max_len, self.dtype,
                                           self.device)
        # MI355X_PREPACK_WEIGHTS semantics of the plugin (on by default since round 3): the image belongs to the layer
        self.images = {}
        if prepack_weights and cfg.quant in ("awq", "gptq"):
            for i, L in enumerate(self.layers):
                for name in ("qkv", "o", "gate_up", "down"):
                    q = getattr(L, name)
                    self.images[(i, name)] = (ops.w4a16_prepack(q.qweight, q.qzeros, q.scales, cfg.quant == "gptq"),
                                              q.n, q.k)

    def _apply(self, i: int, name: str, x: torch.Tensor) -> torch.Tensor:
        q: QLinear = getattr(self.layers[i], name)
        if q.quant == "awq":
            # the reference's declared qweight shape is [N, K/8] over [K/8, N] memory (awq.py:138)
            return self.linear.apply_awq(x, q.qweight.view(q.n, -1), q.scales, q.qzeros, None, 8, q.group,
                                         image=self.images.get((i, name)))
        if q.quant == "gptq":
            return self.linear.apply_gptq(x, q.qweight, q.scales, q.qzeros, None, q.g_idx, True, 4, q.group, False,
                                          image=self.images.get((i, name)))
        return q(x)        # fp8 / int8: dynamic activation quant + cutlass_scaled_mm, already plain ops

    def _w4(self, i: int, name: str):
        from .patch import fused_layers as F
        q: QLinear = getattr(self.layers[i], name)
        return F.W4Linear(kind=q.quant, qweight=q.qweight, qzeros=q.qzeros, scales=q.scales, group_size=q.group,
                          g_idx=getattr(q, "g_idx", None), image=self.images.get((i, name)))

    def _patched_layer(self, i, x, residual, positions, md):
        """The layer forward register_patch() installs (patch/llama.py), on this model's weights."""
        from .patch import fused_layers as F
        L, cfg, C = self.layers[i], self.cfg, torch.ops._C
        qkv, residual = F.fused_norm_linear(x, residual, L.ln1, cfg.eps, self._w4(i, "qkv"))
        q, k, v = qkv.split([L.q_size, L.kv_size, L.kv_size], dim=-1)
        C.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin, True)
        T = q.shape[0]
        out = torch.empty(T, L.q_heads, cfg.head_dim, dtype=q.dtype, device=q.device)
        self.B.paged_attention_forward(q.view(T, L.q_heads, cfg.head_dim), k.view(T, L.kv_heads, cfg.head_dim),
                                       v.view(T, L.kv_heads, cfg.head_dim), self.kv_cache[i], md, out, L.kv_heads,
                                       self.scale, None, self.kv_dtype, self.k_scale, self.v_scale)
        o = self._all_reduce(self._apply(i, "o", out.view(T, L.q_size)))
        mlp, residual = F.fused_norm_mlp(o, residual, L.ln2, cfg.eps, self._w4(i, "gate_up"), self._w4(i, "down"))
        return self._all_reduce(mlp), residual

    def _surface_layer(self, i, x, residual, positions, md):
        if self.patched and self.cfg.quant in ("awq", "gptq"):
            return self._patched_layer(i, x, residual, positions, md)
        L, cfg, C = self.layers[i], self.cfg, torch.ops._C
        if residual is None:
            residual = x
            h = torch.empty_like(x)
            C.rms_norm(h, x, L.ln1, cfg.eps)
        else:
            C.fused_add_rms_norm(x, residual, L.ln1, cfg.eps)
            h = x
        qkv = self._apply(i, "qkv", h)
        q, k, v = qkv.split([L.q_size, L.kv_size, L.kv_size], dim=-1)
        C.rotary_embedding(positions, q, k, cfg.head_dim, self.cos_sin, True)
        T = q.shape[0]
        out = torch.empty(T, L.q_heads, cfg.head_dim, dtype=q.dtype, device=q.device)
        self.B.paged_attention_forward(q.view(T, L.q_heads, cfg.head_dim), k.view(T, L.kv_heads, cfg.head_dim),
                                       v.view(T, L.kv_heads, cfg.head_dim), self.kv_cache[i], md, out, L.kv_heads,
                                       self.scale, None, self.kv_dtype, self.k_scale, self.v_scale)
        o = self._all_reduce(self._apply(i, "o", out.view(T, L.q_size)))
        C.fused_add_rms_norm(o, residual, L.ln2, cfg.eps)
        gu = self._apply(i, "gate_up", o)
        act = torch.empty(T, L.ffn, dtype=gu.dtype, device=gu.device)
        C.silu_and_mul(act, gu)
        return self._all_reduce(self._apply(i, "down", act)), residual

    def _surface_logits(self, x, residual):
        torch.ops._C.fused_add_rms_norm(x, residual, self.final_norm, self.cfg.eps)
        return self._logits_from_hidden(x)

    def _logits_from_hidden(self, x):
        logits = torch.matmul(x, self.lm_head)
        if self._collectives():
            ws = torch.distributed.get_world_size(self.tp_group)
            g = torch.empty(ws * logits.shape[0], logits.shape[1], dtype=logits.dtype, device=logits.device)
            torch.distributed.all_gather_into_tensor(g, logits.contiguous(), group=self.tp_group)
            logits = g.view(ws, logits.shape[0], logits.shape[1]).permute(1, 0, 2).reshape(logits.shape[0], -1)
        return logits

    def prefill(self, token_ids, seq_ids, context_len: int = 0):
        n, q_len = token_ids.shape
        dev = self.device
        sid = torch.tensor(seq_ids, dtype=torch.int64, device=dev)
        pos = (torch.arange(q_len, device=dev) + context_len).repeat(n)
        slots = self._slots(sid.repeat_interleave(q_len), pos)
        cu_cpu = [i * q_len for i in range(n + 1)]
        cu = torch.tensor(cu_cpu, dtype=torch.int32, device=dev)
        sl_cpu = [context_len + q_len] * n
        seq_lens = torch.tensor(sl_cpu, dtype=torch.int32, device=dev)
        L0 = self.layers[0]
        md = self.B.build_metadata(cu, cu_cpu, seq_lens, sl_cpu, self.block_tables[sid], slots, n * q_len, q_len,
                                   context_len + q_len, L0.q_heads, self.cfg.head_dim, self.dtype, self.workspace,
                                   L0.kv_heads, self.BLOCK)
        x, residual = self.embed[token_ids.reshape(-1)], None
        for i in range(self.cfg.layers):
            x, residual = self._surface_layer(i, x, residual, pos, md)
        last = (cu[1:] - 1).long()
        self.last_prefill_logits = self._surface_logits(x[last].contiguous(), residual[last].contiguous())
        return self.last_prefill_logits.argmax(dim=-1)

    def _decode_body(self):
        n = self.d_tokens.shape[0]
        L0 = self.layers[0]
        cu_cpu = list(range(n + 1))
        if getattr(self, "_d_cu", None) is None or self._d_cu.numel() != n + 1:
            self._d_cu = torch.arange(n + 1, dtype=torch.int32, device=self.device)
        # host copies as vLLM's CommonAttentionMetadata carries them; the launch geometry comes from the fixed
        # maximum (HIP-graph capture), the kernels read the true lengths from d_seq_lens on the device
        md = self.B.build_metadata(self._d_cu, cu_cpu, self.d_seq_lens, [self.d_max_seq_len] * n, self.d_bt,
                                   self.d_slots, n, 1, self.d_max_seq_len, L0.q_heads, self.cfg.head_dim, self.dtype,
                                   self.workspace, L0.kv_heads, self.BLOCK, fixed_decode_len=self.d_max_seq_len)
        x, residual = self.embed[self.d_tokens], None
        for i in range(self.cfg.layers):
            x, residual = self._surface_layer(i, x, residual, self.d_positions, md)
        logits = self._surface_logits(x, residual)
        self.last_logits = logits
        self.d_tokens.copy_(logits.argmax(dim=-1))
        self.d_positions.add_(1)
        self.d_seq_lens.add_(1)
        last = self.d_bt.shape[1] * self.BLOCK - 1
        self.d_slots.copy_(self._slots(self.d_seq_ids, self.d_positions.clamp(max=last)))
