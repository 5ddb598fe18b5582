#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X hot path (contract: see the task brief).

Workload (BASELINE.json metric "output tokens/sec + p50 TTFT, Llama-3-8B AWQ-int4 TP=1"):
one STEP = one whole serving job on synthetic data — 64 prompts x 1024 tokens prefilled in
chunks of 8 sequences (8192 tokens, the chunked-prefill budget), then greedy decode to 128
output tokens per sequence (the first comes from the prefill, 127 decode steps follow, context
1024..1150).  Weights are synthetic tensors of the Llama-3-8B AWQ (w4a16, g=128) architecture,
KV cache block_size 16 in the reference's x-split paged layout.  Every hot-path op runs
through the C-ABI HIP library; decode steps are replayed from a HIP graph.

    value = K * 64 * 128 output tokens / wall time of K jobs (inputs resident in HBM).

`--gpus N` (one rank per GPU: launched by torch.distributed.run, or — when WORLD_SIZE is not set —
bench.py starts the N ranks itself as child processes; RCCL for the rendezvous, the collectives,
the barriers and the max-over-ranks clock).  For N > 1 the measured mode is the one north_star
names: TENSOR PARALLELISM TP=N over RCCL (`--parallelism tp`, the default for N > 1): ONE job of 64
prompts, every layer sharded Megatron-style (q/kv heads and FFN columns per rank, RCCL all-reduce of
the [M, hidden] bf16 activations after the row-parallel o_proj and down_proj, all-gather of the
vocab-parallel logits), the decode step INCLUDING its collectives replayed from one HIP graph: total
work fixed -> "scaling": "strong", value = output tokens of the job / max-over-ranks time.  The same
run then also times the collective-free alternative — one model replica per GPU, each serving its
own 64 prompts (`dp`, weak scaling) — and reports it as the extra object "dp_replicas"; `--parallelism
dp` makes that the headline instead.

Extra objects on the JSON line: "roofline" (dominant kernel, timed live with HIP events on
the launch stream inside the timed region), "roofline_other" (the other hot kernels),
"cpu_baseline" (oracle/cpu_port.c — the C restatement — timed on the host cores, rank 0,
N=1 only), "ttft_p50_ms".
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
MFMA_8BIT_PEAK_TFLOPS = 5000.0  # dense fp8 / int8 MFMA peak (the --quant fp8 / int8 GEMMs)
# model -> (harness.ModelConfig constructor, BASELINE.json's weight format, label)
MODELS = {"llama-3-8b": ("llama3_8b", "awq", "Llama-3-8B AWQ-int4"),
          "llama-3-70b": ("llama3_70b", "fp8", "Llama-3-70B FP8"),
          "qwen2-72b": ("qwen2_72b", "gptq", "Qwen2-72B GPTQ-int4"),
          "tiny": ("tiny", "awq", "tiny test model")}
WEIGHT_FORMAT = {"awq": "w4a16 g128", "gptq": "w4a16 g128", "fp8": "w8a8 fp8", "int8": "w8a8 int8",
                 "none": "bf16"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--quant", default=None, choices=["awq", "gptq", "fp8", "int8", "none"],
                    help="weight format; default: the one BASELINE.json names for the model (llama-3-8b: awq, "
                         "llama-3-70b: fp8, qwen2-72b: gptq)")
    ap.add_argument("--model", default="llama-3-8b", choices=sorted(MODELS),
                    help="llama-3-70b / qwen2-72b are BASELINE's TP=8 models: run them with --gpus 8, or rehearse "
                         "ONE rank's shard on one GPU with --tp-rank-of 8")
    ap.add_argument("--tp-rank-of", type=int, default=0, metavar="N",
                    help="single-GPU rehearsal of ONE rank of a TP=N group: builds rank --tp-rank's shard (heads, FFN "
                         "columns, vocabulary slice of 1/N), issues every RCCL collective on a 1-rank group (captured "
                         "in the decode graph) and reports that rank's job time; no peer traffic is measured")
    ap.add_argument("--tp-rank", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (rehearsals / tests only; "
                                                           "the line then says so and is not a BASELINE number)")
    ap.add_argument("--no-plugin-surface", action="store_true",
                    help="skip the extra measurement of the same job through the plain vLLM-reachable op surface "
                         "(the \"plugin_surface\" object; N = 1 only)")
    ap.add_argument("--no-tp8-extra", action="store_true",
                    help="--gpus 8 with the default model: skip the extra Llama-3-70B FP8 TP=8 job "
                         "(the \"baseline_tp8_model\" object)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=128)
    ap.add_argument("--chunk-seqs", type=int, default=8)
    ap.add_argument("--chunk-tokens", type=int, default=0,
                    help="chunked-prefill token budget per step (overrides --chunk-seqs): >= input-len: that many "
                         "tokens of whole sequences per chunk; smaller: every prompt is prefilled in pieces of this "
                         "many tokens against its own cached context (mid-size GEMMs, context > 0 attention)")
    ap.add_argument("--parallelism", default=None, choices=["dp", "tp"],
                    help="N > 1: tp (default) = one model sharded over the N GPUs (column/row-parallel "
                         "GEMMs, RCCL all-reduce after o_proj / down_proj, strong scaling); dp = one model "
                         "replica per GPU, each serving its own batch of --batch sequences (no data-path "
                         "collective, weak scaling)")
    ap.add_argument("--no-dp-extra", action="store_true",
                    help="tp runs: skip the additional dp-replica measurement (the \"dp_replicas\" object)")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8", "fp8_e5m2"],
                    help="fp8: e4m3 KV cache (SURVEY §8f-3), halves the bytes decode attention streams")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--kernel-stats", action="store_true",
                    help="print the per-kernel event timings table to stderr")
    return ap.parse_args()


class EventTimer:
    """HIP-event timing of individual launches on torch's current stream (the stream every
    C-ABI call is launched on).  Keeps (name, flops, bytes, start, stop) tuples."""

    def __init__(self):
        self.records = []
        self.enabled = False
        self.collect = None      # a list: record (name, flops, bytes, closure) of every launch instead of timing it

    def time(self, name, flops, nbytes, fn):
        if self.collect is not None:
            self.collect.append((name, flops, nbytes, fn))
            return fn()
        if not self.enabled:
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.records.append((name, flops, nbytes, a, b))
        return out

    def discard_last(self, name):
        """The call just wrapped reported "not applicable" (nothing was launched): drop its record."""
        lst = self.collect if self.collect is not None else (self.records if self.enabled else None)
        if lst and lst[-1][0] == name:
            lst.pop()

    def summary(self):
        agg = {}
        for name, flops, nbytes, a, b in self.records:
            ms = a.elapsed_time(b)
            d = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
            d["bytes"] += nbytes
        return agg


def instrument(model, timer: EventTimer):
    """Wrap the hot ops of the harness so that each launch is bracketed by HIP events."""
    from vllm_metax_amd import _custom_ops as ops
    from vllm_metax_amd import harness

    orig_call = harness.QLinear.__call__

    def timed_linear(self, x):
        m = x.shape[0]
        flops = 2.0 * m * self.n * self.k
        nbytes = self.weight_bytes() + 2.0 * m * self.k + 2.0 * m * self.n
        name = f"{self.quant}_gemm_{'large' if m >= 128 else 'small'}_m"
        return timer.time(name, flops, nbytes, lambda: orig_call(self, x))

    harness.QLinear.__call__ = timed_linear

    # the fused variants of the same GEMMs (silu epilogue, consumer-side split-K sum) count under
    # the same kernel names: same weights, same flops, one launch less downstream
    orig_silu, orig_def = harness.QLinear.silu_mul, harness.QLinear.deferred

    def timed_silu(self, x):
        m = x.shape[0]
        flops = 2.0 * m * self.n * self.k
        nbytes = self.weight_bytes() + 2.0 * m * self.k + 1.0 * m * self.n
        name = f"{self.quant}_gemm_{'large' if m >= 128 else 'small'}_m"
        out = timer.time(name, flops, nbytes, lambda: orig_silu(self, x))
        if out is None:
            timer.discard_last(name)     # fused path not taken: the caller times the plain GEMM
        return out

    def timed_deferred(self, x, allow_scaled=False):
        m = x.shape[0]
        own = (self.quant == "awq" and m <= 64) or \
            (self.quant == "fp8" and m <= 64 and allow_scaled and harness.QLinear.fp8_defer)
        if not own:
            return orig_def(self, x, allow_scaled)     # falls through to __call__, which is timed
        flops = 2.0 * m * self.n * self.k
        nbytes = self.weight_bytes() + 2.0 * m * self.k + 2.0 * m * self.n
        # (fp8 with an un-quantised input: the record then includes its per-token quant launch, as __call__'s does)
        return timer.time(f"{self.quant}_gemm_small_m", flops, nbytes, lambda: orig_def(self, x, allow_scaled))

    harness.QLinear.silu_mul = timed_silu
    harness.QLinear.deferred = timed_deferred

    def wrap(opname, cost, record_as=None):
        """record_as: the fused / image forms of an op are recorded under the name of the op they stand for; a
        call that reports "not applicable" (None / False: nothing was launched) leaves no record."""
        fn = getattr(ops, opname)
        name = record_as or opname

        def w(*a, **k):
            flops, nbytes = cost(*a, **k)
            out = timer.time(name, flops, nbytes, lambda: fn(*a, **k))
            if record_as and (out is None or out is False):
                timer.discard_last(name)
            return out
        setattr(ops, opname, w)
        if opname == "paged_attention_fused_qkv":
            from vllm_metax_amd.attention import backend
            backend.ops = ops          # (the backend calls it through its own `ops` reference: same module)

    def cost_prefill(out, q, kc, vc, kvh, scale, bt, sl, cu, max_q, bs, *a, **k):
        T, H, D = q.shape
        n = sl.numel()
        ql = T // n
        L = int(model.cfg_ctx_for_cost) + ql
        flops = 4.0 * n * ql * (L - ql / 2.0) * H * D
        nbytes = 2.0 * T * H * D * 2 + n * L * kvh * D * 2 * kc.element_size()
        return flops, nbytes

    def cost_decode(out, es, ml, tmp, q, kc, vc, kvh, scale, bt, sl, bs, max_len, *a, **k):
        S, H, D = q.shape
        mean_len = model.mean_decode_len_for_cost
        nbytes = S * mean_len * kvh * D * 2 * kc.element_size() + 2.0 * S * H * D * 2
        return 4.0 * S * mean_len * H * D, nbytes

    def cost_rows(n_reads, n_writes):
        def c(*a, **k):
            t = a[0]
            return 0.0, float(t.numel() * t.element_size() * (n_reads + n_writes))
        return c

    wrap("paged_prefill_attention", cost_prefill)
    wrap("paged_prefill_attention_image",
         lambda q, kc, vc, kvh, scale, bt, sl, cu, max_q, bs, *a, **k:
         cost_prefill(None, q, kc, vc, kvh, scale, bt, sl, cu, max_q, bs), record_as="paged_prefill_attention")

    def cost_decode_fused(out, es, ml, tmp, qkv, slabs, sk, positions, cos_sin, slots, kc, vc, nh, kvh, scale, bt,
                          sl, bs, max_len, partitioned, *a, **k):
        # the attention bytes (as paged_attention_v1 / _v2) + the qkv row / slabs of the folded qkv_rope_cache
        S, D = qkv.shape[0], kc.shape[2] * kc.shape[4]
        mean_len = model.mean_decode_len_for_cost
        nbytes = S * mean_len * kvh * D * 2 * kc.element_size() + 2.0 * S * nh * D * 2 \
            + qkv.numel() * (4.0 * sk if sk > 0 else 2.0)
        return 4.0 * S * mean_len * nh * D, nbytes
    wrap("paged_attention_fused_qkv", cost_decode_fused, record_as="paged_attention_v1")
    wrap("rms_norm_image", lambda x, w, eps: (0.0, float(x.numel() * x.element_size() * 2)), record_as="rms_norm")
    wrap("fused_add_rms_norm_image", lambda x, r, w, eps: (0.0, float(x.numel() * x.element_size() * 4)),
         record_as="fused_add_rms_norm")
    wrap("rms_norm_dynamic_per_token_quant",
         lambda out, inp, w, sc, eps, ub=None, res=None:
         (0.0, float(inp.numel() * inp.element_size() * (3 if res is not None else 1) + out.numel())))
    wrap("dynamic_per_token_scaled_fp8_quant",
         lambda out, inp, sc, ub=None: (0.0, float(inp.numel() * inp.element_size() + out.numel())))
    wrap("silu_and_mul_per_token_quant", lambda inp: (0.0, float(inp.numel() * inp.element_size() + inp.numel() // 2)),
         record_as="silu_and_mul")
    wrap("silu_and_mul_per_token_quant_slabs",
         lambda slabs, sk, a_s, b_s, n, d, dt: (0.0, float(n * d * (8.0 * sk + 1))), record_as="silu_and_mul")
    wrap("rms_norm_dynamic_per_token_quant_slabs",
         lambda out, slabs, sk, *a, **k: (0.0, float(out.numel() * (4.0 * sk + 1 + 4))),
         record_as="rms_norm_dynamic_per_token_quant")
    wrap("greedy_advance", lambda logits, *a, **k: (0.0, float(logits.numel() * logits.element_size())))
    wrap("rotary_reshape_and_cache",
         lambda pos, key, value, kc, *a, **k: (0.0, 2.0 * key.numel() * (2 + kc.element_size())),
         record_as="reshape_and_cache")
    wrap("paged_attention_v2", cost_decode)
    wrap("paged_attention_v1",
         lambda out, q, kc, vc, kvh, scale, bt, sl, bs, max_len, *a, **k:
         cost_decode(out, None, None, None, q, kc, vc, kvh, scale, bt, sl, bs, max_len))
    wrap("fused_add_rms_norm", cost_rows(2, 2))             # reads x, residual; writes residual, x (in place)
    wrap("fused_add_rms_norm_slabs", cost_rows(3, 2))       # (+ sk fp32 slabs when sk > 0)
    wrap("qkv_rope_cache", cost_rows(1, 1))
    wrap("rms_norm", cost_rows(1, 1))
    wrap("silu_and_mul", cost_rows(2, 1))
    wrap("rotary_embedding", lambda pos, q, k, *a: (0.0, 2.0 * (q.numel() + (k.numel() if k is not None else 0)) * 2))
    wrap("reshape_and_cache", lambda key, value, kc, *a, **k: (0.0, 2.0 * key.numel() * (2 + kc.element_size())))


def max_over_ranks(elapsed: float, device, world: int) -> float:
    """The job is as slow as its slowest rank (contract: MAX over ranks)."""
    if world <= 1:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return t.item()


def whole_job_tokens(steps: int, batch: int, output_len: int, world: int, tp: int) -> int:
    """Output tokens produced by ALL ranks: dp replicas each serve their own batch, a tp group
    serves one."""
    return steps * batch * output_len * (world if tp == 1 else 1)


def run_job(model, tokens, args, timer=None, ttft=None):
    """One whole job.  Returns nothing; `ttft` (list) receives per-sequence first-token events."""
    B, Lin, Lout = args.batch, args.input_len, args.output_len
    first = torch.empty(B, dtype=torch.int64, device=model.device)
    piece = args.chunk_tokens if 0 < args.chunk_tokens < Lin else Lin
    for c0 in range(0, B, args.chunk_seqs):
        ids = list(range(c0, min(c0 + args.chunk_seqs, B)))
        for ctx in range(0, Lin, piece):       # (one pass unless --chunk-tokens < input-len)
            model.cfg_ctx_for_cost = ctx
            nxt = model.prefill(tokens[ids[0]:ids[-1] + 1, ctx:ctx + piece], ids, ctx)
        first[ids[0]:ids[-1] + 1] = nxt
        if ttft is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            ttft.append((len(ids), ev))
    model.d_tokens.copy_(first)
    model.set_decode_lengths(torch.full((B,), Lin, device=model.device))
    for _ in range(Lout - 1):
        model.decode_step(use_graph=not args.no_graph)


def cpu_baseline(args, cfg):
    """oracle/cpu_port.c on the host cores: one decoder layer at the decode shape + lm_head,
    extrapolated to the job (see the 'sample' string)."""
    from oracle import cpu_port as C
    from oracle import ref_ops as R
    import numpy as np
    torch.manual_seed(0)
    # a 1-GPU box grants ~16 of the host's hardware threads: a team as wide as the machine only
    # fights over them (measured: 0.7-1.1 tokens/s with 128 threads, 1.6 with 8)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    C.set_num_threads(min(16, avail))
    B, ctx = args.batch, args.input_len + args.output_len // 2
    h, d, H, KVH, ffn, g = cfg.hidden, cfg.head_dim, cfg.heads, cfg.kv_heads, cfg.ffn, cfg.group_size
    bf = torch.bfloat16

    def mk(k, n):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), dtype=torch.int32)
        qz = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // g, n // 8), dtype=torch.int32)
        sc = (torch.rand(k // g, n) * 4e-3 + 1e-3).to(bf)
        return qw, qz, sc
    W = {"qkv": mk(h, (H + 2 * KVH) * d), "o": mk(H * d, h), "gu": mk(h, 2 * ffn), "down": mk(ffn, h)}
    nblk = (ctx + 15) // 16
    kc = (torch.rand(B * nblk, KVH, d // 8, 16, 8) * 0.2 - 0.1).to(bf)
    vc = (torch.rand(B * nblk, KVH, d, 16) * 0.2 - 0.1).to(bf)
    bt = torch.randperm(B * nblk).to(torch.int32).reshape(B, nblk)
    sl = torch.full((B,), ctx, dtype=torch.int32)
    x = (torch.randn(B, h) * 0.5).to(bf)
    res = x.clone()
    ln = torch.ones(h, dtype=bf)
    cache = torch.randn(ctx + 1, d).to(bf)
    pos = torch.full((B,), ctx - 1, dtype=torch.int64)
    slots = (bt[:, (ctx - 1) // 16].long() * 16 + (ctx - 1) % 16)
    lm = (torch.randn(h, cfg.vocab) * 0.02).to(bf)

    def layer():
        C.fused_add_rms_norm(x, res, ln, cfg.eps)
        qkv = C.w4a16_gemm(x, *[W["qkv"][i] for i in (0, 2, 1)], 0, g)
        q, k, v = qkv[:, :H * d], qkv[:, H * d:(H + KVH) * d], qkv[:, (H + KVH) * d:]
        C.rotary_neox(pos, q, k, cache, H, KVH, d)
        C.reshape_and_cache(k.reshape(B, KVH, d), v.reshape(B, KVH, d).contiguous(), kc, vc, slots)
        a = C.paged_attention_v1(q.reshape(B, H, d), kc, vc, KVH, d ** -0.5, bt, sl)
        o = C.w4a16_gemm(a.reshape(B, H * d), *[W["o"][i] for i in (0, 2, 1)], 0, g)
        C.fused_add_rms_norm(o, res, ln, cfg.eps)
        gu = C.w4a16_gemm(o, *[W["gu"][i] for i in (0, 2, 1)], 0, g)
        act = C.silu_and_mul(gu)
        return C.w4a16_gemm(act, *[W["down"][i] for i in (0, 2, 1)], 0, g)

    layer()  # warm (page in, thread pool)
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps):
        layer()
    t_layer = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        C.gemm_bf16(x, lm)
    t_head = (time.perf_counter() - t0) / reps
    # job estimate: every token position (prefill + decode) pays the per-token layer cost of
    # the M=64 sample; prefill attention is NOT counted (under-estimates CPU time).
    positions = args.batch * (args.input_len + args.output_len - 1)
    steps_equiv = positions / float(B)
    t_job = cfg.layers * t_layer * steps_equiv + t_head * (args.output_len)
    return {
        "value": round(args.batch * args.output_len / t_job, 3),
        "unit": "output tokens/s",
        "cores": C.num_threads(),
        "kind": "port",
        "sample": (f"oracle/cpu_port.c: 1 of {cfg.layers} decoder layers at batch {B}, ctx {ctx} "
                   f"({t_layer:.2f} s, mean of {reps}) + lm_head ({t_head:.2f} s, mean of {reps}); "
                   f"job time extrapolated as "
                   f"layers x per-64-token layer cost x {steps_equiv:.0f} token groups + {args.output_len} "
                   f"lm_head calls, prefill attention not counted"),
    }


def spawn_ranks(n: int, limit_s: float = 3000.0) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (the
    parent never touches the GPU), relay rank 0's JSON line, return the worst exit code.  Every child is
    polled: when one exits non-zero (OOM, import error, RCCL init) or the wall-clock limit passes, the others
    are terminated instead of waiting forever in a rendezvous or a collective."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()          # (released before the children bind it: a lost race shows up as a failed rendezvous -> non-zero)
    procs, rc = [], 0
    out_f = tempfile.TemporaryFile(mode="w+")
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out_f if r == 0 else subprocess.DEVNULL))
        t0 = time.time()
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                rc = max(abs(c) for c in codes)
                break
            failed = [i for i, c in enumerate(codes) if c not in (None, 0)]
            if failed or time.time() - t0 > limit_s:
                why = f"rank {failed[0]} exited with {codes[failed[0]]}" if failed else f"no result after {limit_s:.0f} s"
                print(f"bench.py: {why}: stopping the other ranks", file=sys.stderr)
                rc = max([abs(codes[i]) for i in failed] + [1])
                break
            time.sleep(0.5)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:  # noqa: BLE001
                p.kill()
    out_f.seek(0)
    sys.stdout.write(out_f.read())
    sys.stdout.flush()
    return rc


def timed_jobs(model, tokens, args, world, barrier, timer=None, steps=None, warmup=None):
    """W warm-up jobs (at least one: it builds the decode graph), then EXACTLY K timed jobs bracketed by
    barrier + synchronize; returns (max-over-ranks seconds, ttft samples)."""
    steps = args.steps if steps is None else steps
    for _ in range(max(args.warmup if warmup is None else warmup, 1)):
        run_job(model, tokens, args)
    barrier()
    if timer is not None:
        timer.enabled = True
    ttft_events, start_ev = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        tt = []
        run_job(model, tokens, args, ttft=tt)
        start_ev.append(s)
        ttft_events.append(tt)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, model.device, world)
    if timer is not None:
        timer.enabled = False
    ttfts = []
    for s, tt in zip(start_ev, ttft_events):
        for n, ev in tt:
            ttfts += [s.elapsed_time(ev)] * n
    return elapsed, ttfts


def check_outputs(model, what: str) -> None:
    """A job whose logits are non-finite or degenerate is not a measurement (round 2's --quant fp8 records were
    taken on such a job): fail loudly instead of printing a number."""
    for name in ("last_prefill_logits", "last_logits"):
        lg = getattr(model, name, None)
        if lg is None:
            continue
        lg = lg.float()
        if not bool(torch.isfinite(lg).all()) or float(lg.abs().max()) == 0.0:
            raise SystemExit(f"bench.py: {what}: {name} are non-finite or all zero — the job computed garbage, "
                             f"no number is reported")
    tok = model.d_tokens
    if int(tok.min()) < 0 or int(tok.max()) >= model.cfg.vocab:
        raise SystemExit(f"bench.py: {what}: sampled tokens outside the vocabulary")


def decode_kernel_times(model, timer: EventTimer, args, reps: int = 3):
    """Graph-consistent timings of the decode step's kernels.  One eager decode step is run with the timer in
    collect mode (it records every wrapped launch as a closure over that step's real tensors); then, per kernel
    name, ALL its launches of the step (one per layer and projection: 32-128 launches over different weights /
    KV caches, > 256 MiB in total, so nothing is served from the Infinity Cache that would not be in the real
    step) are captured into one HIP graph and replayed `reps` times between two HIP events.  The eager bracket
    the earlier rounds used put ~13 us of launch latency inside each decode GEMM's events."""
    mid = args.input_len + (args.output_len - 1) // 2
    model.set_decode_lengths(torch.full((args.batch,), mid, device=model.device))
    model.mean_decode_len_for_cost = mid + 1
    saved = (model.d_tokens.clone(), model.d_positions.clone(), model.d_seq_lens.clone(), model.d_slots.clone())
    timer.collect = []
    model.decode_step(use_graph=False)
    torch.cuda.synchronize()
    calls, timer.collect = timer.collect, None
    groups = {}
    for name, flops, nbytes, fn in calls:
        groups.setdefault(name, []).append((flops, nbytes, fn))
    out = {}
    for name, items in groups.items():
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _, _, fn in items:
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _, _, fn in items:
                    fn()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            n = len(items) * reps
            out[name] = {"launches": n, "ms": a.elapsed_time(b), "steps_timed": reps,
                         "flops": sum(i[0] for i in items) * reps,
                         "bytes": sum(i[1] for i in items) * reps,
                         "timing": f"HIP-graph replay of the {len(items)} launches of one decode step, x{reps}"}
            del g
        except Exception as e:  # noqa: BLE001 - an op that cannot be captured keeps no entry
            print(f"bench.py: decode kernel timing of {name} failed: {e!r}", file=sys.stderr)
            torch.cuda.synchronize()
    for dst, src in zip((model.d_tokens, model.d_positions, model.d_seq_lens, model.d_slots), saved):
        dst.copy_(src)
    return out


def no_prepack_entry(model, args, timer: EventTimer):
    """The prefill w4a16 GEMM WITHOUT the load-time weight image (int4 words dequantised inside every call):
    the four projections of layer 0 at the chunk size, 3 calls each, HIP events — reported beside the headline's
    image-based GEMM so that the line shows what the image buys."""
    from vllm_metax_amd import _custom_ops as ops
    L = model.layers[0]
    m = args.chunk_seqs * min(args.input_len, args.chunk_tokens or args.input_len)
    tot_ms, tot_flops, n = 0.0, 0.0, 0
    for q in (L.qkv, L.o, L.gate_up, L.down):
        if q.quant not in ("awq", "gptq"):
            return None
        x = (torch.randn(m, q.k, device=model.device) * 0.5).to(model.dtype)

        def call():
            if q.quant == "awq":
                return ops.awq_gemm(x, q.qweight, q.qzeros, q.scales, 8, torch.empty(0), x.dtype == torch.bfloat16)
            return ops.gptq_gemm(x, q.qweight, q.qzeros, q.scales, q.g_idx, True, 4, q.group, torch.empty(0),
                                 torch.empty(0), x.dtype == torch.bfloat16)
        call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            call()
        b.record()
        torch.cuda.synchronize()
        tot_ms += a.elapsed_time(b)
        tot_flops += 3 * 2.0 * m * q.n * q.k
        n += 3
    ach = tot_flops / (tot_ms * 1e-3) / 1e12
    return {"kernel": "awq_gemm_large_m_no_prepack", "bound": "mfma", "achieved": round(ach, 2),
            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "frac_fp8_peak": round(ach / MFMA_8BIT_PEAK_TFLOPS, 4), "traffic": None,
            "avg_launch_us": round(tot_ms / n * 1e3, 2), "launches": n, "total_ms": round(tot_ms, 2),
            "note": "the same four layer-0 GEMMs at the chunk size through awq_gemm / gptq_gemm (weights dequantised "
                    "per call: dequant-pack + activation pack + MFMA kernel), measured after the timed region"}


def hbm_weights_gb(model) -> dict:
    """Bytes the model keeps resident in HBM, by kind (GB = 1e9 B)."""
    w = img = 0
    for L in model.layers:
        for q in (L.qkv, L.o, L.gate_up, L.down):
            w += q.weight_bytes()
            for im in (getattr(q, "_image", None), getattr(q, "_w8_image", None)):
                if im is not None and im is not False:
                    img += im.numel() * im.element_size()
    other = (model.embed.numel() + model.lm_head.numel()) * model.embed.element_size()
    return {"quantized_weights": round(w / 1e9, 3), "prefill_weight_images": round(img / 1e9, 3),
            "embed_lm_head": round(other / 1e9, 3), "total": round((w + img + other) / 1e9, 3)}


def make_cfg(args, tp_degree, rank, model_name=None, quant=None):
    from vllm_metax_amd import harness
    name = model_name or args.model
    ctor, default_quant, _ = MODELS[name]
    q = quant or args.quant or default_quant
    cfg = getattr(harness.ModelConfig, ctor)(q)
    cfg.tp, cfg.tp_rank = tp_degree, (rank if tp_degree > 1 else 0)
    cfg.kv_cache_dtype = args.kv_cache_dtype
    if args.layers:
        cfg.layers = args.layers
    return cfg


def main():
    args = parse_args()
    if args.chunk_tokens:
        args.chunk_seqs = max(1, args.chunk_tokens // args.input_len)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))      # (before any GPU call in this process)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}), "
                         f"or run bench.py --gpus N without a launcher")
    if args.tp_rank_of and world != 1:
        raise SystemExit("--tp-rank-of is a single-GPU rehearsal: use it with --gpus 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # rehearsal on a 1-GPU box: BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU0=1 runs N ranks on cuda:0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1 or args.tp_rank_of:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.tp_rank_of:      # a 1-rank RCCL group: every collective is issued (and captured), nothing travels
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                                 device_id=torch.device("cuda", local_rank))
        elif backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)
        group = torch.distributed.group.WORLD

    from vllm_metax_amd import harness
    parallelism = args.parallelism or ("tp" if world > 1 else "dp")
    tp = args.tp_rank_of or (world if parallelism == "tp" else 1)
    cfg = make_cfg(args, tp, args.tp_rank if args.tp_rank_of else rank)
    quant = cfg.quant
    max_len = args.input_len + args.output_len
    model = harness.HotPathModel(cfg, args.batch, max_len, device=f"cuda:{local_rank}", seed=0,
                                 tp_group=group if tp > 1 else None)
    if args.tp_rank_of:
        model.collectives_always = True
    model.setup_decode(args.batch, args.input_len, max_len)
    model.cfg_ctx_for_cost = 0
    model.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
    gen = torch.Generator(device=model.device).manual_seed(0)
    tokens = torch.randint(0, cfg.vocab, (args.batch, args.input_len), device=model.device, generator=gen)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # every hot op is wrapped so that a launch CAN be bracketed by HIP events; the timer is off during warm-up /
    # graph capture and on inside the timed region, where it sees the eagerly launched kernels (all of prefill);
    # the graph-replayed decode launches are timed afterwards, again from HIP graphs (decode_kernel_times).
    timer = EventTimer()
    instrument(model, timer)
    # ---- timed region: exactly K jobs -------------------------------------------------
    elapsed, ttfts = timed_jobs(model, tokens, args, world, barrier, timer)
    ttft_p50 = statistics.median(ttfts) if ttfts else None
    check_outputs(model, f"{cfg.name} {quant}")
    graph_ok, graph_err = model._graph not in (None, False), model.graph_error
    weights_gb = hbm_weights_gb(model)

    agg = timer.summary()
    dec = decode_kernel_times(model, timer, args)
    dec_named = {}
    for name, d in dec.items():  # decode launches: graph-consistent timings, beside the prefill-side entries
        final = name if name not in agg else name + "_decode"
        agg[final] = d
        dec_named[final] = d

    def roof(name, d):
        avg_ms = d["ms"] / d["launches"]
        extra = {"timing": d["timing"]} if "timing" in d else {}
        if d["flops"] > 0 and name.endswith("large_m") or name == "paged_prefill_attention":
            ach = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
            peak = MFMA_8BIT_PEAK_TFLOPS if name.startswith(("fp8_gemm", "int8_gemm")) \
                else MFMA_BF16_PEAK_TFLOPS
            r = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2),
                 "peak": peak, "unit": "TFLOP/s",
                 "frac": round(ach / peak, 4), "traffic": None,
                 "avg_launch_us": round(avg_ms * 1e3, 2), "launches": d["launches"],
                 "total_ms": round(d["ms"], 2), **extra}
            if name.endswith("gemm_large_m"):
                # north_star words its GEMM target against the fp8 MFMA peak (5 PFLOP/s dense); the
                # w4a16 GEMM multiplies bf16 operands, so `frac` is against the bf16 peak and this
                # field restates the same rate against the fp8 peak (SURVEY §8d asks for both)
                r["frac_fp8_peak"] = round(ach / MFMA_8BIT_PEAK_TFLOPS, 4)
            return r
        ach = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_us": round(avg_ms * 1e3, 2), "launches": d["launches"],
                "total_ms": round(d["ms"], 2), **extra}

    # ranking by the share of the JOB: a prefill kernel's events cover the K timed jobs, a decode kernel's graph
    # timing covers `reps` steps — scale the latter to the decode steps of the K jobs before sorting
    def job_ms(name, d):
        if name in dec_named:
            return d["ms"] / d["steps_timed"] * (args.output_len - 1) * args.steps
        return d["ms"]
    roofs = []
    for n_, d in agg.items():
        r = roof(n_, d)
        r["job_share"] = round(job_ms(n_, d) / (elapsed * 1e3), 4)
        roofs.append(r)
    roofs.sort(key=lambda r: -r["job_share"])
    # HBM-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs of
    # the same kernels at the same shapes; a counter pass cannot run inside the timed region)
    here = os.path.dirname(os.path.abspath(__file__))
    tpath = next((pth for pth in (os.path.join(here, "profiles", f) for f in ("r03_pmc_traffic.json",
                                                                              "r02_pmc_traffic.json"))
                  if os.path.exists(pth)), None)
    if tpath and args.model == "llama-3-8b" and tp == 1 and args.kv_cache_dtype == "auto" and quant == "awq":
        with open(tpath) as f:
            traffic = json.load(f)
        for r in roofs:
            if r["kernel"] in traffic:
                r["traffic"] = traffic[r["kernel"]]
                r["traffic_unit"] = f"bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/{os.path.basename(tpath)})"
    if quant in ("awq", "gptq") and harness.QLinear.prepack and world == 1 and not args.tp_rank_of:
        try:
            e = no_prepack_entry(model, args, timer)
            if e is not None:
                roofs.append(e)
        except Exception as ex:  # noqa: BLE001
            print(f"bench.py: no-prepack entry failed: {ex!r}", file=sys.stderr)
    if args.kernel_stats and rank == 0:
        for r in roofs:
            print(json.dumps(r), file=sys.stderr)

    # ---- tp runs: the collective-free alternative (one replica per GPU) measured in the same run ----
    dp_extra = None
    if tp > 1 and world > 1 and not args.no_dp_extra and args.model == "llama-3-8b":
        del model
        torch.cuda.empty_cache()
        rep_model = harness.HotPathModel(make_cfg(args, 1, 0), args.batch, max_len, device=f"cuda:{local_rank}", seed=0)
        rep_model.setup_decode(args.batch, args.input_len, max_len)
        rep_model.cfg_ctx_for_cost = 0
        rep_model.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
        dp_elapsed, _ = timed_jobs(rep_model, tokens, args, world, barrier)
        dp_extra = {"value": round(whole_job_tokens(args.steps, args.batch, args.output_len, world, 1) / dp_elapsed, 2),
                    "unit": "output tokens/s", "scaling": "weak", "ms_per_step": round(dp_elapsed / args.steps * 1e3, 3),
                    "global_batch": args.batch * world,
                    "note": "one model replica per GPU, each serving its own batch: no data-path collective"}
        del rep_model
        model = None

    # ---- --gpus 8, default model: BASELINE's TP=8 headline model (Llama-3-70B FP8) in the same run ----
    tp8_extra = None
    if world == 8 and tp == 8 and args.model == "llama-3-8b" and not args.no_tp8_extra:
        model = None
        torch.cuda.empty_cache()
        try:
            c70 = make_cfg(args, 8, rank, model_name="llama-3-70b", quant="fp8")
            m70 = harness.HotPathModel(c70, args.batch, max_len, device=f"cuda:{local_rank}", seed=0, tp_group=group)
            m70.setup_decode(args.batch, args.input_len, max_len)
            m70.cfg_ctx_for_cost = 0
            m70.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
            t70 = torch.randint(0, c70.vocab, (args.batch, args.input_len), device=m70.device, generator=gen)
            k70 = min(args.steps, 3)
            e70, tt70 = timed_jobs(m70, t70, args, world, barrier, steps=k70, warmup=1)
            check_outputs(m70, "llama-3-70b fp8 tp8")
            tp8_extra = {"model": "Llama-3-70B FP8 (w8a8 fp8, per-channel weight scales), TP=8", "steps": k70,
                         "value": round(k70 * args.batch * args.output_len / e70, 2), "unit": "output tokens/s",
                         "ms_per_step": round(e70 / k70 * 1e3, 3),
                         "ttft_p50_ms": round(statistics.median(tt70), 2) if tt70 else None,
                         "decode_graph": m70._graph not in (None, False), "hbm_weights_gb_per_rank": hbm_weights_gb(m70)}
            del m70
        except Exception as ex:  # noqa: BLE001 - the scaling line must survive a failure of the extra job
            tp8_extra = {"model": "Llama-3-70B FP8, TP=8", "error": repr(ex)}

    # ---- N = 1: the same job through the plain op surface upstream vLLM reaches (no cross-op fusion) ----
    surface = None
    if world == 1 and not args.tp_rank_of and not args.no_plugin_surface:
        model = None
        torch.cuda.empty_cache()
        try:
            from vllm_metax_amd import envs as _envs
            surface_prepack = bool(_envs.MI355X_PREPACK_WEIGHTS)
            sm = harness.PluginSurfaceModel(make_cfg(args, 1, 0), args.batch, max_len, device=f"cuda:{local_rank}",
                                            seed=0, prepack_weights=surface_prepack)
            sm.setup_decode(args.batch, args.input_len, max_len)
            sm.cfg_ctx_for_cost = 0
            sm.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
            ks = min(args.steps, 3)
            se, stt = timed_jobs(sm, tokens, args, world, barrier, steps=ks, warmup=1)
            check_outputs(sm, "plugin surface")
            surface = {"value": round(ks * args.batch * args.output_len / se, 2), "unit": "output tokens/s",
                       "steps": ks, "ms_per_step": round(se / ks * 1e3, 3),
                       "ttft_p50_ms": round(statistics.median(stt), 2) if stt else None,
                       "decode_graph": sm._graph not in (None, False),
                       "path": "quant_config.linear.apply_awq/apply_gptq (= torch.ops.vllm._apply_*), "
                               "attention.backend.build_metadata + paged_attention_forward, torch.ops._C.{rms_norm, "
                               "fused_add_rms_norm, rotary_embedding, silu_and_mul}, torch argmax; no operand images, "
                               "no fused epilogues / prologues, MI355X_PREPACK_WEIGHTS "
                               + ("on (the plugin's default: prefill-sized linears multiply by the load-time weight image)"
                                  if surface_prepack else "off")}
            del sm
            if quant in ("awq", "gptq"):
                # ... and with the decoder-layer forwards register_patch() installs under vLLM (patch/fused_layers.py)
                torch.cuda.empty_cache()
                sm = harness.PluginSurfaceModel(make_cfg(args, 1, 0), args.batch, max_len,
                                                device=f"cuda:{local_rank}", seed=0,
                                                prepack_weights=surface_prepack, patched=True)
                sm.setup_decode(args.batch, args.input_len, max_len)
                sm.cfg_ctx_for_cost = 0
                sm.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
                se, stt = timed_jobs(sm, tokens, args, world, barrier, steps=ks, warmup=1)
                check_outputs(sm, "plugin surface (patched)")
                surface["with_register_patch"] = {
                    "value": round(ks * args.batch * args.output_len / se, 2), "ms_per_step": round(se / ks * 1e3, 3),
                    "ttft_p50_ms": round(statistics.median(stt), 2) if stt else None,
                    "path": "the same, with the LlamaDecoderLayer / LlamaMLP forwards of vllm_metax_amd/patch "
                            "(fused_norm_linear, fused_norm_mlp): what register_patch() adds under vLLM"}
                del sm
        except Exception as ex:  # noqa: BLE001
            surface = {"error": repr(ex)}

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    replicas = world if tp == 1 else 1       # dp: every rank served its own batch
    out_tokens = whole_job_tokens(args.steps, args.batch, args.output_len, world, tp)
    label = MODELS[args.model][2]
    if args.quant and args.quant != MODELS[args.model][1]:
        label = f"{label.split()[0]} {WEIGHT_FORMAT.get(quant, quant)}"
    image_on = bool(quant in ("awq", "gptq") and harness.QLinear.prepack and weights_gb["prefill_weight_images"] > 0)
    par = f"tp{world}" if (tp > 1 and not args.tp_rank_of) else f"dp{world}"
    if args.tp_rank_of:
        par = f"rank {args.tp_rank} of tp{tp}, rehearsed on one GPU"
    result = {
        "metric": f"output tokens/sec ({label}, batch {args.batch}, {args.input_len}-in/{args.output_len}-out) + p50 TTFT",
        "value": round(out_tokens / elapsed, 2),
        "unit": "output tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak" if (tp == 1 or args.tp_rank_of) else "strong",
        "vs_baseline": None,
        "dtype": {"fp8": "fp8", "int8": "int8"}.get(quant, "bf16"),
        "data": "synthetic",
        "config": {"workload": f"{cfg.name}-{quant} {WEIGHT_FORMAT.get(quant, quant)}: prefill {args.batch}x{args.input_len} "
                               f"in chunks of {args.chunk_seqs} seqs"
                               + (f" x {args.chunk_tokens} tokens (context > 0 from the second piece on)"
                                  if 0 < args.chunk_tokens < args.input_len else "")
                               + f" + {args.output_len - 1} graph-replayed decode steps (1 step = 1 whole job)",
                   "batch": args.batch, "global_batch": args.batch * replicas,
                   "input_len": args.input_len, "output_len": args.output_len,
                   "parallelism": par, "kv_block_size": 16,
                   "kv_cache_dtype": args.kv_cache_dtype,
                   "layers": cfg.layers,
                   "prefill_weight_image": ("bf16 operand image of the int4 weights, dequantised once at load "
                                            "(MI355X_PREPACK=1): the prefill GEMM multiplies bf16 x bf16, decode "
                                            "streams the int4 words" if image_on else
                                            ("operand image of the 8-bit weights (re-tiled once at load, same bytes "
                                             "again): the packed prefill GEMM reads it instead of re-tiling per call"
                                             if quant in ("fp8", "int8") and harness.QLinear.prepack
                                             and weights_gb["prefill_weight_images"] > 0 else "none")),
                   "hbm_weights_gb": weights_gb,
                   "outputs_finite": True,
                   "decode_graph": bool(graph_ok) and not args.no_graph,
                   "collectives": ("none on the data path" if tp == 1 else
                                   "RCCL all-reduce x2 per layer + all-gather of the logits, "
                                   + ("captured in the decode graph" if graph_ok and not args.no_graph
                                      else "issued eagerly (decode graph not captured)")
                                   + (" — on a 1-rank group: issued, nothing travels" if args.tp_rank_of else ""))},
        "ttft_p50_ms": round(ttft_p50, 2) if ttft_p50 is not None else None,
        "roofline": roofs[0] if roofs else None,
        "roofline_other": roofs[1:],
    }
    if args.layers:
        result["config"]["note"] = f"layer count overridden to {args.layers}: not a BASELINE configuration"
    if graph_err:
        result["config"]["decode_graph_error"] = graph_err
    if dp_extra is not None:
        result["dp_replicas"] = dp_extra
    if tp8_extra is not None:
        result["baseline_tp8_model"] = tp8_extra
    if surface is not None:
        result["plugin_surface"] = surface
    if world == 1 and not args.skip_cpu and not args.tp_rank_of:
        try:
            result["cpu_baseline"] = cpu_baseline(args, cfg)
        except Exception as e:  # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "unit": "output tokens/s", "cores": 0,
                                      "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(result), flush=True)
    if world > 1 or args.tp_rank_of:
        if world > 1:
            torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
