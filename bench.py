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
WEIGHT_FORMAT = {"awq": "w4a16 g128", "gptq": "w4a16 g128", "fp8": "w8a8 fp8", "int8": "w8a8 int8",
                 "none": "bf16"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--quant", default="awq", choices=["awq", "gptq", "fp8", "int8", "none"])
    ap.add_argument("--model", default="llama-3-8b", choices=["llama-3-8b", "tiny"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=128)
    ap.add_argument("--chunk-seqs", type=int, default=8)
    ap.add_argument("--parallelism", default=None, choices=["dp", "tp"],
                    help="N > 1: tp (default) = one model sharded over the N GPUs (column/row-parallel "
                         "GEMMs, RCCL all-reduce after o_proj / down_proj, strong scaling); dp = one model "
                         "replica per GPU, each serving its own batch of --batch sequences (no data-path "
                         "collective, weak scaling)")
    ap.add_argument("--no-dp-extra", action="store_true",
                    help="tp runs: skip the additional dp-replica measurement (the \"dp_replicas\" object)")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8"],
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

    def time(self, name, flops, nbytes, fn):
        if not self.enabled:
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.records.append((name, flops, nbytes, a, b))
        return out

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
        if out is None and timer.enabled and timer.records and timer.records[-1][0] == name:
            timer.records.pop()          # fused path not taken: the caller times the plain GEMM
        return out

    def timed_deferred(self, x):
        m = x.shape[0]
        if not (self.quant == "awq" and m <= 64):
            return orig_def(self, x)     # falls through to __call__, which is timed
        flops = 2.0 * m * self.n * self.k
        nbytes = self.weight_bytes() + 2.0 * m * self.k + 2.0 * m * self.n
        return timer.time(f"{self.quant}_gemm_small_m", flops, nbytes, lambda: orig_def(self, x))

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
            if record_as and (out is None or out is False) and timer.enabled and timer.records \
                    and timer.records[-1][0] == name:
                timer.records.pop()
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
                          sl, bs, max_len, partitioned):
        # the attention bytes (as paged_attention_v1 / _v2) + the qkv row / slabs of the folded qkv_rope_cache
        S, D = qkv.shape[0], kc.shape[2] * kc.shape[4]
        mean_len = model.mean_decode_len_for_cost
        nbytes = S * mean_len * kvh * D * 2 * kc.element_size() + 2.0 * S * nh * D * 2 \
            + qkv.numel() * (4.0 * sk if sk > 0 else 2.0)
        return 4.0 * S * mean_len * nh * D, nbytes
    wrap("paged_attention_fused_qkv", cost_decode_fused, record_as="paged_attention_v1")
    wrap("rms_norm_image", lambda x, w, eps: (0.0, float(x.numel() * x.element_size() * 2)), record_as="rms_norm")
    wrap("fused_add_rms_norm_image", lambda x, r, w, eps: (0.0, float(x.numel() * x.element_size() * 5)),
         record_as="fused_add_rms_norm")
    wrap("greedy_advance", lambda logits, *a, **k: (0.0, float(logits.numel() * logits.element_size())))
    wrap("rotary_reshape_and_cache",
         lambda pos, key, value, kc, *a, **k: (0.0, 2.0 * key.numel() * (2 + kc.element_size())),
         record_as="reshape_and_cache")
    wrap("paged_attention_v2", cost_decode)
    wrap("paged_attention_v1",
         lambda out, q, kc, vc, kvh, scale, bt, sl, bs, max_len, *a, **k:
         cost_decode(out, None, None, None, q, kc, vc, kvh, scale, bt, sl, bs, max_len))
    wrap("fused_add_rms_norm", cost_rows(3, 2))
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
    for c0 in range(0, B, args.chunk_seqs):
        ids = list(range(c0, min(c0 + args.chunk_seqs, B)))
        model.cfg_ctx_for_cost = 0
        nxt = model.prefill(tokens[ids[0]:ids[-1] + 1], ids, 0)
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


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (the
    parent never touches the GPU), relay rank 0's JSON line, return the worst exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    return max(abs(rc) for rc in rcs)


def timed_jobs(model, tokens, args, world, barrier, timer=None):
    """W warm-up jobs (at least one: it builds the decode graph), then EXACTLY K timed jobs bracketed by
    barrier + synchronize; returns (max-over-ranks seconds, ttft samples)."""
    for _ in range(max(args.warmup, 1)):
        run_job(model, tokens, args)
    barrier()
    if timer is not None:
        timer.enabled = True
    ttft_events, start_ev = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        tt = []
        run_job(model, tokens, args, ttft=tt)
        start_ev.append(s)
        ttft_events.append(tt)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, model.device, world)
    ttfts = []
    for s, tt in zip(start_ev, ttft_events):
        for n, ev in tt:
            ttfts += [s.elapsed_time(ev)] * n
    return elapsed, ttfts


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))      # (before any GPU call in this process)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}), "
                         f"or run bench.py --gpus N without a launcher")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # rehearsal on a 1-GPU box: BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU0=1 runs N ranks on cuda:0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)
        group = torch.distributed.group.WORLD

    from vllm_metax_amd import harness
    parallelism = args.parallelism or ("tp" if world > 1 else "dp")
    tp = world if parallelism == "tp" else 1

    def make_cfg(tp_degree):
        cfg = harness.ModelConfig.llama3_8b(args.quant) if args.model == "llama-3-8b" \
            else harness.ModelConfig.tiny(args.quant)
        cfg.tp, cfg.tp_rank = tp_degree, (rank if tp_degree > 1 else 0)
        cfg.kv_cache_dtype = args.kv_cache_dtype
        return cfg
    cfg = make_cfg(tp)
    max_len = args.input_len + args.output_len
    model = harness.HotPathModel(cfg, args.batch, max_len, device=f"cuda:{local_rank}", seed=0,
                                 tp_group=group if tp > 1 else None)
    model.setup_decode(args.batch, args.input_len, max_len)
    model.cfg_ctx_for_cost = 0
    model.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
    gen = torch.Generator(device=model.device).manual_seed(0)
    tokens = torch.randint(0, cfg.vocab, (args.batch, args.input_len), device=model.device, generator=gen)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # every hot op is wrapped so that a launch CAN be bracketed by HIP events; the timer is
    # off during warm-up / graph capture and on inside the timed region, where it sees the
    # eagerly launched kernels (all of prefill); graph-replayed decode launches are timed
    # afterwards by a few eager decode steps on the same stream.
    timer = EventTimer()
    instrument(model, timer)
    # ---- timed region: exactly K jobs -------------------------------------------------
    elapsed, ttfts = timed_jobs(model, tokens, args, world, barrier, timer)
    ttft_p50 = statistics.median(ttfts) if ttfts else None

    # ---- decode kernels: 8 eager decode steps at the mean decode context, bracketed ----------
    mid = args.input_len + (args.output_len - 1) // 2
    model.set_decode_lengths(torch.full((args.batch,), mid, device=model.device))
    model.mean_decode_len_for_cost = mid + 1 + 3.5
    for _ in range(8):
        model.decode_step(use_graph=False)
    torch.cuda.synchronize()
    timer.enabled = False
    agg = timer.summary()

    def roof(name, d):
        avg_ms = d["ms"] / d["launches"]
        if d["flops"] > 0 and name.endswith("large_m") or name == "paged_prefill_attention":
            ach = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
            peak = MFMA_8BIT_PEAK_TFLOPS if name.startswith(("fp8_gemm", "int8_gemm")) \
                else MFMA_BF16_PEAK_TFLOPS
            r = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2),
                 "peak": peak, "unit": "TFLOP/s",
                 "frac": round(ach / peak, 4), "traffic": None,
                 "avg_launch_us": round(avg_ms * 1e3, 2), "launches": d["launches"],
                 "total_ms": round(d["ms"], 2)}
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
                "total_ms": round(d["ms"], 2)}

    roofs = sorted((roof(n, d) for n, d in agg.items()), key=lambda r: -r["total_ms"])
    # HBM-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs of
    # the same kernels at the same shapes; a counter pass cannot run inside the timed region)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_pmc_traffic.json")
    if os.path.exists(tpath) and args.model == "llama-3-8b" and tp == 1 and args.kv_cache_dtype == "auto":
        with open(tpath) as f:
            traffic = json.load(f)
        for r in roofs:
            if r["kernel"] in traffic:
                r["traffic"] = traffic[r["kernel"]]
                r["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r02_pmc_summary.txt)"
    if args.kernel_stats and rank == 0:
        for r in roofs:
            print(json.dumps(r), file=sys.stderr)

    # ---- tp runs: the collective-free alternative (one replica per GPU) measured in the same run ----
    dp_extra = None
    if tp > 1 and not args.no_dp_extra:
        graph_ok, graph_err = model._graph not in (None, False), model.graph_error
        del model
        torch.cuda.empty_cache()
        rep_model = harness.HotPathModel(make_cfg(1), args.batch, max_len, device=f"cuda:{local_rank}", seed=0)
        rep_model.setup_decode(args.batch, args.input_len, max_len)
        rep_model.cfg_ctx_for_cost = 0
        rep_model.mean_decode_len_for_cost = args.input_len + (args.output_len - 1) / 2.0 + 1
        dp_elapsed, _ = timed_jobs(rep_model, tokens, args, world, barrier)
        dp_extra = {"value": round(whole_job_tokens(args.steps, args.batch, args.output_len, world, 1) / dp_elapsed, 2),
                    "unit": "output tokens/s", "scaling": "weak", "ms_per_step": round(dp_elapsed / args.steps * 1e3, 3),
                    "global_batch": args.batch * world,
                    "note": "one model replica per GPU, each serving its own batch: no data-path collective"}
    else:
        graph_ok, graph_err = model._graph not in (None, False), model.graph_error

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    replicas = world if tp == 1 else 1       # dp: every rank served its own batch
    out_tokens = whole_job_tokens(args.steps, args.batch, args.output_len, world, tp)
    result = {
        "metric": "output tokens/sec (Llama-3-8B AWQ-int4, batch 64, 1024-in/128-out) + p50 TTFT",
        "value": round(out_tokens / elapsed, 2),
        "unit": "output tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak" if tp == 1 else "strong",
        "vs_baseline": None,
        "dtype": {"fp8": "fp8", "int8": "int8"}.get(args.quant, "bf16"),
        "data": "synthetic",
        "config": {"workload": f"{cfg.name}-{args.quant} {WEIGHT_FORMAT.get(args.quant, args.quant)}: prefill {args.batch}x{args.input_len} "
                               f"in chunks of {args.chunk_seqs} seqs + {args.output_len - 1} graph-replayed "
                               f"decode steps (1 step = 1 whole job)",
                   "batch": args.batch, "global_batch": args.batch * replicas,
                   "input_len": args.input_len, "output_len": args.output_len,
                   "parallelism": f"tp{world}" if tp > 1 else f"dp{world}", "kv_block_size": 16,
                   "kv_cache_dtype": args.kv_cache_dtype,
                   "decode_graph": bool(graph_ok) and not args.no_graph,
                   "collectives": ("RCCL all-reduce x2 per layer + all-gather of the logits, captured in the "
                                   "decode graph" if tp > 1 else "none on the data path")},
        "ttft_p50_ms": round(ttft_p50, 2) if ttft_p50 is not None else None,
        "roofline": roofs[0] if roofs else None,
        "roofline_other": roofs[1:],
    }
    if graph_err:
        result["config"]["decode_graph_error"] = graph_err
    if dp_extra is not None:
        result["dp_replicas"] = dp_extra
    if world == 1 and not args.skip_cpu:
        try:
            result["cpu_baseline"] = cpu_baseline(args, cfg)
        except Exception as e:  # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "unit": "output tokens/s", "cores": 0,
                                      "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
