"""Micro-benchmark of decode paged attention at the per-rank head shape of a TP = 8 shard (Llama-3-70B /
Qwen2-72B: 8 query heads, ONE kv head, d 128) at batch 64, ctx ~1088: v1 and v2 over a sweep of partition sizes
(mi355x_paged_attention_v2_ps), HIP-graph of 30 launches over 3 cache copies.  Run on the GPU box.
usage: bench_attn_tp8.py [ctx] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_metax_amd import _custom_ops as ops
from vllm_metax_amd.attention.backend import decode_partition_size
d = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
CTX = int(args[0]) if args else 1088
S = int(args[1]) if len(args) > 1 else 64
H, KVH, D, BS = 8, 1, 128, 16
nblk = (CTX + BS - 1) // BS
nb = S * nblk
copies = 3
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
kcs = [torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16) for _ in range(copies)]
vcs = [torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16) for _ in range(copies)]
nbytes = S * CTX * KVH * D * 2 * 2


def timeit(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    reps = 30
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


us = timeit(lambda i: ops.paged_attention_v1(out, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto"))
print(f"v1 (one workgroup per sequence): {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
for ps in (512, 384, 288, 256, 192, 144, 128):
    P = (CTX + ps - 1) // ps
    es = torch.empty(S, H, P, device=d, dtype=torch.float32)
    ml = torch.empty_like(es)
    tmp = torch.empty(S, H, P, D, device=d, dtype=torch.bfloat16)
    us = timeit(lambda i: ops.paged_attention_v2(out, es, ml, tmp, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl,
                                                 BS, CTX, None, "auto", partition_size=ps))
    print(f"v2 partition {ps:4d} ({P:2d} parts, {S * P:4d} workgroups) + reduce: {us:7.1f} us  {nbytes / us / 1e3:7.0f} GB/s", flush=True)
print("decode_partition_size picks", decode_partition_size(S, H, KVH, max(CTX, 1152), BS))
