"""Micro-benchmark of decode paged attention at the bench shape (64 seqs, ctx ~1088, 32/8 heads, d 128):
v1 (one workgroup per (seq, kv head)) vs v2 (512-token partitions + reduce).  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
S, H, KVH, D, BS = 64, 32, 8, 128, 16
CTX = int(sys.argv[1]) if len(sys.argv) > 1 else 1088
nblk = (CTX + BS - 1) // BS
nb = S * nblk
x = 8
copies = 3
kcs = [torch.randn(nb, KVH, D // x, BS, x, device=d).to(torch.bfloat16) for _ in range(copies)]
vcs = [torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16) for _ in range(copies)]
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
P = (CTX + 511) // 512
es = torch.empty(S, H, P, device=d, dtype=torch.float32)
ml = torch.empty_like(es)
tmp = torch.empty(S, H, P, D, device=d, dtype=torch.bfloat16)
ks = torch.ones(1, device=d)
def v1(i): ops.paged_attention_v1(out, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto", ks, ks)
def v2(i): ops.paged_attention_v2(out, es, ml, tmp, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto", ks, ks)
nbytes = S * CTX * KVH * D * 2 * 2
for name, fn in (("v1", v1), ("v2", v2)):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    reps = 30
    with torch.cuda.graph(g):
        for i in range(reps): fn(i)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"{name}: ctx={CTX} {us:7.1f} us per call  {nbytes / us / 1e3:7.0f} GB/s (K+V bytes)", flush=True)
