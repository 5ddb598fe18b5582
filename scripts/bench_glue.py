"""Decode-step glue kernels at the bench shapes (64 tokens, hidden 4096, 32/8 heads of 128): fused_add_rms_norm
on 4 / 8 split-K slabs and qkv_rope_cache on 4 slabs, each inside a HIP graph of 50 launches interleaved with a
cache-flushing copy is NOT done — the slabs are L2/MALL-resident as in the real step (written just before)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
T, H, NH, KVH, D, BS = 64, 4096, 32, 8, 128, 16
def timed(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best
x = torch.randn(T, H, device=d).to(torch.bfloat16)
res = torch.randn(T, H, device=d).to(torch.bfloat16)
w = torch.randn(H, device=d).to(torch.bfloat16)
for sk in (0, 4, 8):
    slabs = torch.randn(max(sk, 1), T, H, device=d) * 0.1
    print(f"fused_add_rms_norm_slabs sk={sk}: {timed(lambda: ops.fused_add_rms_norm_slabs(x, res, w, slabs, sk, 1e-5)):6.2f} us", flush=True)
width = (NH + 2 * KVH) * D
qkv = torch.randn(T, width, device=d).to(torch.bfloat16)
slabs = torch.randn(4, T, width, device=d) * 0.1
pos = torch.randint(0, 2048, (T,), device=d, dtype=torch.int64)
cs = torch.randn(4096, D, device=d).to(torch.bfloat16)
nb = 512
kc = torch.zeros(nb, KVH, D // 8, BS, 8, device=d, dtype=torch.bfloat16)
vc = torch.zeros(nb, KVH, D, BS, device=d, dtype=torch.bfloat16)
slots = torch.randperm(nb * BS, device=d)[:T].to(torch.int64)
for sk in (0, 4):
    print(f"qkv_rope_cache sk={sk}: {timed(lambda: ops.qkv_rope_cache(qkv, slabs, sk, pos, cs, kc, vc, slots, NH, KVH, D)):6.2f} us", flush=True)
out = torch.empty_like(x)
print(f"rms_norm: {timed(lambda: ops.rms_norm(out, x, w, 1e-5)):6.2f} us", flush=True)
