"""Driver for profiling paged_prefill_attention at the bench's prefill chunk shape
(8 sequences x 1024 new tokens, no prior context, 32 q heads / 8 kv heads, d 128, block 16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
S, L, H, KVH, D, BS = 8, 1024, 32, 8, 128, 16
nblk = L // BS
nb = S * nblk
kc = torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16)
vc = torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16)
q = torch.randn(S * L, H, D, device=d).to(torch.bfloat16)
out = torch.empty_like(q)
bt = torch.arange(nb, device=d, dtype=torch.int32).view(S, nblk)
sl = torch.full((S,), L, device=d, dtype=torch.int32)
cu = (torch.arange(S + 1, device=d, dtype=torch.int32) * L)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(reps):
    ops.paged_prefill_attention(out, q, kc, vc, KVH, D ** -0.5, bt, sl, cu, L, BS)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    ops.paged_prefill_attention(out, q, kc, vc, KVH, D ** -0.5, bt, sl, cu, L, BS)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / reps
flops = 4.0 * S * L * (L / 2.0) * H * D
print(f"paged_prefill_attention: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s", flush=True)
