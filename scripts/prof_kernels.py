"""Driver for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass): launches each
hot kernel of the bench workload a few times at the bench's shapes (Llama-3-8B AWQ, batch 64).
usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_kernels.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
g = 128
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
torch.manual_seed(0)
for name, K, N in shapes:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d).view(N, K // 8)
    qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
    sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
    for M in (8192, 64):
        x = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16)
        ws = torch.zeros(8 * 64 * N, dtype=torch.float32, device=d) if M <= 64 else torch.empty(0)
        for _ in range(2):
            ops.awq_gemm(x, qw, qz, sc, 8, ws, True)
    torch.cuda.synchronize()
    del qw, qz, sc
# decode attention: 64 seqs, context 1088, 32 q heads / 8 kv heads, d = 128, block 16
S, H, KVH, D, BS, CTX = 64, 32, 8, 128, 16, 1088
nb = S * (CTX // BS)
x = 16 // 2
kc = torch.randn(nb, KVH, D // x, BS, x, device=d).to(torch.bfloat16)
vc = torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16)
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.arange(nb, device=d, dtype=torch.int32).view(S, CTX // BS)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
P = (CTX + 511) // 512
es = torch.empty(S, H, P, device=d, dtype=torch.float32)
ml = torch.empty_like(es)
tmp = torch.empty(S, H, P, D, device=d, dtype=torch.bfloat16)
ks = torch.ones(1, device=d)
for _ in range(3):
    ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto", ks, ks)
torch.cuda.synchronize()
print("done", flush=True)
