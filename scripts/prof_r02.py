"""Driver for the round-2 rocprofv3 passes: every hot kernel of the bench workload a few times at the
bench's shapes (Llama-3-8B AWQ g128, batch 64, ctx 1088), through the entry points bench.py uses now:
prefill GEMMs on the prepacked weight image, decode GEMMs through the deferred / SILU entries, decode
attention v1 with bf16 and fp8 (e4m3) KV cache.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_r02.py
(one counter group per pass: FETCH_SIZE, WRITE_SIZE, and the SQ group of scripts/prof_r02.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
g = 128
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
torch.manual_seed(0)
ws = torch.zeros(8 * 64 * 28672, dtype=torch.float32, device=d)
for name, K, N in shapes:
    qw = ops.awq_to_gptq_4bit(torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=d))
    qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
    sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
    img = ops.w4a16_prepack(qw, qz, sc)
    x = (torch.randn(8192, K, device=d) * 0.5).to(torch.bfloat16)
    for _ in range(2):
        if name == "gate_up":
            ops.w4a16_gemm_prepacked(x, img, N, K, silu=True, out_image=True)
        else:
            ops.w4a16_gemm_prepacked(x, img, N, K)
    x64 = x[:64].contiguous()
    for _ in range(2):
        if name == "gate_up":
            ops.awq_gemm_silu_mul(x64, qw, qz, sc)
        else:
            ops.awq_gemm_deferred(x64, qw, qz, sc, ws)
    torch.cuda.synchronize()
    del qw, qz, sc, img, x
S, H, KVH, D, BS, CTX = 64, 32, 8, 128, 16, 1088
nb = S * (CTX // BS)
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, CTX // BS)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
ks = torch.ones(1, device=d)
kc = torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16)
vc = torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16)
for _ in range(3):
    ops.paged_attention_v1(out, q, kc, vc, KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto", ks, ks)
kc8 = torch.randn(nb, KVH, D // 16, BS, 16, device=d).to(torch.float8_e4m3fn).view(torch.uint8)
vc8 = torch.randn(nb, KVH, D, BS, device=d).to(torch.float8_e4m3fn).view(torch.uint8)
for _ in range(3):
    ops.paged_attention_v1(out, q, kc8, vc8, KVH, D ** -0.5, bt, sl, BS, CTX, None, "fp8", ks, ks)
torch.cuda.synchronize()
print("done", flush=True)
