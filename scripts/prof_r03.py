"""Driver for the round-3 rocprofv3 passes: the kernels that are NEW this round, a few launches each at the shapes
they serve — the streaming 8-bit decode GEMM (gemm8_decode_kernel) at the Llama-3-70B per-rank (TP=8) and the
Llama-3-8B shapes, the 128-row stripe passes of the mid-size w4a16 GEMM (MT = 8), the prefill attention with
sliding window / soft-cap / ALiBi on the MFMA kernel — plus the unchanged headline kernels for reference (prefill
GEMM on the weight image, decode stripe GEMM, decode attention).
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_r03.py
(one counter group per pass: scripts/prof_r03.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402

d = torch.device("cuda:0")
torch.manual_seed(0)
FP8 = torch.float8_e4m3fn
# ---- 8-bit decode GEMM, M = 64
for tag, shapes in (("70b/tp8", [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]),
                    ("8b", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])):
    for K, N in shapes:
        a = torch.randn(64, K, device=d).to(FP8)
        b = torch.randn(N, K, device=d).to(FP8).t()
        a_s = torch.rand(64, 1, device=d) * 1e-2 + 1e-3
        b_s = torch.rand(1, N, device=d) * 1e-2 + 1e-3
        out = torch.empty(64, N, dtype=torch.bfloat16, device=d)
        for _ in range(3):
            ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
        torch.cuda.synchronize()
# ---- w4a16: prefill image GEMM (M = 8192), decode stripe (M = 64), mid-size passes (M = 256, 512)
g = 128
ws = torch.zeros(8 * 64 * 28672, dtype=torch.float32, device=d)
for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
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
    for m in (256, 512):
        xm = x[:m].contiguous()
        for _ in range(2):
            ops.awq_gemm(xm, qw, qz, sc, 8, torch.empty(0), True)
    torch.cuda.synchronize()
    del qw, qz, sc, img, x
# ---- decode attention (bf16 cache) and prefill attention plain / with options
S, H, KVH, D, BS, CTX = 64, 32, 8, 128, 16, 1088
nb = S * (CTX // BS)
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, CTX // BS)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
kc = torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16)
vc = torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16)
for _ in range(3):
    ops.paged_attention_v1(out, q, kc, vc, KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto")
PS, L = 8, 1024
qp = (torch.randn(PS * L, H, D, device=d) * 0.5).to(torch.bfloat16)
op = torch.empty_like(qp)
cu = (torch.arange(PS + 1, dtype=torch.int32, device=d) * L)
slp = torch.full((PS,), L, dtype=torch.int32, device=d)
slopes = torch.tensor([2.0 ** (-8.0 * (h + 1) / H) for h in range(H)], dtype=torch.float32, device=d)
for kw in ({}, {"sliding_window": 256}, {"softcap": 50.0}, {"alibi_slopes": slopes}):
    for _ in range(2):
        ops.paged_prefill_attention(op, qp, kc, vc, KVH, D ** -0.5, bt[:PS, :L // BS].contiguous(), slp, cu, L, BS,
                                    "auto", None, None, kw.get("sliding_window"), kw.get("softcap"),
                                    kw.get("alibi_slopes"))
torch.cuda.synchronize()
print("done", flush=True)
