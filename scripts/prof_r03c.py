"""Driver for the late-round-3 rocprofv3 PMC passes: the kernels changed after profiles/r03b_pmc_summary.txt — the fp8
prefill GEMM reading both operands in place (M = 8192, Llama-3-8B and 70B-rank shapes), its K split at M = 576, the
w4a16 image GEMM with the K split at M = 576, and the 8-bit decode GEMM with whole-line activation copies.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_r03c.py
(one counter group per pass: scripts/prof_r03c.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402

d = torch.device("cuda:0")
torch.manual_seed(0)
FP8 = torch.float8_e4m3fn
SH8B = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
SH70 = [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]
for M, shapes in ((8192, SH8B), (8192, SH70), (576, SH8B), (64, SH70)):
    for K, N in shapes:
        a = torch.randn(M, K, device=d).to(FP8)
        b = torch.randn(N, K, device=d).to(FP8).t()
        a_s = torch.rand(M, 1, device=d) * 1e-2 + 1e-3
        b_s = torch.rand(1, N, device=d) * 1e-2 + 1e-3
        out = torch.empty(M, N, dtype=torch.bfloat16, device=d)
        for _ in range(2):
            ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
        torch.cuda.synchronize()
        del a, b, out
g = 128
for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("down", 14336, 4096)]:
    qw = ops.awq_to_gptq_4bit(torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=d))
    qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
    sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
    img = ops.w4a16_prepack(qw, qz, sc)
    x = (torch.randn(576, K, device=d) * 0.5).to(torch.bfloat16)
    for _ in range(2):
        ops.w4a16_gemm_prepacked(x, img, N, K)
    torch.cuda.synchronize()
    del qw, qz, sc, img, x
print("done", flush=True)
