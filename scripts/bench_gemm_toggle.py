"""Does operand bit toggling (power -> clock) limit the prefill GEMM?  Same launch, random vs
all-zero weights and activations (M = 8192, Llama-3-8B qkv and gate_up shapes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
M, g = 8192, 128
for name, K, N in (("qkv", 4096, 6144), ("gate_up", 4096, 28672)):
    for mode in ("random", "zeros"):
        if mode == "random":
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d).view(N, K // 8)
            qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
            x = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16)
        else:
            qw = torch.zeros(K // 8, N, dtype=torch.int32, device=d).view(N, K // 8)
            qz = torch.zeros(K // g, N // 8, dtype=torch.int32, device=d)
            x = torch.zeros(M, K, device=d, dtype=torch.bfloat16)
        sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
        for _ in range(3):
            ops.awq_gemm(x, qw, qz, sc, 8, torch.empty(0), True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.awq_gemm(x, qw, qz, sc, 8, torch.empty(0), True)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        print(f"{name:8s} {mode:7s}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
