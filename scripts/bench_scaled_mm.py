"""Micro-benchmark of cutlass_scaled_mm (fp8 and int8 operands) at the Llama-3-8B layer shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
# --prepacked: the weights as their load-time operand image (scaled_mm_prepack / scaled_mm_prepacked)
PRE = "--prepacked" in sys.argv
Ms = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [64, 8192]
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
if "--70b-rank" in sys.argv:     # one TP = 8 rank of Llama-3-70B
    shapes = [("qkv", 8192, 1280), ("o", 1024, 8192), ("gate_up", 8192, 7168), ("down", 3584, 8192)]
KINDS = ("fp8",) if "--fp8" in sys.argv else ("fp8", "int8")
for kind in KINDS:
    for M in Ms:
        tot = 0.0
        for name, K, N in shapes:
            if kind == "fp8":
                a = torch.randn(M, K, device=d).to(torch.float8_e4m3fn)
                b = torch.randn(N, K, device=d).to(torch.float8_e4m3fn).t()
            else:
                a = torch.randint(-127, 128, (M, K), device=d, dtype=torch.int32).to(torch.int8)
                b = torch.randint(-127, 128, (N, K), device=d, dtype=torch.int32).to(torch.int8).t()
            a_s = torch.rand(M, 1, device=d) * 1e-2 + 1e-3
            b_s = torch.rand(1, N, device=d) * 1e-2 + 1e-3
            out = torch.empty(M, N, dtype=torch.bfloat16, device=d)
            img = ops.scaled_mm_prepack(b) if PRE and M > 320 else None
            if img is not None:
                call = lambda: ops.scaled_mm_prepacked(out, a, img, N, a_s, b_s, None)
            else:
                call = lambda: ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"{kind} M={M:5d} {name:8s}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} T(FL)OP/s  {N * K / us / 1e3:7.0f} GB/s(weights)", flush=True)
            tot += us
        print(f"{kind} M={M}: per-layer total {tot:.1f} us", flush=True)
