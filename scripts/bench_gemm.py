"""Micro-benchmark of the w4a16 GEMM at the Llama-3-8B layer shapes (run on the GPU box).
usage: python scripts/bench_gemm.py [M ...] [--image]   (default 64 and 8192; --image: the weights as their load-time
operand image, mi355x_w4a16_gemm_prepacked, for M >= W4_PREPACKED_MIN_M)"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
RESIDENT = "--resident" in sys.argv
IMAGE = "--image" in sys.argv
Ms = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [64, 8192]
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
g = 128
ONLY = [a.split("=")[1] for a in sys.argv if a.startswith("--only=")]
if ONLY:
    shapes = [s for s in shapes if s[0] in ONLY]
for M in Ms:
    tot = 0.0
    for name, K, N in shapes:
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d).view(N, K // 8)
        qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
        sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
        x = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16)
        ws = torch.zeros(8 * M * N, dtype=torch.float32, device=d) if M <= 64 else torch.empty(0)
        # rotate over 8 weight copies so that the weights are not L2/MALL resident
        copies = [qw.clone() for _ in range(8 if (M <= 64 and not RESIDENT) else 1)]
        if IMAGE and M >= ops.W4_PREPACKED_MIN_M:
            img = ops.w4a16_prepack(qw, qz, sc, False)
            call = lambda c: ops.w4a16_gemm_prepacked(x, img, N, K)
        else:
            call = lambda c: ops.awq_gemm(x, c, qz, sc, 8, ws, True)
        for c in copies[:2]:
            call(c)
        torch.cuda.synchronize()
        reps = 40 if M <= 64 else 10
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(reps):
                call(copies[i % len(copies)])
        gr.replay(); torch.cuda.synchronize()
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / reps
        wbytes = K * N / 2 + (K // g) * N * 2.5
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        print(f"M={M:5d} {name:8s} K={K:6d} N={N:6d}: {us:9.1f} us  {wbytes / (us * 1e-6) / 1e9:8.0f} GB/s(weights)  {tf:8.1f} TFLOP/s")
        tot += us
    print(f"M={M}: per-layer GEMM total {tot:.1f} us")
