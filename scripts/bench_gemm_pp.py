"""Prefill GEMM kernel alone (weights AND activations already operand images): down_proj of Llama-3-8B at M = 8192
fed by the gate_up SILU->image epilogue; gate_up itself with prepacked weights (+ its activation re-tiling launch).
Random vs all-zero operands (power -> clock).  usage: bench_gemm_pp.py  (MI355X_HOTPATH_LIB selects the build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
M, g, K, F = 8192, 128, 4096, 14336
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best
for mode in ("random", "zeros"):
    def mk(k, n):
        if mode == "random":
            qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=d).view(n, k // 8)
            qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), dtype=torch.int32, device=d)
        else:
            qw = torch.zeros(k // 8, n, dtype=torch.int32, device=d).view(n, k // 8)
            qz = torch.zeros(k // g, n // 8, dtype=torch.int32, device=d)
        sc = (torch.rand(k // g, n, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
        return ops.w4a16_prepack(qw, qz, sc, False)
    x = ((torch.randn(M, K, device=d) * 0.5) if mode == "random" else torch.zeros(M, K, device=d)).to(torch.bfloat16)
    gu, dn = mk(K, 2 * F), mk(F, K)
    act = ops.w4a16_gemm_prepacked(x, gu, 2 * F, K, silu=True, out_image=True)
    t = timed(lambda: ops.w4a16_gemm_prepacked(x, gu, 2 * F, K, silu=True, out_image=True))
    print(f"{mode:7s} gate_up+silu->image (pack_a + kernel): {t:8.1f} us  {2.0 * M * 2 * F * K / t / 1e6:7.1f} TFLOP/s", flush=True)
    t = timed(lambda: ops.w4a16_gemm_prepacked(act, dn, K, F))
    print(f"{mode:7s} down (kernel alone):                   {t:8.1f} us  {2.0 * M * F * K / t / 1e6:7.1f} TFLOP/s", flush=True)
