"""fp8 decode layer, op by op: the K split of each scaled GEMM reduced by its own finish launch (unfused) against the
consumer-side reduction (mi355x_scaled_mm_fp8_deferred + *_slabs consumers), HIP graph of 20 repetitions each, at the
Llama-3-8B shapes (TP = 1) and one TP = 8 rank of Llama-3-70B, M = 64.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
FP8 = torch.float8_e4m3fn
M = 64


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def gemm_case(K, N):
    a = torch.randn(M, K, device=d).to(FP8)
    b = torch.randn(N, K, device=d).to(FP8).t()
    a_s = torch.rand(M, 1, device=d) * 1e-2 + 1e-3
    b_s = torch.rand(1, N, device=d) * 1e-2 + 1e-3
    out = torch.empty(M, N, dtype=torch.bfloat16, device=d)
    ws = torch.empty(16 * M * N, dtype=torch.float32, device=d)
    return a, b, a_s, b_s, out, ws


for tag, hidden, ffn in (("llama-3-8b tp1", 4096, 14336), ("llama-3-70b rank of tp8", 8192, 3584)):
    print(f"== {tag}")
    # gate_up -> silu + quant
    a, b, a_s, b_s, out, ws = gemm_case(hidden, 2 * ffn)
    sk = ops.scaled_mm_fp8_deferred(out, a, b, a_s, b_s, ws)

    def unf():
        ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
        ops.silu_and_mul_per_token_quant(out)

    def fus():
        s = ops.scaled_mm_fp8_deferred(out, a, b, a_s, b_s, ws)
        if s > 0:
            ops.silu_and_mul_per_token_quant_slabs(ws, s, a_s, b_s, M, ffn, torch.bfloat16)
        else:
            ops.silu_and_mul_per_token_quant(out)
    print(f"gate_up [{hidden} -> {2 * ffn}] sk={sk}: GEMM(+finish) + silu_quant {timeit(unf):6.1f} us | deferred + silu_quant_slabs {timeit(fus):6.1f} us"
          f" | GEMM(+finish) alone {timeit(lambda: ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)):6.1f} | deferred alone {timeit(lambda: ops.scaled_mm_fp8_deferred(out, a, b, a_s, b_s, ws)):6.1f}")
    # down / o -> norm + quant (TP = 1 only; measured for both)
    for name, K in (("o_proj", hidden if "8b" in tag else 1024), ("down", ffn)):
        a, b, a_s, b_s, out, ws = gemm_case(K, hidden)
        w = torch.ones(hidden, dtype=torch.bfloat16, device=d)
        res = torch.randn(M, hidden, device=d).to(torch.bfloat16)
        q = torch.empty(M, hidden, dtype=FP8, device=d)
        sc = torch.empty(M, 1, dtype=torch.float32, device=d)
        sk = ops.scaled_mm_fp8_deferred(out, a, b, a_s, b_s, ws)

        def unf():
            ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
            ops.rms_norm_dynamic_per_token_quant(q, out, w, sc, 1e-5, None, res)

        def fus():
            s = ops.scaled_mm_fp8_deferred(out, a, b, a_s, b_s, ws)
            if s > 0:
                ops.rms_norm_dynamic_per_token_quant_slabs(q, ws, s, a_s, b_s, w, sc, 1e-5, None, res)
            else:
                ops.rms_norm_dynamic_per_token_quant(q, out, w, sc, 1e-5, None, res)
        print(f"{name} [{K} -> {hidden}] sk={sk}: GEMM(+finish) + norm_quant {timeit(unf):6.1f} us | deferred + norm_quant_slabs {timeit(fus):6.1f} us")
