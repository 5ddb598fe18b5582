"""A/B of the prefill MLP pair at the Llama-3-8B shapes (M = 8192): gate_up + silu (row-major act)
-> down (with its re-tiling launch)  vs  gate_up + silu written as the operand image -> down."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
M, H, F, g = 8192, 4096, 14336, 128


def mk(k, n):
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32, device=d)
    qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), dtype=torch.int32, device=d)
    sc = (torch.rand(k // g, n, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
    return ops.awq_to_gptq_4bit(qw), qz, sc


gu, dn = mk(H, 2 * F), mk(F, H)
x = (torch.randn(M, H, device=d) * 0.5).to(torch.bfloat16)


def plain():
    act = ops.awq_gemm_silu_mul(x, *gu)
    return ops.awq_gemm(act, dn[0], dn[1], dn[2], 8, torch.empty(0), True)


def packed():
    act = ops.awq_gemm_silu_mul_packed(x, *gu)
    return ops.awq_gemm_packed_a(act, *dn)


assert torch.equal(plain(), packed())
for rnd in range(3):
    for name, fn in (("plain", plain), ("packed", packed)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {name:7s}: {e0.elapsed_time(e1) * 100:8.1f} us per gate_up+down pair", flush=True)
