"""Micro-benchmark of the row-major norm family at prefill sizes (8192 tokens; hidden 4096 and 8192): rms_norm,
fused_add_rms_norm, rms_norm_dynamic_per_token_quant (fused add), GB/s of the bytes each moves.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
T = 8192
for hidden in (4096, 8192):
    x = torch.randn(T, hidden, device=d).to(torch.bfloat16)
    res = torch.randn(T, hidden, device=d).to(torch.bfloat16)
    w = torch.ones(hidden, dtype=torch.bfloat16, device=d)
    out = torch.empty_like(x)
    q = torch.empty(T, hidden, dtype=torch.float8_e4m3fn, device=d)
    sc = torch.empty(T, 1, dtype=torch.float32, device=d)
    cases = [("rms_norm", lambda: ops.rms_norm(out, x, w, 1e-5), 2 * x.numel() * 2),
             ("fused_add_rms_norm", lambda: ops.fused_add_rms_norm(x, res, w, 1e-5), 4 * x.numel() * 2),
             ("rms_norm_dynamic_per_token_quant(+res)", lambda: ops.rms_norm_dynamic_per_token_quant(q, x, w, sc, 1e-5, None, res),
              3 * x.numel() * 2 + x.numel())]
    for name, fn, nbytes in cases:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        print(f"hidden {hidden} {name:40s} {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s", flush=True)
