"""Diagnostic: how exact is v_mfma_f32_16x16x32_fp8_fp8 accumulation?  (run on the GPU box)"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
from oracle import ref_ops as R
FP8 = torch.float8_e4m3fn
d = torch.device("cuda:0")
torch.manual_seed(0)
for kind in ("int", "randn"):
    for k in (64, 512, 4096):
        m, n = 128, 256
        if kind == "int":
            a = torch.randint(-8, 9, (m, k)).float().to(FP8)
            b = torch.randint(-8, 9, (n, k)).float().to(FP8).t()
        else:
            a = (torch.randn(m, k) * 2).to(FP8)
            b = (torch.randn(n, k) * 2).to(FP8).t()
        s = torch.tensor([[1.0 / 64]], dtype=torch.float32)
        one = torch.ones(1, 1, dtype=torch.float32)
        ref64 = (a.float().double() @ b.float().double()) / 64
        out = torch.empty(m, n, dtype=torch.float16, device=d)
        ops.cutlass_scaled_mm(out, a.to(d), b.t().contiguous().to(d).t(), s.to(d), one.to(d), None)
        got = out.cpu().double()
        ref16 = ref64.to(torch.float16).double()
        rel = ((got - ref64).abs() / ref64.abs().clamp(min=1e-3)).max().item()
        frac = (got != ref16).double().mean().item()
        print(f"{kind:6s} k={k:5d}  frac_differ_vs_exact_rounded={frac:.4f}  max_rel_err_vs_exact={rel:.3e}")
