"""Child process of tests/test_gpu_fp8_gemm.py::test_scaled_mm_fp8_operand_sources_agree: one m > 320 fp8
cutlass_scaled_mm under the MI355X_F8_ROWMAJOR / MI355X_F8_PACKED_SK the parent set (both read once per process),
output bits saved to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from vllm_metax_amd import _custom_ops as ops  # noqa: E402


def main():
    out_path = sys.argv[1]
    m, n, k = (int(x) for x in sys.argv[2:5])
    g = torch.Generator().manual_seed(m + n + k)
    d = torch.device("cuda:0")
    a = (torch.randn(m, k, generator=g) * 2).clamp(-448, 448).to(torch.float8_e4m3fn).to(d)
    b = (torch.randn(n, k, generator=g) * 2).clamp(-448, 448).to(torch.float8_e4m3fn).to(d).t()
    a_s = (torch.rand(m, 1, generator=g) * 9e-3 + 1e-3).to(d)
    b_s = (torch.rand(1, n, generator=g) * 9e-3 + 1e-3).to(d)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=d)
    ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
    torch.cuda.synchronize()
    np.save(out_path, out.view(torch.int16).cpu().numpy())


if __name__ == "__main__":
    main()
