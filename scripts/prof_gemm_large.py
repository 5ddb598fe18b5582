"""Single-shape driver for rocprofv3 PMC runs of the large-M w4a16 GEMM (qkv shape, M=8192)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
M, K, N, g = 8192, 4096, 6144, 128
if len(sys.argv) > 2:
    K, N = int(sys.argv[1]), int(sys.argv[2])
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d).view(N, K // 8)
qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
x = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16)
for i in range(6):
    ops.awq_gemm(x, qw, qz, sc, 8, torch.empty(0), True)
torch.cuda.synchronize()
