"""Isolate the --quant fp8 graph-replay defect: cutlass_scaled_mm (M = 64, split-K through a memset + atomics
workspace) captured into a HIP graph and replayed, against the eager result."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402

d = torch.device("cuda:0")
FP8 = torch.float8_e4m3fn
g = torch.Generator(device=d).manual_seed(0)
for (k, n) in [(4096, 6144), (4096, 28672), (14336, 4096)]:
    m = 64
    a = torch.randn(m, k, device=d, generator=g).to(FP8)
    b = torch.randn(n, k, device=d, generator=g).to(FP8).t()
    a_s = torch.rand(m, 1, device=d, generator=g) * 1e-2 + 1e-3
    b_s = torch.rand(1, n, device=d, generator=g) * 1e-2 + 1e-3
    ref = torch.empty(m, n, dtype=torch.bfloat16, device=d)
    ops.cutlass_scaled_mm(ref, a, b, a_s, b_s, None)
    torch.cuda.synchronize()
    out = torch.zeros_like(ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):     # several calls per graph, like the layers of a decode step
            ops.cutlass_scaled_mm(out, a, b, a_s, b_s, None)
    for r in range(4):
        out.zero_()
        gr.replay()
        torch.cuda.synchronize()
        diff = (out.float() - ref.float()).abs().max().item()
        print(f"k={k} n={n} replay {r}: max|out-ref| = {diff:.4g}  out absmax {out.float().abs().max().item():.4g} "
              f"ref absmax {ref.float().abs().max().item():.4g}", flush=True)

# ---- does a bare hipMemsetAsync node replay correctly?  (memset + a torch add into the same buffer, captured) ----
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
buf = torch.full((64, 6144), 7.0, device=d)
one = torch.ones(64, 6144, device=d)
res = torch.zeros(64, 6144, device=d)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    buf.add_(one)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(3):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, buf.numel() * 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        buf.add_(one)
        res.copy_(buf)
for r in range(4):
    gr.replay()
    torch.cuda.synchronize()
    print(f"memset-node check replay {r}: expect 1.0 everywhere, got min {res.min().item()} max {res.max().item()}",
          flush=True)
