"""Decode-sized (M = 64) cutlass_scaled_mm: per-call time from a HIP graph of 40 calls cycling over 8 weight
copies (nothing is served from the Infinity Cache that a decode step would not find there), plan sweep through
MI355X_F8_DECODE_FORCE=nt,sk.  usage: bench_scaled_mm_decode.py [fp8|int8] [70b|8b] [sweep]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402

d = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "fp8"
which = sys.argv[2] if len(sys.argv) > 2 else "70b"
sweep = len(sys.argv) > 3
SHAPES = {"70b": [("qkv", 8192, 1280), ("o", 1024, 8192), ("gate_up", 8192, 7168), ("down", 3584, 8192)],
          "8b": [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]}[which]
M, COPIES, CALLS = 64, 8, 40


def mk(K, N):
    if kind == "fp8":
        a = torch.randn(M, K, device=d).to(torch.float8_e4m3fn)
        bs = [torch.randn(N, K, device=d).to(torch.float8_e4m3fn).t() for _ in range(COPIES)]
    else:
        a = torch.randint(-127, 128, (M, K), device=d, dtype=torch.int32).to(torch.int8)
        bs = [torch.randint(-127, 128, (N, K), device=d, dtype=torch.int32).to(torch.int8).t() for _ in range(COPIES)]
    return a, bs, torch.rand(M, 1, device=d) * 1e-2 + 1e-3, torch.rand(1, N, device=d) * 1e-2 + 1e-3


def time_graph(a, bs, a_s, b_s, out):
    def body():
        for i in range(CALLS):
            ops.cutlass_scaled_mm(out, a, bs[i % COPIES], a_s, b_s, None)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * CALLS)


tot_new = tot_old = 0.0
for name, K, N in SHAPES:
    a, bs, a_s, b_s = mk(K, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=d)
    os.environ.pop("MI355X_F8_DECODE_FORCE", None)
    os.environ["MI355X_F8_DECODE_OLD"] = "1"
    t_old = time_graph(a, bs, a_s, b_s, out)
    ref = out.clone()
    del os.environ["MI355X_F8_DECODE_OLD"]
    t_new = time_graph(a, bs, a_s, b_s, out)
    diff = (out.float() - ref.float()).abs().max().item()
    print(f"{kind} {name:8s} K={K:5d} N={N:5d}: old {t_old:6.1f} us  new {t_new:6.1f} us ({N * K / t_new / 1e3:5.0f} GB/s)  "
          f"max|new-old| {diff:.3g}", flush=True)
    tot_new += t_new
    tot_old += t_old
    if sweep:
        for nt in (4, 2, 1):
            line = []
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                os.environ["MI355X_F8_DECODE_FORCE"] = f"{nt},{sk}"
                try:
                    line.append(f"sk{sk}:{time_graph(a, bs, a_s, b_s, out):5.1f}")
                except Exception as e:  # noqa: BLE001
                    line.append(f"sk{sk}: err")
            print(f"    nt={nt}: " + "  ".join(line), flush=True)
        os.environ.pop("MI355X_F8_DECODE_FORCE", None)
print(f"{kind} {which} per-layer total: old {tot_old:.1f} us, new {tot_new:.1f} us")
