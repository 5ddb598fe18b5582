"""In-kernel cycle stamps of the decode stripe GEMM (build with -DSTRIPE_STAMP): average cycles per stage of
{memory cluster, wait at barrier 1, dequant + MFMA cluster, wait at barrier 2} for every wave of workgroups 0-7.
usage: MI355X_HOTPATH_LIB=variants/libstamp.so python scripts/stamp_stripe.py [N K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (28672, 4096)
M, g = 64, 128
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d).view(N, K // 8)
qz = torch.randint(-2**31, 2**31 - 1, (K // g, N // 8), dtype=torch.int32, device=d)
sc = (torch.rand(K // g, N, device=d) * 4e-3 + 1e-3).to(torch.bfloat16)
x = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16)
ws = torch.zeros(8 * M * N, dtype=torch.float32, device=d)
for _ in range(3):
    ops.awq_gemm(x, qw, qz, sc, 8, ws, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.awq_gemm(x, qw, qz, sc, 8, ws, True)
e1.record(); torch.cuda.synchronize()
print(f"kernel wall time (back to back, weights L2/MALL-warm): {e0.elapsed_time(e1) * 50:.1f} us")
t = ws[:8 * 16 * 8].view(8, 16, 8).cpu()
print("cycles per stage: [memory cluster, barrier 1, compute cluster, barrier 2]  (s_memtime runs at 100 MHz if these look ~20x small)")
for b in (0, 3):
    for w in range(16):
        if t[b, w].abs().sum() > 0:
            print(f"wg {b} wave {w:2d}: " + " ".join(f"{v:8.1f}" for v in t[b, w, :4].tolist()) + f"   stage {t[b, w, :4].sum():8.1f}   entry->loop {t[b, w, 4]:8.0f}  loop {t[b, w, 5]:8.0f}  reduction {t[b, w, 6]:7.0f}")
