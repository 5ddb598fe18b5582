"""Bit-for-bit A/B of two builds of the library on the prefill GEMM (M = 8192, Llama-3-8B layer shapes, plain /
SILU / SILU->image epilogues): every output is hashed `reps` times per build; all hashes of a shape must agree
(the K order of an output element is fixed, so a schedule change must not move a single bit; a race shows up
as a hash that comes and goes).  usage: ab_exact.py variants/libX.so [reps]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    other, reps = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "6")
    outs = {}
    for tag, env in (("default", {}), ("variant", {"MI355X_HOTPATH_LIB": os.path.join(ROOT, other)})):
        r = subprocess.run([sys.executable, __file__, "--child", reps], env=dict(os.environ, **env), capture_output=True, text=True)
        if r.returncode:
            print(tag, "FAILED", r.stderr[-3000:]); sys.exit(1)
        outs[tag] = r.stdout.splitlines()
        print(f"[{tag}]"); print(r.stdout, flush=True)
    ok = outs["default"] == outs["variant"] and all("MISMATCH" not in l for l in outs["variant"])
    print("IDENTICAL" if ok else "DIFFERENT"); sys.exit(0 if ok else 1)
import torch
from vllm_metax_amd import _custom_ops as ops
reps = int(sys.argv[2])
d = torch.device("cuda:0")
g = 128
def sha(t): return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
gen = torch.Generator(device=d).manual_seed(7)
for M in (8192, 1300):
    for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096), ("odd", 1056, 1344)]:
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=d, generator=gen).view(N, K // 8)
        qz = torch.randint(-2**31, 2**31 - 1, (K // g if K % g == 0 else K // 32, N // 8), dtype=torch.int32, device=d, generator=gen)
        sc = (torch.rand(qz.shape[0], N, device=d, generator=gen) * 4e-3 + 1e-3).to(torch.bfloat16)
        x = (torch.randn(M, K, device=d, generator=gen) * 0.5).to(torch.bfloat16)
        img = ops.w4a16_prepack(qw, qz, sc, False)
        hs = set()
        for r in range(reps):
            h = [sha(ops.w4a16_gemm_prepacked(x, img, N, K)), sha(ops.awq_gemm(x, qw, qz, sc, 8, torch.empty(0), True))]
            if N % 256 == 0:
                h.append(sha(ops.w4a16_gemm_prepacked(x, img, N, K, silu=True)))
                p = ops.w4a16_gemm_prepacked(x, img, N, K, silu=True, out_image=True)
                h.append(sha(p.data if hasattr(p, "data") else p))
            hs.add(tuple(h))
        print(f"M={M} {name}: {'MISMATCH across reps ' if len(hs) != 1 else ''}{sorted(hs)[0]}", flush=True)
