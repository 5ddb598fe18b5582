"""Micro-benchmark of decode paged attention at the bench shape (64 seqs, ctx ~1088, 32/8 heads, d 128):
v1 (one workgroup per (seq, kv head)) vs v2 (512-token partitions + reduce), bf16 and fp8 (e4m3) KV cache.
Run on the GPU box.  usage: bench_attn.py [ctx] [--ab LIB.so]  (--ab: also run the same benchmark in a child
process against another build of the library, e.g. variants/libpa_old.so, 3 alternating rounds)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--ab" in sys.argv:
    other = sys.argv[sys.argv.index("--ab") + 1]
    args = [a for a in args if a != other]
    for r in range(3):
        for tag, env in (("new", {}), ("old", {"MI355X_HOTPATH_LIB": os.path.join(ROOT, other)})):
            out = subprocess.run([sys.executable, __file__] + args, env=dict(os.environ, **env), capture_output=True, text=True)
            for line in out.stdout.splitlines():
                print(f"[{tag} round {r}] {line}", flush=True)
            if out.returncode:
                print(out.stderr[-2000:])
    sys.exit(0)
import torch
from vllm_metax_amd import _custom_ops as ops
d = torch.device("cuda:0")
S, H, KVH, D, BS = 64, 32, 8, 128, 16
CTX = int(args[0]) if args else 1088
nblk = (CTX + BS - 1) // BS
nb = S * nblk
copies = 3
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
P = (CTX + 511) // 512
es = torch.empty(S, H, P, device=d, dtype=torch.float32)
ml = torch.empty_like(es)
tmp = torch.empty(S, H, P, D, device=d, dtype=torch.bfloat16)
ks = torch.ones(1, device=d)
for kvname in ("auto", "fp8"):
    if kvname == "auto":
        kcs = [torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16) for _ in range(copies)]
        vcs = [torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16) for _ in range(copies)]
    else:
        kcs = [torch.randn(nb, KVH, D // 16, BS, 16, device=d).to(torch.float8_e4m3fn).view(torch.uint8) for _ in range(copies)]
        vcs = [torch.randn(nb, KVH, D, BS, device=d).to(torch.float8_e4m3fn).view(torch.uint8) for _ in range(copies)]
    def v1(i): ops.paged_attention_v1(out, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl, BS, CTX, None, kvname, ks, ks)
    def v2(i): ops.paged_attention_v2(out, es, ml, tmp, q, kcs[i % copies], vcs[i % copies], KVH, D ** -0.5, bt, sl, BS, CTX, None, kvname, ks, ks)
    nbytes = S * CTX * KVH * D * 2 * kcs[0].element_size()
    for name, fn in (("v1", v1), ("v2", v2)):
        try:
            for i in range(3): fn(i)
        except RuntimeError as e:
            print(f"{name} kv={kvname}: not supported by this build ({str(e)[:60]})", flush=True)
            continue
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        reps = 30
        with torch.cuda.graph(g):
            for i in range(reps): fn(i)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / reps)
        print(f"{name} kv={kvname}: ctx={CTX} {best:7.1f} us per call  {nbytes / best / 1e3:7.0f} GB/s (K+V bytes)", flush=True)
