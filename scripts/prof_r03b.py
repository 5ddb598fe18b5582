"""Second round-3 PMC driver: the kernels added later in the round, a few launches each — decode attention at the
per-rank head shape of a TP = 8 shard (8 q / 1 kv head, 64 seqs x 1088: one 8-head workgroup per kv head, 288-token
partitions) through the plain v2 entry and through the fused-qkv entry with the fp8 GEMM's slabs and the reduce + quant
launch; the 8-bit GEMM at M = 128 (passes) and M = 512 with the weight image.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_r03b.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402
d = torch.device("cuda:0")
torch.manual_seed(0)
FP8 = torch.float8_e4m3fn
S, H, KVH, D, BS, CTX, PS = 64, 8, 1, 128, 16, 1088, 288
nblk = (CTX + BS - 1) // BS
nb = S * nblk
q = torch.randn(S, H, D, device=d).to(torch.bfloat16)
bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
sl = torch.full((S,), CTX, device=d, dtype=torch.int32)
out = torch.empty_like(q)
kc = torch.randn(nb, KVH, D // 8, BS, 8, device=d).to(torch.bfloat16)
vc = torch.randn(nb, KVH, D, BS, device=d).to(torch.bfloat16)
P = (CTX + PS - 1) // PS
es = torch.empty(S, H, P, device=d, dtype=torch.float32)
ml = torch.empty_like(es)
tmp = torch.empty(S, H, P, D, device=d, dtype=torch.bfloat16)
for _ in range(3):
    ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, D ** -0.5, bt, sl, BS, CTX, None, "auto", partition_size=PS)
# fused qkv with an fp8 qkv GEMM's slabs (K = 8192 -> 1280 columns) and the quantising reduce
K, N = 8192, (H + 2 * KVH) * D
a = torch.randn(S, K, device=d).to(FP8)
b = torch.randn(N, K, device=d).to(FP8).t()
a_s = torch.rand(S, 1, device=d) * 1e-2 + 1e-3
b_s = torch.rand(1, N, device=d) * 1e-2 + 1e-3
qkv = torch.empty(S, N, dtype=torch.bfloat16, device=d)
ws = torch.empty(16 * S * N, dtype=torch.float32, device=d)
pos = (sl - 1).to(torch.int64)
slots = bt[torch.arange(S, device=d), (pos // BS)].long() * BS + pos % BS
cos_sin = torch.randn(2048, D, device=d).to(torch.bfloat16)
q8 = torch.empty(S, H * D, dtype=FP8, device=d)
qs = torch.empty(S, 1, dtype=torch.float32, device=d)
for _ in range(3):
    sk = ops.scaled_mm_fp8_deferred(qkv, a, b, a_s, b_s, ws)
    ops.paged_attention_fused_qkv(out, es, ml, tmp, qkv, ws if sk else None, sk, pos, cos_sin, slots, kc, vc, H, KVH,
                                  D ** -0.5, bt, sl, BS, CTX, True, PS, slab_scales=(a_s, b_s) if sk else None,
                                  quant_out=(q8, qs))
# 8-bit GEMM at mid-size M: passes of 64 rows (M = 128), packed kernel on the weight image (M = 512)
for K, N in ((4096, 6144), (14336, 4096)):
    bw = torch.randn(N, K, device=d).to(FP8).t()
    bs_ = torch.rand(1, N, device=d) * 1e-2 + 1e-3
    img = ops.scaled_mm_prepack(bw)
    for M in (128, 512):
        am = torch.randn(M, K, device=d).to(FP8)
        as_ = torch.rand(M, 1, device=d) * 1e-2 + 1e-3
        o = torch.empty(M, N, dtype=torch.bfloat16, device=d)
        for _ in range(2):
            if M > 320:
                ops.scaled_mm_prepacked(o, am, img, N, as_, bs_, None)
            else:
                ops.cutlass_scaled_mm(o, am, bw, as_, bs_, None)
torch.cuda.synchronize()
print("done", flush=True)
