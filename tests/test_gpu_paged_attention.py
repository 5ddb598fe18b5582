"""GPU parity: paged_attention_v1 / v2 (decode) against the CPU oracle.

Grid after the reference's tests/kernels/attention/test_attention.py:29-47 (num_seqs 7,
heads (40,40) and (64,8), head sizes, block sizes 16/32, alibi on/off, three dtypes) plus
the BASELINE shapes scaled down (H/KVH = 32/8, L in {1, 15, 16, 17, 513, 1151}).

Tolerance: the oracle restates the kernel's rounding points (fp32 logits, probabilities
rounded to scalar_t before PV, fp32 accumulate); what remains is fp32 summation order
and exp() ulps, so the bound is max|err| <= 1e-3 * max|ref| + one output ulp
(north_star: <= 1e-3 rel; the reference's own test uses atol 1e-3 on outputs of
magnitude <= 0.09).
"""
import random

import pytest
import torch

from tests.util import assert_bit_exact, assert_close_rel, dev, make_kv_cache_x

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _setup(num_seqs, num_heads, num_kv_heads, head_size, block_size, dtype, seq_lens, seed=0,
           alibi=False, num_blocks=None):
    torch.manual_seed(seed)
    random.seed(seed)
    max_len = max(max(seq_lens), 1)
    max_blocks = (max_len + block_size - 1) // block_size
    if num_blocks is None:
        num_blocks = max(num_seqs * max_blocks + 3, 8)
    kc, vc = make_kv_cache_x(num_blocks, block_size, num_kv_heads, head_size, dtype, seed)
    scale = head_size ** -0.5
    q = (torch.rand(num_seqs, num_heads, head_size) * 2 - 1).mul(scale).to(dtype)
    perm = torch.randperm(num_blocks)
    bt = torch.zeros(num_seqs, max_blocks, dtype=torch.int32)
    for s in range(num_seqs):
        for b in range(max_blocks):
            bt[s, b] = int(perm[(s * max_blocks + b) % num_blocks])
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    slopes = torch.randn(num_heads, dtype=torch.float32) if alibi else None
    return q, kc, vc, bt, sl, slopes, scale, max_len


def _run_v1(q, kc, vc, bt, sl, slopes, scale, max_len, num_kv_heads, block_size):
    d = dev()
    out = torch.full_like(q, float("nan"), device=d)
    ops().paged_attention_v1(out, q.to(d), kc.to(d), vc.to(d), num_kv_heads, scale, bt.to(d),
                             sl.to(d), block_size, max_len,
                             slopes.to(d) if slopes is not None else None, "auto")
    torch.cuda.synchronize()
    return out


def _run_v2(q, kc, vc, bt, sl, slopes, scale, max_len, num_kv_heads, block_size):
    d = dev()
    S, H, D = q.shape
    P = (max_len + R.PARTITION_SIZE - 1) // R.PARTITION_SIZE
    out = torch.full_like(q, float("nan"), device=d)
    tmp = torch.empty(S, H, P, D, dtype=q.dtype, device=d)
    es = torch.empty(S, H, P, dtype=torch.float32, device=d)
    ml = torch.empty_like(es)
    ops().paged_attention_v2(out, es, ml, tmp, q.to(d), kc.to(d), vc.to(d), num_kv_heads, scale,
                             bt.to(d), sl.to(d), block_size, max_len,
                             slopes.to(d) if slopes is not None else None, "auto")
    torch.cuda.synchronize()
    return out, es, ml, tmp


def _tol(ref):
    # one unit in the last place of the output dtype at the largest magnitude
    eps = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2.0 ** -23}[ref.dtype]
    return eps * ref.float().abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("heads", [(40, 40), (64, 8), (32, 8), (10, 2)])
@pytest.mark.parametrize("head_size", [32, 80, 128, 256])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("alibi", [False, True])
def test_paged_attention_v1_v2_grid(dtype, heads, head_size, block_size, alibi):
    H, KVH = heads
    if head_size == 256 and H == 64:
        pytest.skip("oracle time")
    random.seed(0)
    seq_lens = [random.randint(1, 700) for _ in range(7)]
    seq_lens[-1] = 700
    args = _setup(7, H, KVH, head_size, block_size, dtype, seq_lens, alibi=alibi)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref1 = R.paged_attention_v1(q, kc, vc, KVH, scale, bt, sl, slopes)
    out1 = _run_v1(*args, KVH, block_size)
    assert_close_rel(out1, ref1, 1e-3, "v1", abs_floor=_tol(ref1))
    ref2, es_r, ml_r, tmp_r = R.paged_attention_v2(q, kc, vc, KVH, scale, bt, sl, max_len, slopes)
    out2, es, ml, tmp = _run_v2(*args, KVH, block_size)
    assert_close_rel(out2, ref2, 1e-3, "v2", abs_floor=_tol(ref2))
    # per-partition statistics of the partitions that exist
    for s in range(7):
        np_ = (seq_lens[s] + 511) // 512
        assert_close_rel(ml[s, :, :np_], ml_r[s, :, :np_], 1e-5, "max_logits", abs_floor=1e-5)
        assert_close_rel(es[s, :, :np_], es_r[s, :, :np_], 1e-4, "exp_sums")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("seq_lens", [[1, 15, 16, 17], [513, 512, 511, 1], [1151, 1024, 1088, 33]])
def test_paged_attention_llama3_shape_edges(dtype, block_size, seq_lens):
    """H/KVH = 32/8, d = 128 (Llama-3-8B): block- and partition-boundary lengths."""
    args = _setup(4, 32, 8, 128, block_size, dtype, seq_lens, seed=3)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref1 = R.paged_attention_v1(q, kc, vc, 8, scale, bt, sl, None)
    assert_close_rel(_run_v1(*args, 8, block_size), ref1, 1e-3, "v1", abs_floor=_tol(ref1))
    ref2 = R.paged_attention_v2(q, kc, vc, 8, scale, bt, sl, max_len, None)[0]
    assert_close_rel(_run_v2(*args, 8, block_size)[0], ref2, 1e-3, "v2", abs_floor=_tol(ref2))


def test_paged_attention_nan_in_unused_tail():
    """Slots past seq_len inside the last block may hold NaN garbage and must not leak
    (attention_kernels.cuh:409-419)."""
    args = _setup(3, 32, 8, 128, 16, torch.bfloat16, [5, 21, 530], seed=5)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref = R.paged_attention_v1(q, kc, vc, 8, scale, bt, sl, None)
    kc2, vc2 = kc.clone(), vc.clone()
    for s, L in enumerate([5, 21, 530]):
        blk = int(bt[s, (L - 1) // 16])
        off = L % 16
        if off:
            kc2[blk, :, :, off:, :] = float("nan")
            vc2[blk, :, :, off:] = float("nan")
    a2 = (q, kc2, vc2, bt, sl, slopes, scale, max_len)
    assert_close_rel(_run_v1(*a2, 8, 16), ref, 1e-3, "v1 nan tail", abs_floor=_tol(ref))
    ref2 = R.paged_attention_v2(q, kc, vc, 8, scale, bt, sl, max_len, None)[0]
    assert_close_rel(_run_v2(*a2, 8, 16)[0], ref2, 1e-3, "v2 nan tail", abs_floor=_tol(ref2))


def test_paged_attention_strided_query_and_errors():
    """query rows taken from a fused qkv tensor (q_stride != H*D); unsupported shapes raise."""
    args = _setup(5, 32, 8, 128, 16, torch.bfloat16, [100, 7, 64, 300, 17], seed=7)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref = R.paged_attention_v1(q, kc, vc, 8, scale, bt, sl, None)
    d = dev()
    qkv = torch.zeros(5, 48, 128, dtype=torch.bfloat16, device=d)
    qkv[:, :32] = q.to(d)
    out = torch.empty(5, 32, 128, dtype=torch.bfloat16, device=d)
    ops().paged_attention_v1(out, qkv[:, :32], kc.to(d), vc.to(d), 8, scale, bt.to(d), sl.to(d), 16,
                             max_len, None, "auto")
    assert_close_rel(out, ref, 1e-3, "strided q", abs_floor=_tol(ref))
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        bad = torch.empty(5, 32, 72, dtype=torch.bfloat16, device=d)
        ops().paged_attention_v1(bad, bad, kc.to(d), vc.to(d), 8, scale, bt.to(d), sl.to(d), 16,
                                 max_len, None, "auto")
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        ops().paged_attention_v1(out, qkv[:, :32], kc.to(d), vc.to(d), 8, scale, bt.to(d), sl.to(d), 64,
                                 max_len, None, "auto")
    with pytest.raises(RuntimeError):
        ops().paged_attention_v1(out, qkv[:, :32], kc.to(d), vc.to(d), 8, scale, bt.to(d), sl.to(d), 16,
                                 max_len, None, "fp8")


def test_paged_attention_full_bench_shape_properties():
    """BASELINE shape (64 seqs, 32/8 heads, d 128, L = 1024..1151): size-independent
    properties instead of the slow oracle — (a) v1 == v2 within tolerance, (b) output rows
    are convex combinations of V (|out| <= max|V|), (c) oracle spot-check of 2 sequences."""
    random.seed(1)
    seq_lens = [random.randint(1024, 1151) for _ in range(64)]
    args = _setup(64, 32, 8, 128, 16, torch.bfloat16, seq_lens, seed=11)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    out1 = _run_v1(*args, 8, 16).cpu()
    out2 = _run_v2(*args, 8, 16)[0].cpu()
    assert_close_rel(out2, out1, 2e-3, "v1 vs v2", abs_floor=_tol(out1))
    assert out1.float().abs().max().item() <= vc.float().abs().max().item() * 1.001
    pick = [0, 63]
    ref = R.paged_attention_v1(q[pick], kc, vc, 8, scale, bt[pick], sl[pick], None)
    assert_close_rel(out1[pick], ref, 1e-3, "spot check", abs_floor=_tol(ref))


@pytest.mark.parametrize("num_seqs", [24, 110])
def test_paged_attention_both_workgroup_sizes(num_seqs):
    """The launcher uses 8-wave workgroups when the grid has < 768 workgroups and 4-wave ones above:
    24 seqs x 8 kv heads = 192 (v1) / 384 (v2) and 110 x 8 = 880 / 1760 cover both on the
    Llama-3-8B head shape; results must not depend on it."""
    random.seed(3)
    seq_lens = [random.randint(1, 600) for _ in range(num_seqs)]
    seq_lens[0] = 600
    args = _setup(num_seqs, 32, 8, 128, 16, torch.bfloat16, seq_lens, seed=5)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref1 = R.paged_attention_v1(q, kc, vc, 8, scale, bt, sl, slopes)
    assert_close_rel(_run_v1(*args, 8, 16), ref1, 1e-3, "v1", abs_floor=_tol(ref1))
    ref2 = R.paged_attention_v2(q, kc, vc, 8, scale, bt, sl, max_len, slopes)[0]
    assert_close_rel(_run_v2(*args, 8, 16)[0], ref2, 1e-3, "v2", abs_floor=_tol(ref2))


def test_decode_attention_bench_shape_properties():
    """BASELINE decode shape (64 sequences, context 1024..1151, 32 / 8 heads, d 128, block 16) — too
    large for the python oracle, so size-independent properties: (a) sequences are independent (a
    sub-batch reproduces its rows bit for bit, v1 and v2); (b) v1 and v2 agree within the 1e-3
    budget; (c) permuting the sequences permutes the rows."""
    torch.manual_seed(13)
    d = dev()
    S, H, KVH, D, BS = 64, 32, 8, 128, 16
    seq_lens = [1024 + (i * 37) % 128 for i in range(S)]
    max_len = max(seq_lens)
    nblk = (max_len + BS - 1) // BS
    nb = S * nblk
    kc = (torch.randn(nb, KVH, D // 8, BS, 8, device=d) * 0.3).to(torch.bfloat16)
    vc = (torch.randn(nb, KVH, D, BS, device=d) * 0.3).to(torch.bfloat16)
    q = (torch.randn(S, H, D, device=d) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
    sl = torch.tensor(seq_lens, device=d, dtype=torch.int32)
    scale = D ** -0.5
    P = (max_len + 511) // 512

    def v1(qq, btt, sll):
        o = torch.empty_like(qq)
        ops().paged_attention_v1(o, qq, kc, vc, KVH, scale, btt, sll, BS, max_len, None, "auto")
        return o

    def v2(qq, btt, sll):
        n = qq.shape[0]
        o = torch.empty_like(qq)
        es = torch.empty(n, H, P, device=d, dtype=torch.float32)
        ml = torch.empty_like(es)
        tmp = torch.empty(n, H, P, D, device=d, dtype=torch.bfloat16)
        ops().paged_attention_v2(o, es, ml, tmp, qq, kc, vc, KVH, scale, btt, sll, BS, max_len, None, "auto")
        return o

    o1, o2 = v1(q, bt, sl), v2(q, bt, sl)
    assert torch.isfinite(o1.float()).all() and torch.isfinite(o2.float()).all()
    assert_close_rel(o1, o2.cpu(), 1e-3, "v1 vs v2", abs_floor=_tol(o2.cpu()))
    sub = torch.tensor([3, 17, 40, 63], device=d)
    assert torch.equal(v1(q[sub].contiguous(), bt[sub].contiguous(), sl[sub].contiguous()), o1[sub])
    # (the v2 sub-batch launch has < 768 workgroups and therefore 8-wave workgroups, the full batch
    #  4-wave ones: the cross-wave summation order differs, so equality is up to rounding there)
    assert_close_rel(v2(q[sub].contiguous(), bt[sub].contiguous(), sl[sub].contiguous()), o2[sub].cpu(),
                     1e-3, "v2 sub-batch", abs_floor=_tol(o2.cpu()))
    perm = torch.randperm(S, device=d)
    assert torch.equal(v1(q[perm].contiguous(), bt[perm].contiguous(), sl[perm].contiguous()), o1[perm])


@pytest.mark.parametrize("dtype,max_len,expect_v1", [
    (torch.bfloat16, 6656, True), (torch.bfloat16, 7000, False), (torch.bfloat16, 8192, False),
    (torch.float32, 4928, True), (torch.float32, 5100, False)])
def test_decode_attention_v1_v2_boundary(dtype, max_len, expect_v1):
    """Round-1 defect (VERDICT / ADVICE): the backend chose v1 for every max_seq_len <= 8192, but v1 keeps
    4 heads' logits in one workgroup's LDS and its launcher refuses contexts past 6656 (bf16) / 4928
    (fp32) at 32/8 heads, so decode_attention raised.  The choice now comes from the launcher's budget
    (mi355x_paged_attention_v1_max_seq_len) and falls to v2; both sides of the boundary match the oracle."""
    from vllm_metax_amd.attention import backend as B
    S, H, KVH, D, BS = 17, 32, 8, 128, 16                       # 17 x 32 > 512 (seq, head) pairs
    assert B.use_paged_attention_v1(S, H, max_len, KVH, D, BS, dtype) is expect_v1
    seq_lens = [1 + (i * 13) % 60 for i in range(S)]
    seq_lens[3] = max_len
    args = _setup(S, H, KVH, D, BS, dtype, seq_lens, seed=9, num_blocks=max_len // BS + 64)
    q, kc, vc, bt, sl, slopes, scale, _ = args
    d = dev()
    P = (max_len + 511) // 512
    out = torch.full_like(q, float("nan"), device=d)
    es = torch.empty(S, H, P, dtype=torch.float32, device=d)
    B.decode_attention(out, es, torch.empty_like(es), torch.empty(S, H, P, D, dtype=dtype, device=d), q.to(d),
                       kc.to(d), vc.to(d), KVH, scale, bt.to(d), sl.to(d), BS, max_len)
    torch.cuda.synchronize()
    rows = [0, 3, 16]                                            # oracle on the long row and two short ones
    ref = R.paged_attention_v1(q[rows], kc, vc, KVH, scale, bt[rows], sl[rows])
    assert_close_rel(out[rows], ref, 1e-3, f"decode_attention max_len {max_len}", abs_floor=_tol(ref))
    # and the launcher itself still refuses what does not fit (the message names the limit query)
    if not expect_v1:
        with pytest.raises(RuntimeError, match="paged_attention_v2"):
            ops().paged_attention_v1(out, q.to(d), kc.to(d), vc.to(d), KVH, scale, bt.to(d), sl.to(d), BS, max_len,
                                     None, "auto")


def test_paged_attention_tp8_per_rank_heads():
    """Per-rank head shape of Llama-3-70B / Qwen2-72B at TP=8: 8 query heads, ONE kv head
    (one workgroup serves all 8 since round 3), contexts across partition boundaries."""
    seq_lens = [1, 16, 511, 513, 1151, 2049]
    for dtype in (torch.bfloat16, torch.float16):
        args = _setup(len(seq_lens), 8, 1, 128, 16, dtype, seq_lens, seed=21)
        q, kc, vc, bt, sl, slopes, scale, max_len = args
        ref1 = R.paged_attention_v1(q, kc, vc, 1, scale, bt, sl, slopes)
        assert_close_rel(_run_v1(*args, 1, 16), ref1, 1e-3, "v1 8q/1kv", abs_floor=_tol(ref1))
        ref2 = R.paged_attention_v2(q, kc, vc, 1, scale, bt, sl, max_len, slopes)[0]
        assert_close_rel(_run_v2(*args, 1, 16)[0], ref2, 1e-3, "v2 8q/1kv", abs_floor=_tol(ref2))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("heads", [(8, 1), (32, 8), (10, 2), (16, 1)])
@pytest.mark.parametrize("ps", [128, 208, 288, 512])
def test_paged_attention_v2_partition_sizes(dtype, heads, ps):
    """mi355x_paged_attention_v2_ps: the v2 arithmetic (attention_kernels.cuh:519-658) with the caller's partition
    size against the oracle run with the same partition size — output and the per-partition statistics; head
    geometries: one TP = 8 rank (8 q / 1 kv: one 8-head workgroup), Llama-3-8B, 5 q per kv (8-head workgroup with
    absent heads), 16 q per kv (two 8-head tiles)."""
    H, KVH = heads
    seq_lens = [1, 16, ps - 1, ps, ps + 1, 1151, 2 * ps + 17]
    args = _setup(len(seq_lens), H, KVH, 128, 16, dtype, seq_lens, seed=5 + ps)
    q, kc, vc, bt, sl, slopes, scale, max_len = args
    ref, es_r, ml_r, tmp_r = R.paged_attention_v2(q, kc, vc, KVH, scale, bt, sl, max_len, slopes, ps)
    d = dev()
    S, _, D = q.shape
    P = (max_len + ps - 1) // ps
    out = torch.full_like(q, float("nan"), device=d)
    tmp = torch.empty(S, H, P, D, dtype=dtype, device=d)
    es = torch.empty(S, H, P, dtype=torch.float32, device=d)
    ml = torch.empty_like(es)
    ops().paged_attention_v2(out, es, ml, tmp, q.to(d), kc.to(d), vc.to(d), KVH, scale, bt.to(d), sl.to(d), 16,
                             max_len, None, "auto", partition_size=ps)
    torch.cuda.synchronize()
    assert_close_rel(out, ref, 1e-3, f"v2 ps={ps}", abs_floor=_tol(ref))
    for s in range(S):
        np_ = (seq_lens[s] + ps - 1) // ps
        assert_close_rel(ml[s, :, :np_], ml_r[s, :, :np_], 1e-5, "max_logits", abs_floor=1e-5)
        assert_close_rel(es[s, :, :np_], es_r[s, :, :np_], 1e-4, "exp_sums")
    with pytest.raises(RuntimeError, match="partition"):
        ops().paged_attention_v2(out, es, ml, tmp, q.to(d), kc.to(d), vc.to(d), KVH, scale, bt.to(d), sl.to(d), 16,
                                 max_len, None, "auto", partition_size=ps // 2)   # workspaces too small


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sk", [0, 3])
@pytest.mark.parametrize("partitioned", [False, True])
@pytest.mark.parametrize("geom", [(8, 2, 512), (8, 1, 512), (8, 1, 128), (8, 2, 272)])
def test_fused_qkv_attention_is_bit_identical(dtype, sk, partitioned, geom):
    """mi355x_paged_attention_fused_qkv == qkv_rope_cache followed by paged_attention_v1 / _v2, bit for bit:
    attention output and both caches (ragged lengths incl. a token that opens a new block, a sequence of one
    token, a context that spans several partitions).  Geometries: 4 query heads per kv head (Llama-3-8B) and
    8 per kv head = ONE workgroup for all heads of a TP = 8 rank of Llama-3-70B / Qwen2-72B (round 3), with the
    reference's 512-token partitions and with the finer ones decode_partition_size picks."""
    from vllm_metax_amd import _custom_ops as ops
    H, KVH, PS = geom
    D, BS = 128, 16
    if not partitioned and PS != 512:
        pytest.skip("partition size only matters for the partitioned launch")
    lens = [1, 16, 17, 33, 512, 513, 1100, 640]
    n = len(lens)
    g = torch.Generator().manual_seed(11 + sk)
    max_len = max(lens)
    nblk = (max_len + BS - 1) // BS
    nb = n * nblk + 3
    d = dev()
    kc = (torch.randn(nb, KVH, D // 8, BS, 8, generator=g) * 0.5).to(dtype).to(d)
    vc = (torch.randn(nb, KVH, D, BS, generator=g) * 0.5).to(dtype).to(d)
    bt = torch.randperm(nb, generator=g)[:n * nblk].to(torch.int32).view(n, nblk).to(d)
    sl = torch.tensor(lens, dtype=torch.int32, device=d)
    pos = (sl - 1).to(torch.int64)
    slots = (bt[torch.arange(n, device=d), (pos // BS)].long() * BS + pos % BS)
    width = (H + 2 * KVH) * D
    qkv = (torch.randn(n, width, generator=g) * 0.5).to(dtype).to(d)
    slabs = (torch.randn(max(sk, 1), n, width, generator=g) * 0.3).to(d)
    cos_sin = torch.randn(2048, D, generator=g).to(dtype).to(d)
    scale = D ** -0.5
    P = (max_len + PS - 1) // PS
    es = torch.empty(n, H, P, dtype=torch.float32, device=d)
    ml = torch.empty_like(es)
    tmp = torch.empty(n, H, P, D, dtype=dtype, device=d)

    def unfused():
        q2, k2, v2 = qkv.clone(), kc.clone(), vc.clone()
        ops.qkv_rope_cache(q2, slabs, sk, pos, cos_sin, k2, v2, slots, H, KVH, D)
        out = torch.empty(n, H, D, dtype=dtype, device=d)
        q3 = q2[:, :H * D].view(n, H, D)
        if partitioned:
            ops.paged_attention_v2(out, es, ml, tmp, q3, k2, v2, KVH, scale, bt, sl, BS, max_len, None,
                                   partition_size=PS)
        else:
            ops.paged_attention_v1(out, q3, k2, v2, KVH, scale, bt, sl, BS, max_len, None)
        return out, k2, v2

    ref_out, ref_k, ref_v = unfused()
    k3, v3 = kc.clone(), vc.clone()
    out = torch.empty(n, H, D, dtype=dtype, device=d)
    ok = ops.paged_attention_fused_qkv(out, es, ml, tmp, qkv.clone(), slabs, sk, pos, cos_sin, slots, k3, v3,
                                       H, KVH, scale, bt, sl, BS, max_len, partitioned, PS)
    assert ok
    assert_bit_exact(k3, ref_k, "key cache")
    assert_bit_exact(v3, ref_v, "value cache")
    assert_bit_exact(out, ref_out, "attention output")
    # not applicable (10 query heads per kv head -> two workgroups per kv head): reports False, launches nothing
    k4 = kc.clone()
    out10 = torch.empty(n, 10, D, dtype=dtype, device=d)
    es10 = torch.empty(n, 10, P, dtype=torch.float32, device=d)
    tmp10 = torch.empty(n, 10, P, D, dtype=dtype, device=d)
    qkv12 = torch.zeros(n, 12 * D, dtype=dtype, device=d)
    slabs12 = torch.zeros(max(sk, 1), n, 12 * D, device=d)
    assert not ops.paged_attention_fused_qkv(out10, es10, es10.clone(), tmp10, qkv12, slabs12, sk, pos, cos_sin,
                                             slots, k4[:, :1], v3[:, :1], 10, 1, scale, bt, sl, BS, max_len,
                                             partitioned, PS)
    assert_bit_exact(k4, kc, "untouched")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(8, 1, 288), (8, 2, 512), (32, 8, 512)])
@pytest.mark.parametrize("quant_out", [False, True])
def test_fused_qkv_w8_slabs_and_quantised_output_are_bit_identical(dtype, geom, quant_out):
    """mi355x_paged_attention_fused_qkv_w8 (fp8 model decode, round 3): the qkv row arrives as the split-K slabs of an
    fp8 GEMM with its scales still to apply, and the reduce launch may quantise the attention output per token.
    Against the op sequence it replaces: T(slab sum x scales) [= the GEMM's finish launch] -> fused_qkv on that row
    -> dynamic_per_token_scaled_fp8_quant."""
    from vllm_metax_amd import _custom_ops as ops
    H, KVH, PS = geom
    D, BS, sk = 128, 16, 3
    if quant_out and H > 16:
        pytest.skip("the reduce + quant kernel serves <= 16 heads (one wave per head)")
    lens = [1, 16, 17, 33, PS, PS + 1, 1100, 640]
    n = len(lens)
    g = torch.Generator().manual_seed(23)
    max_len = max(lens)
    nblk = (max_len + BS - 1) // BS
    nb = n * nblk + 3
    d = dev()
    kc = (torch.randn(nb, KVH, D // 8, BS, 8, generator=g) * 0.5).to(dtype).to(d)
    vc = (torch.randn(nb, KVH, D, BS, generator=g) * 0.5).to(dtype).to(d)
    bt = torch.randperm(nb, generator=g)[:n * nblk].to(torch.int32).view(n, nblk).to(d)
    sl = torch.tensor(lens, dtype=torch.int32, device=d)
    pos = (sl - 1).to(torch.int64)
    slots = (bt[torch.arange(n, device=d), (pos // BS)].long() * BS + pos % BS)
    width = (H + 2 * KVH) * D
    slabs = (torch.randn(sk, n, width, generator=g) * 30).to(d)
    a_s = (torch.rand(n, 1, generator=g) * 9e-3 + 1e-3).to(d)
    b_s = (torch.rand(1, width, generator=g) * 9e-3 + 1e-3).to(d)
    cos_sin = torch.randn(2048, D, generator=g).to(dtype).to(d)
    scale = D ** -0.5
    P = (max_len + PS - 1) // PS
    es = torch.empty(n, H, P, dtype=torch.float32, device=d)
    ml = torch.empty_like(es)
    tmp = torch.empty(n, H, P, D, dtype=dtype, device=d)
    # unfused: the finished qkv row, then the fused-qkv attention on it (itself tested against its op sequence), then quant
    acc = slabs[0].clone()
    for i in range(1, sk):
        acc += slabs[i]
    qkv_ref = (acc * a_s * b_s + 0.0).to(dtype)
    k2, v2 = kc.clone(), vc.clone()
    ref = torch.empty(n, H, D, dtype=dtype, device=d)
    assert ops.paged_attention_fused_qkv(ref, es, ml, tmp, qkv_ref, None, 0, pos, cos_sin, slots, k2, v2, H, KVH, scale,
                                         bt, sl, BS, max_len, True, PS)
    rq = torch.empty(n, H * D, dtype=torch.float8_e4m3fn, device=d)
    rs = torch.empty(n, 1, dtype=torch.float32, device=d)
    ops.dynamic_per_token_scaled_fp8_quant(rq, ref.view(n, H * D), rs, None)
    # fused
    k3, v3 = kc.clone(), vc.clone()
    out = torch.full((n, H, D), float("nan"), dtype=dtype, device=d)
    qo = (torch.empty_like(rq), torch.empty_like(rs)) if quant_out else None
    dummy = torch.empty(n, width, dtype=dtype, device=d)
    assert ops.paged_attention_fused_qkv(out, es, ml, tmp, dummy, slabs, sk, pos, cos_sin, slots, k3, v3, H, KVH, scale,
                                         bt, sl, BS, max_len, True, PS, slab_scales=(a_s, b_s), quant_out=qo)
    assert_bit_exact(k3, k2, "key cache")
    assert_bit_exact(v3, v2, "value cache")
    if quant_out:
        assert torch.equal(qo[1], rs)
        assert torch.equal(qo[0].view(torch.uint8), rq.view(torch.uint8))
    else:
        assert_bit_exact(out, ref, "attention output")
