"""GPU parity: paged prefill / chunked prefill attention against the CPU oracle.

Batch specs after the reference's backend-level test tests/v1/attention/
test_attention_backends.py:47-70 (prefill and mixed batches, query_len << / == seq_len) and
tests/kernels/attention/test_flash_attn.py (varlen, paged KV, GQA, causal bottom-right).
Tolerance: probabilities are rounded to the value dtype in both (oracle: normalised,
kernel: flash-style unnormalised) so outputs differ by rounding noise only:
one output ulp + 2e-3 * max|ref| (reference tests: atol 1.5e-2 / rtol 1e-2).
"""
import pytest
import torch

from tests.util import assert_close_rel, dev, make_kv_cache_x

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _run(q_lens, seq_lens, H, KVH, D, bs, dtype, seed=0, poison_tail=False, window=None, softcap=None, alibi=False,
         kv_scale=1.0):
    torch.manual_seed(seed)
    S = len(q_lens)
    max_blocks = (max(seq_lens) + bs - 1) // bs
    nb = S * max_blocks + 5
    kc, vc = make_kv_cache_x(nb, bs, KVH, D, dtype, seed)
    perm = torch.randperm(nb)
    bt = torch.zeros(S, max_blocks, dtype=torch.int32)
    for s in range(S):
        for b in range(max_blocks):
            bt[s, b] = int(perm[s * max_blocks + b])
    cu = torch.zeros(S + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(q_lens).cumsum(0)
    T = int(cu[-1])
    scale = D ** -0.5
    q = (torch.randn(T, H, D) * 0.5 * kv_scale).to(dtype)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    slopes = None
    if alibi:      # geometric slopes, as ALiBi models use them
        slopes = torch.tensor([2.0 ** (-8.0 * (h + 1) / H) for h in range(H)], dtype=torch.float32)
    ref = R.paged_prefill_attention(q, kc, vc, KVH, scale, bt, sl, cu, window, softcap, slopes)
    if poison_tail:
        for s, L in enumerate(seq_lens):
            if L % bs:
                blk = int(bt[s, (L - 1) // bs])
                kc[blk, :, :, L % bs:, :] = float("nan")
                vc[blk, :, :, L % bs:] = float("nan")
    d = dev()
    out = torch.full((T, H, D), float("nan"), dtype=dtype, device=d)
    ops().paged_prefill_attention(out, q.to(d), kc.to(d), vc.to(d), KVH, scale, bt.to(d), sl.to(d),
                                  cu.to(d), max(q_lens), bs, "auto", None, None, window, softcap,
                                  slopes.to(d) if slopes is not None else None)
    torch.cuda.synchronize()
    eps = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2.0 ** -23}[dtype]
    assert_close_rel(out, ref, 2e-3, "prefill", abs_floor=eps * ref.float().abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("spec", [
    ([1], [1]), ([5], [5]), ([16], [16]), ([17], [40]), ([128], [128]), ([129], [129]),
    ([130, 7, 64], [130, 300, 64]),                   # pure prefill + chunked (context > 0)
    ([32, 1, 200], [1000, 77, 200]),                  # mixed: long context, decode-like row
    ([300], [813]),
])
def test_prefill_llama_heads(dtype, spec):
    q_lens, seq_lens = spec
    _run(q_lens, seq_lens, 8, 2, 128, 16, dtype, poison_tail=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("window,softcap,alibi", [(256, None, False), (None, 50.0, False), (256, 50.0, False),
                                                  (None, None, True), (100, 30.0, True), (1, None, False),
                                                  (17, None, False), (4096, None, False)])
@pytest.mark.parametrize("spec", [
    ([130, 7, 64], [130, 300, 64]),                   # pure prefill + chunked (context > 0)
    ([300], [813]),                                   # window edge inside the context
    ([32, 1, 200], [1000, 77, 200]),                  # long context: whole key stages below the window are skipped
    ([513], [513]),
])
def test_prefill_window_softcap_alibi_on_the_mfma_kernel(dtype, window, softcap, alibi, spec):
    """The optional arguments of the reference call site (flash_attn.py:725-747: window_size, softcap,
    alibi_slopes) on the MFMA kernel (head 128, block 16): sliding windows from 1 key to longer than the
    sequence, soft-cap, ALiBi and their combinations, with context > 0 and poisoned cache tails.  Scores are made
    large enough (q scaled up) for the cap to bite.  Window / cap semantics are pinned by the reference-generated
    fixture ref_flash_paged_opts (tests/test_gpu_ref_fixtures.py); ALiBi on the prefill path is parity unpinned
    (the reference's prefill oracle has no ALiBi case; the bias is the decode kernel's)."""
    q_lens, seq_lens = spec
    _run(q_lens, seq_lens, 8, 2, 128, 16, dtype, seed=4, poison_tail=True, window=window, softcap=softcap,
         alibi=alibi, kv_scale=4.0 if softcap else 1.0)


def test_prefill_opts_generic_shapes_still_served():
    """head sizes / block sizes the MFMA kernel does not take keep the general kernel for window / soft-cap."""
    _run([9, 33], [20, 33], 4, 2, 64, 16, torch.bfloat16, seed=3, window=8, softcap=20.0)
    with pytest.raises(RuntimeError):
        _run([9], [20], 4, 2, 64, 16, torch.bfloat16, seed=3, alibi=True)


def test_prefill_gqa_32_8_chunk():
    """Llama-3-8B heads (32/8): one 512-token chunk with 256 tokens of context."""
    _run([512, 100], [768, 100], 32, 8, 128, 16, torch.bfloat16, seed=2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("D,bs", [(64, 16), (80, 8), (128, 32), (256, 16)])
def test_prefill_generic_path(dtype, D, bs):
    _run([9, 33], [20, 33], 4, 2, D, bs, dtype, seed=3)


def test_prefill_matches_decode_kernel_on_last_token():
    """Size-independent cross-check at the bench shape (8 seqs x 1024 new tokens): the last
    query row of each sequence must equal paged_attention_v1 on the same cache."""
    torch.manual_seed(5)
    S, H, KVH, D, bs, L = 8, 32, 8, 128, 16, 1024
    dtype = torch.bfloat16
    d = dev()
    nblk = L // bs
    kc, vc = make_kv_cache_x(S * nblk, bs, KVH, D, dtype, 5)
    kc, vc = kc.to(d), vc.to(d)
    bt = torch.randperm(S * nblk).to(torch.int32).reshape(S, nblk).to(d)
    cu = (torch.arange(S + 1, dtype=torch.int32) * L).to(d)
    sl = torch.full((S,), L, dtype=torch.int32, device=d)
    q = (torch.randn(S * L, H, D) * 0.5).to(dtype).to(d)
    out = torch.empty_like(q)
    scale = D ** -0.5
    ops().paged_prefill_attention(out, q, kc, vc, KVH, scale, bt, sl, cu, L, bs)
    last = torch.arange(S, device=d) * L + (L - 1)
    dec = torch.empty(S, H, D, dtype=dtype, device=d)
    ops().paged_attention_v1(dec, q[last].contiguous(), kc, vc, KVH, scale, bt, sl, bs, L, None, "auto")
    eps = 2.0 ** -7
    assert_close_rel(out[last], dec, 2e-3, "prefill vs decode", abs_floor=eps * dec.float().abs().max().item())
    assert torch.isfinite(out.float()).all()


def test_prefill_bench_chunk_properties_at_full_size():
    """BASELINE chunk shape (8 sequences x 1024 new tokens, 32 / 8 heads, d 128) — too large for the
    oracle, so size-independent properties: (a) sequences are independent: the batched launch equals
    eight single-sequence launches bit for bit; (b) causality: the first 512 rows of a 1024-token
    prefill equal a 512-token prefill of the same sequence bit for bit; (c) the launch order
    (heaviest query block first) does not leak into the result: permuting the sequences permutes the
    output."""
    torch.manual_seed(11)
    d = dev()
    S, L, H, KVH, D, BS = 8, 1024, 32, 8, 128, 16
    nblk = L // BS
    nb = S * nblk
    kc = (torch.randn(nb, KVH, D // 8, BS, 8, device=d) * 0.3).to(torch.bfloat16)
    vc = (torch.randn(nb, KVH, D, BS, device=d) * 0.3).to(torch.bfloat16)
    q = (torch.randn(S * L, H, D, device=d) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(nb, device=d).to(torch.int32).view(S, nblk)
    sl = torch.full((S,), L, device=d, dtype=torch.int32)
    cu = torch.arange(S + 1, device=d, dtype=torch.int32) * L
    scale = D ** -0.5
    out = torch.empty_like(q)
    ops().paged_prefill_attention(out, q, kc, vc, KVH, scale, bt, sl, cu, L, BS)
    assert torch.isfinite(out.float()).all()
    one_cu = torch.tensor([0, L], device=d, dtype=torch.int32)
    for s in (0, 3, 7):
        o1 = torch.empty(L, H, D, dtype=torch.bfloat16, device=d)
        ops().paged_prefill_attention(o1, q[s * L:(s + 1) * L].contiguous(), kc, vc, KVH, scale,
                                      bt[s:s + 1].contiguous(), sl[:1], one_cu, L, BS)
        assert torch.equal(o1, out[s * L:(s + 1) * L]), "batch independence"
    half = torch.empty(512, H, D, dtype=torch.bfloat16, device=d)
    ops().paged_prefill_attention(half, q[:512].contiguous(), kc, vc, KVH, scale, bt[:1].contiguous(),
                                  torch.tensor([512], device=d, dtype=torch.int32),
                                  torch.tensor([0, 512], device=d, dtype=torch.int32), 512, BS)
    assert torch.equal(half, out[:512]), "causal prefix"
    perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device=d)
    qp = q.view(S, L, H, D)[perm].reshape(S * L, H, D).contiguous()
    outp = torch.empty_like(q)
    ops().paged_prefill_attention(outp, qp, kc, vc, KVH, scale, bt[perm].contiguous(), sl, cu, L, BS)
    assert torch.equal(outp.view(S, L, H, D), out.view(S, L, H, D)[perm]), "sequence permutation"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lens,ctx", [([128, 300, 77, 1], 0), ([64, 200], 48)])
def test_prefill_attention_image_is_bit_identical(dtype, lens, ctx):
    """MI355X-side prefill fusion: the attention output written as o_proj's operand image == the row-major output
    re-tiled, bit for bit (ragged query lengths, a total that is not a multiple of 16, cached context)."""
    from vllm_metax_amd import _custom_ops as ops
    H, KVH, D, BS = 8, 2, 128, 16
    g = torch.Generator().manual_seed(len(lens) + ctx)
    n = len(lens)
    seq = [ctx + l for l in lens]
    nblk = (max(seq) + BS - 1) // BS
    nb = n * nblk + 2
    d = dev()
    kc = (torch.randn(nb, KVH, D // 8, BS, 8, generator=g) * 0.5).to(dtype).to(d)
    vc = (torch.randn(nb, KVH, D, BS, generator=g) * 0.5).to(dtype).to(d)
    bt = torch.randperm(nb, generator=g)[:n * nblk].to(torch.int32).view(n, nblk).to(d)
    T = sum(lens)
    q = (torch.randn(T, H, D, generator=g) * 0.5).to(dtype).to(d)
    sl = torch.tensor(seq, dtype=torch.int32, device=d)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=d)
    out = torch.empty_like(q)
    ops.paged_prefill_attention(out, q, kc, vc, KVH, D ** -0.5, bt, sl, cu, max(lens), BS)
    img = ops.paged_prefill_attention_image(q, kc, vc, KVH, D ** -0.5, bt, sl, cu, max(lens), BS)
    assert img is not None and img.shape == (T, H * D)
    mt, kt = (T + 15) // 16, H * D // 32
    pad = torch.zeros(mt * 16, H * D, dtype=dtype, device=d)
    pad[:T] = out.view(T, H * D)
    t5 = pad.view(mt, 16, kt, 4, 8)
    exp = torch.empty(mt, kt, 64, 8, dtype=dtype, device=d)
    for lr in range(4):
        gx = ((lr & 1) * 12) | (lr & 2)
        for lc in range(16):
            exp[:, :, lr * 16 + (lc ^ gx)] = t5[:, lc, :, lr]
    assert torch.equal(img.data.view(mt, kt, 64, 8).view(torch.int16), exp.view(torch.int16))
    # query rotary applied while the rows are loaded == rotary_embedding on q first
    pos = torch.randint(0, 4096, (T,), generator=g).to(d)
    cos_sin = torch.randn(4096, D, generator=g).to(dtype).to(d)
    q_rot = q.clone()
    ops.rotary_embedding(pos, q_rot.view(T, H * D), None, D, cos_sin, True)
    img_a = ops.paged_prefill_attention_image(q_rot, kc, vc, KVH, D ** -0.5, bt, sl, cu, max(lens), BS)
    img_b = ops.paged_prefill_attention_image(q, kc, vc, KVH, D ** -0.5, bt, sl, cu, max(lens), BS,
                                              positions=pos, cos_sin_cache=cos_sin)
    assert torch.equal(img_a.data.view(torch.int16), img_b.data.view(torch.int16))


@pytest.mark.parametrize("kind", ["plain", "opts"])
def test_prefill_query_tile_shapes_agree(kind, tmp_path):
    """The MFMA prefill kernel with 128 and with 64 query rows per workgroup (the launcher picks 64 for grids below two
    workgroups per CU; MI355X_PF_QT forces one, read once per process): every row's arithmetic is the same — the two
    shapes must give the same bits.  Ragged query lengths, context > 0, window + soft cap in the `opts` case."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for qt in ("1", "2"):
        path = str(tmp_path / f"qt{qt}.npy")
        env = dict(os.environ, MI355X_PF_QT=qt)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "prefill_qt_child.py"), path, kind],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
    assert not (outs[0] == 0x7FC0).any(), "every output row must have been written (bf16 NaN fill left)"
