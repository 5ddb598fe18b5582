"""GPU parity: AWQ / GPTQ int4 weight-only path against the CPU oracle.

Integer ops (awq_to_gptq_4bit, gptq_shuffle incl. act-order make_sequential) and the
dequantised weights (awq_dequantize) are bit-exact.  GEMM outputs: the dequantised
weights are identical to the oracle's bit for bit, products are exact in fp32, so the
only freedom is the fp32 accumulation order (MFMA k-order, split-K) before the single
rounding of C to scalar_t: every element within one output ulp (a rounding flip) plus
2e-4 * max|ref| (order noise where the sum cancels) — tighter than north_star's 1e-3 rel —
and at most 5 % of the elements different at all (tests/util.py: assert_gemm_close).

Shapes: scaled-down then full Llama-3-8B AWQ (K,N) from BASELINE.md §3, g = 128, M in
{1, 7, 16, 33, 64, 100}; weights nibbles U{0..15}, zeros U{0..15}, scales U(1e-3, 1e-2).
The reference has no numeric test for these native ops (tests/kernels/quantization/
test_awq.py:11-47 and test_gptq.py:10-32 are opcheck-only): parity is pinned by the
oracle's restatement plus tests/kernels/quantization/test_awq_triton.py's dequant reference.
"""
import numpy as np
import pytest
import torch

from tests.util import assert_bit_exact, assert_gemm_close, dev

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def make_awq(k, n, group, dtype, seed=0):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 16, size=(k, n), dtype=np.uint8)
    z = rng.integers(0, 16, size=(k // group, n), dtype=np.uint8)
    g = torch.Generator().manual_seed(seed)
    scales = (torch.rand(k // group, n, generator=g) * 9e-3 + 1e-3).to(dtype)
    return R.awq_pack(w), R.awq_pack(z), scales, w, z


def make_gptq(k, n, group, dtype, seed=0, sym=False):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 16, size=(k, n), dtype=np.uint8)
    z = np.full((k // group, n), 7, dtype=np.uint8) if sym else \
        rng.integers(0, 15, size=(k // group, n), dtype=np.uint8)
    g = torch.Generator().manual_seed(seed)
    scales = (torch.rand(k // group, n, generator=g) * 9e-3 + 1e-3).to(dtype)
    qz = torch.from_numpy(R._pack_nibbles(z.reshape(k // group, n // 8, 8)).view(np.int32))
    return R.gptq_pack_rows(w), qz, scales


@pytest.mark.parametrize("k,n", [(32, 64), (128, 256), (4096, 4096), (1000 * 8, 72 * 8)])
def test_awq_to_gptq_4bit(k, n):
    qw, _, _, _, _ = make_awq(k, n, k if k < 128 else 8 if k % 128 else 128, torch.float16)
    ref = R.awq_to_gptq_4bit(qw)
    out = ops().awq_to_gptq_4bit(qw.to(dev()))
    assert out.shape == ref.shape == (n, k // 8)       # declared [N, K/8] (reference quirk)
    assert_bit_exact(out, ref, "awq_to_gptq_4bit")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k,n,group", [(128, 64, 32), (512, 256, 128), (4096, 1024, 128)])
def test_awq_dequantize(dtype, k, n, group):
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=1)
    ref = R.awq_dequantize(qw, sc, qz)
    d = dev()
    out = ops().awq_dequantize(qw.to(d), sc.to(d), qz.to(d), 0, 0, 0)
    assert_bit_exact(out, ref, "awq_dequantize")


@pytest.mark.parametrize("act_order", [False, True])
@pytest.mark.parametrize("k,n", [(128, 64), (4096, 512)])
def test_gptq_shuffle(k, n, act_order):
    qw, _, _ = make_gptq(k, n, 128, torch.float16, seed=2)
    perm = torch.randperm(k, generator=torch.Generator().manual_seed(3)).to(torch.int32) if act_order \
        else torch.empty(0, dtype=torch.int32)
    ref = R.gptq_shuffle(qw, perm)
    d = dev()
    qd = qw.to(d).clone()
    ops().gptq_shuffle(qd, perm.to(d) if act_order else perm, 4)
    assert_bit_exact(qd, ref, "gptq_shuffle")


def _check_gemm(out, ref, what):
    assert_gemm_close(out, ref, what)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m", [1, 7, 16, 33, 64, 100, 128, 300])
@pytest.mark.parametrize("k,n,group", [(256, 128, 128), (1024, 768, 128), (512, 256, 32), (2048, 64, 64)])
def test_awq_gemm_small_shapes(dtype, m, k, n, group):
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=4)
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(5)).to(dtype)
    q2 = R.awq_to_gptq_4bit(qw)
    ref = R.awq_gemm(x, q2, sc, qz)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    ws = torch.full((8 * min(m, 64) * n,), float("nan"), dtype=torch.float32, device=d)  # scratch: any contents
    out = ops().awq_gemm(x.to(d), q2d, qz.to(d), sc.to(d), 8, ws, dtype == torch.bfloat16)
    _check_gemm(out, ref, f"awq_gemm m={m}")
    # without a workspace the kernel must not split K and give the same answer
    out2 = ops().awq_gemm(x.to(d), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), dtype == torch.bfloat16)
    _check_gemm(out2, ref, f"awq_gemm (no workspace) m={m}")


@pytest.mark.parametrize("k,n", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
def test_awq_gemm_llama3_8b_decode_shapes(k, n):
    """Full Llama-3-8B AWQ layer shapes at the decode batch M = 64."""
    dtype, m, group = torch.bfloat16, 64, 128
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=6)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(7)) * 0.5).to(dtype)
    ref = R.awq_gemm(x, R.awq_to_gptq_4bit(qw), sc, qz)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    ws = torch.full((8 * min(m, 64) * n,), float("nan"), dtype=torch.float32, device=d)  # scratch: any contents
    out = ops().awq_gemm(x.to(d), q2d, qz.to(d), sc.to(d), 8, ws, True)
    _check_gemm(out, ref, "awq_gemm llama3")


def test_awq_gemm_strided_activations_and_errors():
    dtype, m, k, n, group = torch.bfloat16, 9, 512, 128, 128
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=8)
    xfull = torch.randn(m, 2 * k).to(dtype)
    x = xfull[:, :k]
    ref = R.awq_gemm(x, R.awq_to_gptq_4bit(qw), sc, qz)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    out = ops().awq_gemm(xfull.to(d)[:, :k], q2d, qz.to(d), sc.to(d), 8, torch.empty(0), True)
    _check_gemm(out, ref, "strided")
    with pytest.raises(RuntimeError):   # dtype flag mismatch
        ops().awq_gemm(x.to(d), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), False)
    with pytest.raises(RuntimeError):   # k not a multiple of 32 -> raise, never a silent false
        ops().awq_gemm(x.to(d)[:, :500].contiguous(), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m", [1, 16, 64, 77])
@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("act_order", [False, True])
def test_gptq_gemm(dtype, m, sym, act_order):
    k, n, group = 1024, 256, 128
    qw, qz, sc = make_gptq(k, n, group, dtype, seed=9, sym=sym)
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(10)).to(dtype)
    perm = torch.randperm(k, generator=torch.Generator().manual_seed(11)).to(torch.int32) if act_order \
        else torch.empty(0, dtype=torch.int32)
    shuf = R.gptq_shuffle(qw, perm)
    ref = R.gptq_gemm(x, shuf, qz, sc, perm, group)
    d = dev()
    qd = qw.to(d).clone()
    pd = perm.to(d) if act_order else perm
    ops().gptq_shuffle(qd, pd, 4)
    ws = torch.full((8 * min(m, 64) * n,), float("nan"), dtype=torch.float32, device=d)  # scratch: any contents
    pspace = torch.empty(m, k, dtype=torch.float16, device=d) if act_order else torch.empty(0)
    out = ops().gptq_gemm(x.to(d), qd, qz.to(d), sc.to(d), pd, True, 4, group, pspace, ws,
                          dtype == torch.bfloat16)
    _check_gemm(out, ref, "gptq_gemm")


def test_gptq_gemm_qwen2_72b_tp8_shapes():
    """Qwen2-72B GPTQ per-rank shapes at TP=8, M = 64, symmetric zeros."""
    dtype, m, group = torch.bfloat16, 64, 128
    for k, n in [(8192, 1280), (1024, 8192), (8192, 7424), (3712, 8192)]:
        qw, qz, sc = make_gptq(k, n, group, dtype, seed=12, sym=True)
        x = (torch.randn(m, k, generator=torch.Generator().manual_seed(13)) * 0.5).to(dtype)
        shuf = R.gptq_shuffle(qw, None)
        ref = R.gptq_gemm(x, shuf, qz, sc, None, group)
        d = dev()
        qd = qw.to(d).clone()
        ops().gptq_shuffle(qd, torch.empty(0, dtype=torch.int32), 4)
        ws = torch.full((8 * min(m, 64) * n,), float("nan"), dtype=torch.float32, device=d)  # scratch: any contents
        out = ops().gptq_gemm(x.to(d), qd, qz.to(d), sc.to(d), torch.empty(0, dtype=torch.int32),
                              True, 4, group, torch.empty(0), ws, True)
        _check_gemm(out, ref, f"gptq_gemm {k}x{n}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(128, 4096, 512), (1000, 1024, 1280), (2048, 4096, 6144), (513, 14336, 256)])
def test_awq_gemm_prefill_shapes(dtype, m, k, n):
    """Prefill-sized M (the 128 x 256 MFMA tile kernel), incl. ragged M and N % 256 != 0."""
    group = 128
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=14)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(15)) * 0.5).to(dtype)
    ref = R.awq_gemm(x, R.awq_to_gptq_4bit(qw), sc, qz)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    out = ops().awq_gemm(x.to(d), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), dtype == torch.bfloat16)
    _check_gemm(out, ref, f"awq_gemm prefill {m}x{k}x{n}")


def test_awq_gemm_linearity_full_size():
    """Full prefill chunk (M = 8192) on the Llama-3-8B qkv shape: too big for the CPU oracle,
    so check size-independent properties: (a) rows are independent — the first 64 rows equal
    the small-M kernel's result for those rows up to rounding; (b) zero input rows give
    exactly zero; (c) the large-M and small-M kernels agree on a 2x scaling of the input."""
    dtype, m, k, n, group = torch.bfloat16, 8192, 4096, 6144, 128
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=16)
    d = dev()
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(17)) * 0.5).to(dtype).to(d)
    x[100:164] = 0
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    big = ops().awq_gemm(x, q2d, qz.to(d), sc.to(d), 8, torch.empty(0), True)
    small = ops().awq_gemm(x[:64].contiguous(), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), True)
    _check_gemm(big[:64], small, "rows independent of M")
    assert (big[100:164] == 0).all()
    big2 = ops().awq_gemm((x * 2).contiguous(), q2d, qz.to(d), sc.to(d), 8, torch.empty(0), True)
    assert torch.equal(big2, big * 2)  # power-of-two scaling is exact in every rounding step


@pytest.mark.parametrize("m,k,n", [(64, 4096, 4096), (64, 14336, 4096), (7, 1024, 768), (33, 512, 256)])
def test_deferred_split_k_plus_norm_is_bit_identical(m, k, n):
    """MI355X-side fusion: awq_gemm_deferred + fused_add_rms_norm_slabs must give exactly the bits
    of awq_gemm + fused_add_rms_norm (same slab order, same rounding points), repeatedly."""
    dtype, group = torch.bfloat16, 128
    qw, qz, sc, _, _ = make_awq(k, n, group, dtype, seed=21)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    qz, sc = qz.to(d), sc.to(d)
    g = torch.Generator().manual_seed(22)
    x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(d)
    res0 = torch.randn(m, n, generator=g).to(dtype).to(d)
    w = (torch.rand(n, generator=g) + 0.5).to(dtype).to(d)
    ws = torch.full((8 * m * n,), float("nan"), dtype=torch.float32, device=d)
    ref_out = ops().awq_gemm(x, q2d, qz, sc, 8, ws, True)
    ref_res = res0.clone()
    ops().fused_add_rms_norm(ref_out, ref_res, w, 1e-5)
    saw_slabs = False
    for _ in range(5):
        ws.fill_(float("nan"))
        out, sk = ops().awq_gemm_deferred(x, q2d, qz, sc, ws)
        saw_slabs |= sk >= 2
        res = res0.clone()
        ops().fused_add_rms_norm_slabs(out, res, w, ws, sk, 1e-5)
        assert_bit_exact(out, ref_out, "deferred norm out")
        assert_bit_exact(res, ref_res, "deferred norm residual")
    if (m, k, n) == (64, 4096, 4096):
        assert saw_slabs          # the Llama-3-8B o_proj shape does split K


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(1024, 512, 512), (1300, 256, 1280), (64, 4096, 28672), (7, 256, 384),
                                   (33, 512, 128)])
def test_gate_up_gemm_with_fused_silu_is_bit_identical(dtype, m, k, n):
    """MI355X-side prefill fusion: awq_gemm_silu_mul == silu_and_mul(awq_gemm) bit for bit
    (n = 2 * ffn; ragged M, several 128-column blocks, both dtypes)."""
    qw, qz, sc, _, _ = make_awq(k, n, 128, dtype, seed=31)
    d = dev()
    q2d = ops().awq_to_gptq_4bit(qw.to(d))
    qz, sc = qz.to(d), sc.to(d)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(32)) * 0.5).to(dtype).to(d)
    gu = ops().awq_gemm(x, q2d, qz, sc, 8, torch.empty(0), dtype == torch.bfloat16)
    ref = torch.empty(m, n // 2, dtype=dtype, device=d)
    ops().silu_and_mul(ref, gu)
    out = ops().awq_gemm_silu_mul(x, q2d, qz, sc)
    assert out is not None
    assert_bit_exact(out, ref, "fused silu epilogue")
    assert ops().awq_gemm_silu_mul(x[:1].expand(300, k).contiguous(), q2d, qz, sc) is None   # 64 < M < 1024


def _unpack_operand(p, m, k):
    """Row-major [m, k] view of an operand image (include/mi355x_hotpath.h,
    mi355x_awq_gemm_silu_mul_packed): [row tile][k tile][64 slots of 8 elements]."""
    m_pad = (m + 15) // 16 * 16
    img = p.data.view(m_pad // 16, k // 32, 64, 8).cpu()
    out = torch.empty(m_pad, k, dtype=img.dtype)
    swz = [0, 12, 2, 14]
    for lr in range(4):
        for lc in range(16):
            slot = 16 * lr + (lc ^ swz[lr])
            # rows 16 mt + lc, columns 32 kt + 8 lr .. + 7
            out.view(m_pad // 16, 16, k // 32, 32)[:, lc, :, 8 * lr:8 * lr + 8] = img[:, :, slot, :]
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,ffn,n2", [(1030, 256, 384, 192), (2048, 512, 1024, 512), (1300, 128, 128, 64)])
def test_gate_up_silu_into_packed_operand_then_down_is_bit_identical(dtype, m, k, ffn, n2):
    """MI355X-side prefill fusion pair: awq_gemm_silu_mul_packed writes act as the operand image
    that awq_gemm_packed_a reads; image == silu_and_mul(awq_gemm) re-tiled (rows >= m zero), and
    down(image) == awq_gemm(act) bit for bit (ragged M, several row / column blocks)."""
    n = 2 * ffn
    qw, qz, sc, _, _ = make_awq(k, n, 128, dtype, seed=41)
    qw2, qz2, sc2, _, _ = make_awq(ffn, n2, 128, dtype, seed=42)
    d = dev()
    q2d, qz, sc = ops().awq_to_gptq_4bit(qw.to(d)), qz.to(d), sc.to(d)
    q2d2, qz2, sc2 = ops().awq_to_gptq_4bit(qw2.to(d)), qz2.to(d), sc2.to(d)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(43)) * 0.5).to(dtype).to(d)
    act = ops().awq_gemm_silu_mul(x, q2d, qz, sc)
    ref = ops().awq_gemm(act, q2d2, qz2, sc2, 8, torch.empty(0), dtype == torch.bfloat16)
    packed = ops().awq_gemm_silu_mul_packed(x, q2d, qz, sc)
    assert packed is not None and packed.shape == (m, ffn)
    img = _unpack_operand(packed, m, ffn)
    assert_bit_exact(img[:m], act, "packed act image")
    assert not img[m:].view(torch.int16).any(), "rows >= m of the image must be zero"
    out = ops().awq_gemm_packed_a(packed, q2d2, qz2, sc2)
    assert_bit_exact(out, ref, "down_proj from the packed image")
    assert ops().awq_gemm_silu_mul_packed(x[:512], q2d, qz, sc) is None     # M < 1024


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("quant", ["awq", "gptq"])
def test_prepacked_weight_image_is_bit_identical(dtype, quant):
    """mi355x_w4a16_prepack + mi355x_w4a16_gemm_prepacked (the weights' operand image computed once at
    load time) against the per-call path: plain, SILU epilogue, and SILU -> operand image -> down_proj."""
    d = dev()
    m, k, ffn, n2 = 1100, 256, 512, 192
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(3)) * 0.5).to(dtype).to(d)
    if quant == "awq":
        qw, qz, sc, _, _ = make_awq(k, 2 * ffn, 128, dtype, seed=1)
        qg = ops().awq_to_gptq_4bit(qw.to(d))
        qw2, qz2, sc2, _, _ = make_awq(ffn, n2, 128, dtype, seed=2)
        qg2 = ops().awq_to_gptq_4bit(qw2.to(d))
        gemm = lambda a, w, z, s: ops().awq_gemm(a, w, z, s, 8, torch.empty(0), dtype == torch.bfloat16)   # noqa: E731
    else:
        qw, qz, sc = make_gptq(k, 2 * ffn, 128, dtype, seed=1)
        qg = qw.to(d)
        ops().gptq_shuffle(qg, torch.empty(0, dtype=torch.int32), 4)
        qw2, qz2, sc2 = make_gptq(ffn, n2, 128, dtype, seed=2)
        qg2 = qw2.to(d)
        ops().gptq_shuffle(qg2, torch.empty(0, dtype=torch.int32), 4)
        e = torch.empty(0, dtype=torch.int32, device=d)
        gemm = lambda a, w, z, s: ops().gptq_gemm(a, w, z, s, e, True, 4, 128, torch.empty(0), torch.empty(0),   # noqa: E731
                                                  dtype == torch.bfloat16)
    qz, sc, qz2, sc2 = qz.to(d), sc.to(d), qz2.to(d), sc2.to(d)
    img = ops().w4a16_prepack(qg, qz, sc, quant == "gptq")
    img2 = ops().w4a16_prepack(qg2, qz2, sc2, quant == "gptq")
    assert img.numel() == 2 * ffn * k and img.dtype == dtype
    gu = gemm(x, qg, qz, sc)
    assert_bit_exact(ops().w4a16_gemm_prepacked(x, img, 2 * ffn, k), gu, "plain")
    act = torch.empty(m, ffn, dtype=dtype, device=d)
    ops().silu_and_mul(act, gu)
    assert_bit_exact(ops().w4a16_gemm_prepacked(x, img, 2 * ffn, k, silu=True), act, "silu epilogue")
    down = gemm(act, qg2, qz2, sc2)
    packed = ops().w4a16_gemm_prepacked(x, img, 2 * ffn, k, silu=True, out_image=True)
    assert_bit_exact(ops().w4a16_gemm_prepacked(packed, img2, n2, ffn), down, "silu -> image -> down_proj")
    assert_bit_exact(ops().w4a16_gemm_prepacked(act, img2, n2, ffn), down, "down_proj from row-major act")
    with pytest.raises(RuntimeError):
        ops().w4a16_gemm_prepacked(x[:64], img, 2 * ffn, k)          # decode sizes stream the int4 words


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(576, 4096, 1024), (400, 2048, 320), (1100, 4096, 512), (2048, 8192, 1280)])
def test_prepacked_image_gemm_mid_m_and_split_k(dtype, m, k, n):
    """mi355x_w4a16_gemm_prepacked below 1024 rows (from 384: chunked-prefill budgets) and on shapes with few
    256 x 256 tiles, where it splits K through fp32 slabs (mi355x_w4a16_prepacked_split_elems > 0): the oracle within
    the GEMM tolerance; at m >= 1024 the per-call awq_gemm plans the same split and stays bit-identical."""
    from vllm_metax_amd import _abi
    assert _abi.load().mi355x_w4a16_prepacked_split_elems(m, n, k) >= 2 * m * n
    qw, qz, sc, _, _ = make_awq(k, n, 128, dtype, seed=m)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(k)) * 0.5).to(dtype)
    ref = R.awq_gemm(x, R.awq_to_gptq_4bit(qw), sc, qz)
    d = dev()
    q2d, qzd, scd = ops().awq_to_gptq_4bit(qw.to(d)), qz.to(d), sc.to(d)
    img = ops().w4a16_prepack(q2d, qzd, scd, False)
    out = ops().w4a16_gemm_prepacked(x.to(d), img, n, k)
    _check_gemm(out, ref, f"image gemm {m}x{k}x{n}")
    per_call = ops().awq_gemm(x.to(d), q2d, qzd, scd, 8, torch.empty(0), dtype == torch.bfloat16)
    if m >= 1024:
        assert_bit_exact(out, per_call, "image gemm == per-call gemm at the same m")
    else:
        _check_gemm(per_call, ref, f"stripe passes {m}x{k}x{n}")
    if n % 256 == 0:
        # the split forms of the SILU epilogue (row-major and operand image): the bits of the plain split GEMM
        # followed by silu_and_mul
        act = torch.empty(m, n // 2, dtype=dtype, device=d)
        ops().silu_and_mul(act, out)
        assert_bit_exact(ops().w4a16_gemm_prepacked(x.to(d), img, n, k, silu=True), act, "split K + silu epilogue")
        packed = ops().w4a16_gemm_prepacked(x.to(d), img, n, k, silu=True, out_image=True)
        unp = _unpack_operand(packed, m, n // 2)
        assert_bit_exact(unp[:m], act, "split K + silu epilogue -> operand image")
        assert not unp[m:].view(torch.int16).any(), "rows >= m of the image must be zero"
    with pytest.raises(RuntimeError):
        ops().w4a16_gemm_prepacked(x[:320].to(d), img, n, k)      # below W4_PREPACKED_MIN_M


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,n", [(1, 256, 128), (33, 1024, 320), (64, 4096, 512), (100, 512, 64), (1100, 512, 256)])
@pytest.mark.parametrize("sym", [True, False])
def test_gptq_gemm_8bit(dtype, m, k, n, sym):
    """gptq_gemm bit = 8 (q_gemm.cu:1998-2003): zero = qzeros + 1 (= the fast path's fixed 128 on symmetric
    checkpoints, stored zero 127); gptq_shuffle with bit 8 leaves the words as they are (qdq_8.cuh:14)."""
    rng = np.random.default_rng(7)
    group = 128
    w = rng.integers(0, 256, size=(k, n), dtype=np.uint8)
    z = np.full((k // group, n), 127, dtype=np.uint8) if sym else rng.integers(0, 255, size=(k // group, n), dtype=np.uint8)
    zb = z.reshape(k // group, n // 4, 4).astype(np.uint32)
    qz = torch.from_numpy((zb[..., 0] | (zb[..., 1] << 8) | (zb[..., 2] << 16) | (zb[..., 3] << 24)).view(np.int32))
    qw = R.gptq8_pack_rows(w)
    sc = (torch.rand(k // group, n, generator=torch.Generator().manual_seed(1)) * 4e-4 + 1e-4).to(dtype)
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(2)) * 0.5).to(dtype)
    ref = R.gptq8_gemm(x, qw, qz, sc, group)
    d = dev()
    qd = qw.to(d)
    before = qd.clone()
    ops().gptq_shuffle(qd, torch.empty(0, dtype=torch.int32), 8)
    assert torch.equal(qd, before)
    out = ops().gptq_gemm(x.to(d), qd, qz.to(d), sc.to(d), torch.empty(0, dtype=torch.int32, device=d), True, 8,
                          group, torch.empty(0), torch.empty(0), dtype == torch.bfloat16)
    _check_gemm(out, ref, f"gptq 8-bit {m}x{n}x{k}")
