"""GPU parity: fp8 (e4m3fn) KV cache — SURVEY §8f-3.

Grids after the upstream-style tests the reference holds: tests/kernels/attention/test_cache.py:
167-210 (reshape_and_cache, kv_cache_dtype "fp8": tokens 42, heads 8, head sizes 64/80/256, blocks
8/16/32) and test_attention.py:303-344 (paged attention with an fp8 cache: dequantise, then the torch
reference; atol 1e-2 there).  Expected values: the reference-generated fixture
tests/golden/ref_paged_attention_fp8kv.npz, and the oracle (oracle/ref_ops.py, "fp8 KV cache") for the
cache write / convert (bit-exact) and for k_scale / v_scale != 1.
"""
import numpy as np
import pytest
import torch

from tests import ref_inputs as RI
from tests.ref_inputs import BF, F16, F32, FP8
from tests.test_cpu_ref_fixtures import close_to_f32
from tests.util import assert_bit_exact, dev

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _scales(ks, vs):
    d = dev()
    return torch.tensor([ks], dtype=F32, device=d), torch.tensor([vs], dtype=F32, device=d)


# ------------------------------------------------------------------ cache write / convert
@pytest.mark.parametrize("dtype", [BF, F16, F32])
@pytest.mark.parametrize("head_size", [64, 80, 128, 256])
@pytest.mark.parametrize("block_size", [8, 16, 32])
def test_reshape_and_cache_fp8(dtype, head_size, block_size):
    torch.manual_seed(0)
    T, H, nb = 42, 8, 11
    d = dev()
    key = (torch.randn(T, H, head_size) * 2).to(dtype)
    value = (torch.randn(T, H, head_size) * 2).to(dtype)
    slots = torch.randperm(nb * block_size)[:T].to(torch.int64)
    slots[5] = -1
    ks, vs = 0.37, 2.0
    kc = torch.zeros(nb, H, head_size // 16, block_size, 16, dtype=torch.uint8)
    vc = torch.zeros(nb, H, head_size, block_size, dtype=torch.uint8)
    R.reshape_and_cache_fp8(key, value, kc, vc, slots, ks, vs)
    kd, vd = torch.zeros_like(kc, device=d), torch.zeros_like(vc, device=d)
    k_s, v_s = _scales(ks, vs)
    ops().reshape_and_cache(key.to(d), value.to(d), kd, vd, slots.to(d), "fp8", k_s, v_s)
    assert_bit_exact(kd.cpu(), kc, "key_cache")
    assert_bit_exact(vd.cpu(), vc, "value_cache")
    # strided source rows (a qkv slice) and the generic (unaligned) path
    qkv = torch.randn(T, 3 * H * head_size + 8).to(dtype)
    k2 = qkv[:, 8:8 + H * head_size].view(T, H, head_size)
    v2 = qkv[:, 8 + H * head_size:8 + 2 * H * head_size].view(T, H, head_size)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    R.reshape_and_cache_fp8(k2, v2, kc2, vc2, slots, ks, vs)
    qd = qkv.to(d)
    kd.zero_(), vd.zero_()
    ops().reshape_and_cache(qd[:, 8:8 + H * head_size].view(T, H, head_size),
                            qd[:, 8 + H * head_size:8 + 2 * H * head_size].view(T, H, head_size), kd, vd,
                            slots.to(d), "fp8_e4m3", k_s, v_s)
    assert_bit_exact(kd.cpu(), kc2, "key_cache (strided)")
    assert_bit_exact(vd.cpu(), vc2, "value_cache (strided)")


@pytest.mark.parametrize("layout", ["NHD", "HND"])
def test_reshape_and_cache_flash_fp8(layout):
    torch.manual_seed(1)
    T, H, D, bs, nb = 42, 8, 128, 16, 9
    d = dev()
    key = torch.randn(T + 3, H, D).to(BF)          # longer than slot_mapping (graph padding)
    value = torch.randn(T + 3, H, D).to(BF)
    slots = torch.randperm(nb * bs)[:T].to(torch.int64)
    ks, vs = 0.5, 0.25
    shape = (nb, bs, H, D) if layout == "NHD" else (nb, H, bs, D)
    kc, vc = torch.zeros(shape, dtype=torch.uint8), torch.zeros(shape, dtype=torch.uint8)
    kd, vd = kc.to(d), vc.to(d)
    view = (lambda t: t) if layout == "NHD" else (lambda t: t.permute(0, 2, 1, 3))
    R.reshape_and_cache_flash_fp8(key, value, view(kc), view(vc), slots, ks, vs)
    k_s, v_s = _scales(ks, vs)
    ops().reshape_and_cache_flash(key.to(d), value.to(d), view(kd), view(vd), slots.to(d), "fp8", k_s, v_s)
    assert_bit_exact(kd.cpu(), kc, "key_cache")
    assert_bit_exact(vd.cpu(), vc, "value_cache")


@pytest.mark.parametrize("dtype", [BF, F16, F32])
def test_convert_fp8(dtype):
    torch.manual_seed(2)
    d = dev()
    x = (torch.randn(7, 4, 64, 16) * 3).to(dtype)
    x[0, 0, 0, :4] = torch.tensor([1000.0, -1000.0, 448.0, 1e-4]).to(dtype)
    b = torch.empty(x.shape, dtype=torch.uint8, device=d)
    ops().convert_fp8(b, x.to(d), 0.7, "fp8")
    assert_bit_exact(b.cpu(), R.fp8_quant(x, 0.7), "to fp8")
    codes = torch.arange(256, dtype=torch.uint8)
    codes = codes[(codes & 0x7f) != 0x7f].repeat(4)          # every finite e4m3 code
    back = torch.empty(codes.shape, dtype=dtype, device=d)
    ops().convert_fp8(back, codes.to(d), 1.5, "fp8_e4m3")
    # (+0 and -0 compare equal here: for f16 the multiply is selected as v_fma_mixlo_f16 x, scale, +0,
    #  which turns the product -0 * scale into +0; every other code is bit-exact)
    got, want = back.cpu(), R.fp8_dequant(codes, 1.5, dtype)
    assert_bit_exact(got + 0.0, want + 0.0, "from fp8")
    with pytest.raises(RuntimeError):
        ops().convert_fp8(back, codes.to(d), 1.0, "fp8_e3m4")


# ------------------------------------------------------------------ decode attention
def _decode(m, q, kc8, vc8, bt, sl, slopes, version, ks=1.0, vs=1.0, name="fp8"):
    d = dev()
    S, H, D = q.shape
    max_len = int(sl.max())
    out = torch.full_like(q, float("nan"), device=d)
    k_s, v_s = _scales(ks, vs)
    al = slopes.to(d) if slopes is not None else None
    if version == 1:
        ops().paged_attention_v1(out, q.to(d), kc8.to(d), vc8.to(d), m["KVH"], m["scale"], bt.to(d), sl.to(d),
                                 m["bs"], max_len, al, name, k_s, v_s)
    else:
        P = (max_len + 511) // 512
        tmp = torch.empty(S, H, P, D, dtype=q.dtype, device=d)
        es = torch.empty(S, H, P, dtype=F32, device=d)
        ops().paged_attention_v2(out, es, torch.empty_like(es), tmp, q.to(d), kc8.to(d), vc8.to(d), m["KVH"],
                                 m["scale"], bt.to(d), sl.to(d), m["bs"], max_len, al, name, k_s, v_s)
    torch.cuda.synchronize()
    return out.cpu()


def test_paged_attention_fp8kv_reference_fixture():
    """HIP (fp8 cache, scales 1.0) vs the reference's torch attention on the dequantised cache."""
    z, meta = RI.load("ref_paged_attention_fp8kv")
    for i, m in enumerate(meta):
        q, kc8, vc8, bt, sl, slopes = RI.decode_inputs_fp8(m)
        RI.check_crc(m, q=q, kc=kc8, vc=vc8)
        ref32 = RI.arr(z, i, "out_f32")
        for version in (1, 2):
            o = _decode(m, q, kc8, vc8, bt, sl, slopes, version)
            close_to_f32(o, ref32, f"fp8kv[{i}] v{version}")
            torch.testing.assert_close(o.float(), ref32, atol=1e-2, rtol=1e-5)      # test_attention.py:340-343


@pytest.mark.parametrize("ks,vs", [(0.5, 2.0), (0.3, 1.7)])
@pytest.mark.parametrize("dtype", [BF, F16, F32])
def test_paged_attention_fp8kv_scales(ks, vs, dtype):
    """k_scale / v_scale != 1 against the oracle (upstream semantics: T(byte * scale) per element).
    Powers of two are exact in both forms; for 0.3 / 1.7 the kernel folds the scale into the logits /
    the output (one rounding less), which stays inside 1e-3 * max|ref| + one output rounding."""
    m = dict(H=32, KVH=8, d=128, bs=16, nb=64, alibi=False, seed=4242, scale=128 ** -0.5,
             dtype={BF: "bfloat16", F16: "float16", F32: "float32"}[dtype], seq_lens=[1, 17, 300, 777])
    q, kc8, vc8, bt, sl, _ = RI.decode_inputs_fp8(m)
    kc, vc = R.fp8_dequant(kc8, ks, dtype), R.fp8_dequant(vc8, vs, dtype)
    ref1 = R.paged_attention_v1(q, kc, vc, 8, m["scale"], bt, sl)
    for version in (1, 2):
        o = _decode(m, q, kc8, vc8, bt, sl, None, version, ks, vs)
        close_to_f32(o, ref1.float(), f"scales v{version}", rel=1e-3 + 2.0 ** -8)


def test_paged_attention_fp8kv_full_size_halves_the_bytes():
    """Bench shape (64 seqs x 32/8 heads x 128, ctx 1088) by properties: every sequence's output equals
    the output of the same sequence run alone (batch independence) and v1 == v2 within rounding."""
    torch.manual_seed(3)
    S, H, KVH, D, bs = 64, 32, 8, 128, 16
    L = 1088
    nb = S * (L // bs)
    d = dev()
    kc8 = torch.randint(0, 256, (nb, KVH, D // 16, bs, 16), dtype=torch.uint8)
    vc8 = torch.randint(0, 256, (nb, KVH, D, bs), dtype=torch.uint8)
    kc8[(kc8 & 0x7f) == 0x7f] = 0x38          # no NaN codes
    vc8[(vc8 & 0x7f) == 0x7f] = 0x38
    kc8 = (kc8 & 0xbf) | 0x00                 # |value| < 2: bit 6 cleared keeps exponents small
    vc8 = vc8 & 0xbf
    q = torch.randn(S, H, D).to(BF)
    bt = torch.randperm(nb).to(torch.int32).reshape(S, L // bs)
    sl = torch.full((S,), L, dtype=torch.int32)
    m = dict(KVH=KVH, scale=D ** -0.5, bs=bs)
    o1 = _decode(m, q, kc8, vc8, bt, sl, None, 1)
    o2 = _decode(m, q, kc8, vc8, bt, sl, None, 2)
    assert torch.isfinite(o1.float()).all()
    assert (o1.float() - o2.float()).abs().max() <= 2.0 ** -6 * o1.float().abs().max()
    for s in (0, 37, 63):
        alone = _decode(m, q[s:s + 1], kc8, vc8, bt[s:s + 1], sl[s:s + 1], None, 1)
        assert_bit_exact(alone[0], o1[s], f"sequence {s} alone")
    ref = R.paged_attention_v1(q[5:6], R.fp8_dequant(kc8, 1.0, BF), R.fp8_dequant(vc8, 1.0, BF), KVH, D ** -0.5,
                               bt[5:6], sl[5:6])
    close_to_f32(o1[5:6], ref.float(), "oracle spot check", rel=1e-3 + 2.0 ** -8)


def test_fp8kv_rejections():
    d = dev()
    q = torch.zeros(1, 8, 128, dtype=BF, device=d)
    out = torch.empty_like(q)
    kc = torch.zeros(4, 8, 8, 8, 16, dtype=torch.uint8, device=d)       # block size 8
    vc = torch.zeros(4, 8, 128, 8, dtype=torch.uint8, device=d)
    bt = torch.zeros(1, 1, dtype=torch.int32, device=d)
    sl = torch.ones(1, dtype=torch.int32, device=d)
    k_s, v_s = _scales(1.0, 1.0)
    with pytest.raises(RuntimeError, match="block size"):
        ops().paged_attention_v1(out, q, kc, vc, 8, 1.0, bt, sl, 8, 1, None, "fp8", k_s, v_s)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops().paged_attention_v1(out, q, kc, vc, 8, 1.0, bt, sl, 8, 1, None, "fp8_e3m4", k_s, v_s)
    with pytest.raises(RuntimeError, match="1-byte cache"):
        ops().paged_attention_v1(out, q, kc.to(BF), vc.to(BF), 8, 1.0, bt, sl, 8, 1, None, "fp8", k_s, v_s)


# ------------------------------------------------------------------ prefill attention
@pytest.mark.parametrize("spec", [([130, 7, 64], [130, 300, 64]), ([32, 1, 200], [1000, 77, 200]), ([1], [1])])
@pytest.mark.parametrize("ks,vs", [(1.0, 1.0), (0.5, 1.7)])
def test_paged_prefill_fp8kv(spec, ks, vs):
    """Fast path (d 128, bs 16, bf16) and generic path (d 64 / fp32) against the oracle's prefill on the
    dequantised cache; tolerance as tests/test_gpu_prefill_attention.py."""
    q_lens, seq_lens = spec
    for (H, KVH, D, bs, dtype) in [(8, 2, 128, 16, BF), (4, 4, 64, 16, F16), (4, 2, 128, 32, F32)]:
        torch.manual_seed(5)
        S = len(q_lens)
        mb = (max(seq_lens) + bs - 1) // bs
        nb = S * mb + 3
        kc8 = (torch.rand(nb, KVH, D // 16, bs, 16) * 2 - 1).to(FP8).view(torch.uint8)
        vc8 = (torch.rand(nb, KVH, D, bs) * 2 - 1).to(FP8).view(torch.uint8)
        bt = torch.randperm(nb)[:S * mb].to(torch.int32).reshape(S, mb)
        cu = torch.zeros(S + 1, dtype=torch.int32)
        cu[1:] = torch.tensor(q_lens).cumsum(0)
        q = (torch.randn(int(cu[-1]), H, D) * 0.7).to(dtype)
        sl = torch.tensor(seq_lens, dtype=torch.int32)
        scale = D ** -0.5
        ref = R.paged_prefill_attention(q, R.fp8_dequant(kc8, ks, dtype), R.fp8_dequant(vc8, vs, dtype), KVH, scale,
                                        bt, sl, cu)
        # poison the tail slots of the last block of every sequence: must not reach the output
        for s, L in enumerate(seq_lens):
            if L % bs:
                blk = int(bt[s, (L - 1) // bs])
                kc8[blk, :, :, L % bs:, :] = 0x7f
                vc8[blk, :, :, L % bs:] = 0x7f
        d = dev()
        out = torch.full(q.shape, float("nan"), dtype=dtype, device=d)
        k_s, v_s = _scales(ks, vs)
        ops().paged_prefill_attention(out, q.to(d), kc8.to(d), vc8.to(d), KVH, scale, bt.to(d), sl.to(d), cu.to(d),
                                      max(q_lens), bs, "fp8", k_s, v_s)
        torch.cuda.synchronize()
        close_to_f32(out.cpu(), ref.float(), f"prefill fp8 {H}/{KVH} d{D} {dtype}", rel=2e-3 + 2.0 ** -8)


# ------------------------------------------------------------------ e5m2 cache ("fp8_e5m2", round 3)
BF8 = torch.float8_e5m2


@pytest.mark.parametrize("dtype", [BF, F16, F32])
@pytest.mark.parametrize("head_size,block_size", [(64, 16), (128, 16), (128, 32), (80, 8)])
def test_reshape_and_cache_and_convert_e5m2(dtype, head_size, block_size):
    """Cache write and convert_fp8 with kv_cache_dtype "fp8_e5m2" against the oracle (sat_e5m2(float(x) / scale),
    RNE): bit-exact on the tiled and the generic path; the saturation bound 57344 is exercised."""
    torch.manual_seed(1)
    T, H, nb = 42, 8, 11
    d = dev()
    key = (torch.randn(T, H, head_size) * 2).to(dtype)
    value = (torch.randn(T, H, head_size) * 2).to(dtype)
    if dtype != F16:
        key[0, 0, :2] = torch.tensor([1.0e5, -1.0e5]).to(dtype)      # saturates at +-57344 / scale
    slots = torch.randperm(nb * block_size)[:T].to(torch.int64)
    slots[5] = -1
    ks, vs = 0.37, 2.0
    kc = torch.zeros(nb, H, head_size // 16, block_size, 16, dtype=torch.uint8)
    vc = torch.zeros(nb, H, head_size, block_size, dtype=torch.uint8)
    R.reshape_and_cache_fp8(key, value, kc, vc, slots, ks, vs, "e5m2")
    kd, vd = torch.zeros_like(kc, device=d), torch.zeros_like(vc, device=d)
    k_s, v_s = _scales(ks, vs)
    ops().reshape_and_cache(key.to(d), value.to(d), kd, vd, slots.to(d), "fp8_e5m2", k_s, v_s)
    assert_bit_exact(kd.cpu(), kc, "key_cache")
    assert_bit_exact(vd.cpu(), vc, "value_cache")
    x = (torch.randn(5, 64) * 3).to(dtype)
    b = torch.empty(x.shape, dtype=torch.uint8, device=d)
    ops().convert_fp8(b, x.to(d), 0.7, "fp8_e5m2")
    assert_bit_exact(b.cpu(), R.fp8_quant(x, 0.7, "e5m2"), "to e5m2")
    codes = torch.arange(256, dtype=torch.uint8)
    codes = codes[(codes & 0x7c) != 0x7c].repeat(4)           # every finite e5m2 code
    back = torch.empty(codes.shape, dtype=dtype, device=d)
    ops().convert_fp8(back, codes.to(d), 1.5, "fp8_e5m2")
    assert_bit_exact(back.cpu() + 0.0, R.fp8_dequant(codes, 1.5, dtype, "e5m2") + 0.0, "from e5m2")


@pytest.mark.parametrize("dtype", [BF, F16, F32])
@pytest.mark.parametrize("heads", [(32, 8), (8, 1)])
@pytest.mark.parametrize("ks,vs", [(1.0, 1.0), (0.5, 2.0)])
def test_paged_attention_e5m2kv(dtype, heads, ks, vs):
    """Decode attention over an e5m2 cache (v1 and v2) against the oracle on the dequantised cache."""
    H, KVH = heads
    torch.manual_seed(7)
    D, bs, seq_lens = 128, 16, [1, 17, 300, 777]
    S = len(seq_lens)
    mb = (max(seq_lens) + bs - 1) // bs
    nb = S * mb + 2
    kc8 = (torch.rand(nb, KVH, D // 16, bs, 16) * 2 - 1).to(BF8).view(torch.uint8)
    vc8 = (torch.rand(nb, KVH, D, bs) * 2 - 1).to(BF8).view(torch.uint8)
    bt = torch.randperm(nb)[:S * mb].to(torch.int32).reshape(S, mb)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    q = (torch.randn(S, H, D) * 0.5).to(dtype)
    m = dict(KVH=KVH, scale=D ** -0.5, bs=bs)
    kc, vc = R.fp8_dequant(kc8, ks, dtype, "e5m2"), R.fp8_dequant(vc8, vs, dtype, "e5m2")
    ref = R.paged_attention_v1(q, kc, vc, KVH, m["scale"], bt, sl)
    for version in (1, 2):
        o = _decode(m, q, kc8, vc8, bt, sl, None, version, ks, vs, name="fp8_e5m2")
        close_to_f32(o, ref.float(), f"e5m2 v{version}", rel=1e-3 + 2.0 ** -8)


def test_paged_prefill_e5m2kv():
    """Prefill attention over an e5m2 cache: MFMA path (d 128, bs 16, 16-bit; plain and with a sliding window) and
    the general kernel, against the oracle on the dequantised cache."""
    q_lens, seq_lens = [130, 7, 64], [130, 300, 64]
    for (H, KVH, D, bs, dtype, window) in [(8, 2, 128, 16, BF, None), (8, 2, 128, 16, F16, 96), (4, 4, 64, 16, F16, None),
                                           (4, 2, 128, 32, F32, None)]:
        torch.manual_seed(5)
        S = len(q_lens)
        mb = (max(seq_lens) + bs - 1) // bs
        nb = S * mb + 3
        kc8 = (torch.rand(nb, KVH, D // 16, bs, 16) * 2 - 1).to(BF8).view(torch.uint8)
        vc8 = (torch.rand(nb, KVH, D, bs) * 2 - 1).to(BF8).view(torch.uint8)
        bt = torch.randperm(nb)[:S * mb].to(torch.int32).reshape(S, mb)
        cu = torch.zeros(S + 1, dtype=torch.int32)
        cu[1:] = torch.tensor(q_lens).cumsum(0)
        q = (torch.randn(int(cu[-1]), H, D) * 0.7).to(dtype)
        sl = torch.tensor(seq_lens, dtype=torch.int32)
        scale = D ** -0.5
        ref = R.paged_prefill_attention(q, R.fp8_dequant(kc8, 0.5, dtype, "e5m2"), R.fp8_dequant(vc8, 1.7, dtype, "e5m2"),
                                        KVH, scale, bt, sl, cu, sliding_window=window)
        d = dev()
        out = torch.full(q.shape, float("nan"), dtype=dtype, device=d)
        k_s, v_s = _scales(0.5, 1.7)
        ops().paged_prefill_attention(out, q.to(d), kc8.to(d), vc8.to(d), KVH, scale, bt.to(d), sl.to(d), cu.to(d),
                                      max(q_lens), bs, "fp8_e5m2", k_s, v_s, window)
        torch.cuda.synchronize()
        close_to_f32(out.cpu(), ref.float(), f"prefill e5m2 {H}/{KVH} d{D} {dtype}", rel=2e-3 + 2.0 ** -8)
