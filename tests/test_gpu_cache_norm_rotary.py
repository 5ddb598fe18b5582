"""GPU parity: cache ops, layernorm family, fp8 quant, rotary, silu_and_mul.

HIP path (through the C-ABI, vllm_metax_amd._custom_ops) vs the CPU oracle
(oracle/ref_ops.py) on the same seeded inputs.  Parameter grids follow the reference's
tests: tests/kernels/attention/test_cache.py:13-40, tests/kernels/core/test_layernorm.py
:12-17, test_fused_quant_layernorm.py:13-24, test_pos_encoding.py:14-25.

Tolerances (stated per SURVEY §8 / north_star):
  * cache / copy ops: bit-exact.
  * 16-bit normalised outputs: the only freedom is the fp32 summation order of the
    variance (and v_rsq_f32's last bit), which can flip the rounding of T(x*inv_rms) by
    one ulp; the following T*T product can turn that into 2 ulp of the output:
    <= 2 ulp, on <= 0.5 % of the elements.
  * fp8 outputs: <= 1 fp8 ulp on <= 0.5 % of elements; per-token scales rel 1e-6.
  * rotary / silu: bit-exact for 16-bit types (pure per-element arithmetic with the
    same rounding points); fp32 within 1e-6 relative (fma contraction).
"""
import numpy as np
import pytest
import torch

from tests.util import (assert_bit_exact, assert_close_rel, assert_mostly_exact, dev)

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


DTYPES = [torch.float16, torch.bfloat16, torch.float32]


# ------------------------------------------------------------------------------ cache
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("head_size", [64, 80, 128, 256])
@pytest.mark.parametrize("block_size", [8, 16, 32])
def test_reshape_and_cache(dtype, head_size, block_size):
    torch.manual_seed(0)
    T, H, nb = 42, 8, 64
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    slots = torch.randperm(nb * block_size)[:T].to(torch.int64)
    slots[5] = -1  # padding token must be skipped
    qkv = torch.randn(T, 3, H, head_size).to(dtype)
    key, value = qkv[:, 1], qkv[:, 2]  # strided rows like a split qkv
    kc = torch.randn(nb, H, head_size // x, block_size, x).to(dtype)
    vc = torch.randn(nb, H, head_size, block_size).to(dtype)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    R.reshape_and_cache(key, value, kc_ref, vc_ref, slots)
    d = dev()
    qkv_d = qkv.to(d)
    kc_d, vc_d = kc.to(d), vc.to(d)
    ops().reshape_and_cache(qkv_d[:, 1], qkv_d[:, 2], kc_d, vc_d, slots.to(d), "auto")
    torch.cuda.synchronize()
    assert_bit_exact(kc_d, kc_ref, "key_cache")
    assert_bit_exact(vc_d, vc_ref, "value_cache")


def test_reshape_and_cache_consecutive_prefill_slots():
    """prefill-shaped case: 300 tokens with consecutive slots (the tiled path's fast case)."""
    torch.manual_seed(1)
    dtype, T, H, D, bs, nb = torch.bfloat16, 300, 8, 128, 16, 40
    slots = (torch.arange(T) + 37).to(torch.int64)
    key, value = torch.randn(T, H, D).to(dtype), torch.randn(T, H, D).to(dtype)
    kc = torch.zeros(nb, H, D // 8, bs, 8, dtype=dtype)
    vc = torch.zeros(nb, H, D, bs, dtype=dtype)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    R.reshape_and_cache(key, value, kc_ref, vc_ref, slots)
    d = dev()
    kc_d, vc_d = kc.to(d), vc.to(d)
    ops().reshape_and_cache(key.to(d), value.to(d), kc_d, vc_d, slots.to(d), "auto")
    assert_bit_exact(kc_d, kc_ref, "key_cache")
    assert_bit_exact(vc_d, vc_ref, "value_cache")


def test_reshape_and_cache_empty_and_bad_dtype():
    d = dev()
    key = torch.zeros(0, 8, 128, dtype=torch.bfloat16, device=d)
    kc = torch.zeros(4, 8, 16, 16, 8, dtype=torch.bfloat16, device=d)
    vc = torch.zeros(4, 8, 128, 16, dtype=torch.bfloat16, device=d)
    ops().reshape_and_cache(key, key, kc, vc, torch.zeros(0, dtype=torch.int64, device=d), "auto")
    with pytest.raises(RuntimeError):
        ops().reshape_and_cache(key, key, kc, vc, torch.zeros(0, dtype=torch.int64, device=d), "fp8")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("head_size", [64, 80, 128])
@pytest.mark.parametrize("layout", ["NHD", "HND"])
def test_reshape_and_cache_flash(dtype, head_size, layout):
    torch.manual_seed(0)
    T, H, nb, bs = 42, 8, 32, 16
    slots = torch.randperm(nb * bs)[:T].to(torch.int64)
    slots[0] = -1
    key = torch.randn(T + 6, H, head_size).to(dtype)  # longer than slot_mapping (graph padding)
    value = torch.randn(T + 6, H, head_size).to(dtype)
    if layout == "NHD":
        cache = torch.randn(2, nb, bs, H, head_size).to(dtype)
        view = lambda c: c
    else:
        cache = torch.randn(2, nb, H, bs, head_size).to(dtype)
        view = lambda c: c.permute(0, 1, 3, 2, 4)
    ref = cache.clone()
    R.reshape_and_cache_flash(key, value, view(ref)[0], view(ref)[1], slots)
    d = dev()
    cd = cache.to(d)
    ops().reshape_and_cache_flash(key.to(d), value.to(d), view(cd)[0], view(cd)[1], slots.to(d), "auto")
    assert_bit_exact(cd, ref, "flash cache")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_copy_blocks(dtype):
    torch.manual_seed(0)
    L, nb, H, D, bs = 5, 32, 4, 64, 16
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    kcs = [torch.randn(nb, H, D // x, bs, x).to(dtype) for _ in range(L)]
    vcs = [torch.randn(nb, H, D, bs).to(dtype) for _ in range(L)]
    src = torch.randperm(nb)[:6]
    rest = [b for b in range(nb) if b not in src.tolist()]
    mapping = []
    for i, s in enumerate(src.tolist()):
        mapping.append((s, rest[2 * i]))
        mapping.append((s, rest[2 * i + 1]))
    bm = torch.tensor(mapping, dtype=torch.int64)
    kr, vr = [k.clone() for k in kcs], [v.clone() for v in vcs]
    R.copy_blocks(kr, vr, bm)
    d = dev()
    kd, vd = [k.to(d) for k in kcs], [v.to(d) for v in vcs]
    ops().copy_blocks(kd, vd, bm.to(d))
    for a, b in zip(kd + vd, kr + vr):
        assert_bit_exact(a, b, "copy_blocks")


@pytest.mark.parametrize("direction", ["d2d", "d2h", "h2d"])
def test_swap_blocks(direction):
    torch.manual_seed(0)
    nb = 24
    src = torch.randn(nb, 4, 8, 16, 8).to(torch.bfloat16)
    dst = torch.randn(nb, 4, 8, 16, 8).to(torch.bfloat16)
    bm = torch.tensor([(1, 5), (2, 6), (3, 7), (10, 0), (20, 21)], dtype=torch.int64)
    ref = dst.clone()
    R.swap_blocks(src, ref, bm)
    d = dev()
    s = src.to(d) if direction in ("d2d", "d2h") else src.pin_memory()
    t = dst.to(d) if direction in ("d2d", "h2d") else dst.pin_memory()
    ops().swap_blocks(s, t, bm)
    torch.cuda.synchronize()
    assert_bit_exact(t, ref, "swap_blocks")
    with pytest.raises(RuntimeError):
        ops().swap_blocks(s, t, bm.to(d))  # block_mapping must be on CPU


# -------------------------------------------------------------------------- layernorm
NORM_TOKENS = [7, 83, 512]
NORM_HIDDEN = [8, 768, 769, 4096, 5120, 8192, 8199]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hidden", NORM_HIDDEN)
@pytest.mark.parametrize("tokens", NORM_TOKENS)
@pytest.mark.parametrize("add_residual,strided", [(False, False), (True, False), (True, True), (False, True)])
def test_rms_norm(dtype, hidden, tokens, add_residual, strided):
    torch.manual_seed(0)
    eps = 1e-6
    w = (torch.randn(hidden) * 0.1 + 1.0).to(dtype)
    last = hidden * 2 if strided else hidden
    xfull = (torch.randn(tokens, last) * (1.0 / (2 * hidden))).to(dtype)
    x = xfull[..., :hidden]
    res = torch.randn(tokens, hidden).to(dtype) * 0 + (torch.randn(tokens, hidden) * (1.0 / (2 * hidden))).to(dtype)
    d = dev()
    xd = xfull.to(d)[..., :hidden]
    frac, ulp = (5e-3, 2) if dtype != torch.float32 else (1.0, 64)
    if add_residual:
        ref_out, ref_res = R.fused_add_rms_norm(x, res, w, eps)
        rd = res.to(d)
        ops().fused_add_rms_norm(xd, rd, w.to(d), eps)
        assert_bit_exact(rd, ref_res, "residual") if dtype != torch.float32 else assert_close_rel(rd, ref_res, 1e-6)
        assert_mostly_exact(xd.contiguous(), ref_out, ulp, frac, "fused_add_rms_norm out")
    else:
        ref_out = R.rms_norm(x, w, eps)
        out = torch.empty(tokens, hidden, dtype=dtype, device=d)
        ops().rms_norm(out, xd, w.to(d), eps)
        assert_mostly_exact(out, ref_out, ulp, frac, "rms_norm out")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hidden", [64, 769, 4096, 8192])
@pytest.mark.parametrize("tokens", [1, 7, 83])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm_static_fp8_quant(dtype, hidden, tokens, add_residual):
    torch.manual_seed(0)
    eps = 1e-6
    w = (torch.randn(hidden) * 0.1 + 1.0).to(dtype)
    x = torch.randn(tokens, hidden).to(dtype)
    res = torch.randn(tokens, hidden).to(dtype)
    scale = torch.tensor([0.02], dtype=torch.float32)
    d = dev()
    out = torch.empty(tokens, hidden, dtype=torch.float8_e4m3fn, device=d)
    if add_residual:
        ref_q, ref_res = R.fused_add_rms_norm_static_fp8_quant(x, res, w, scale, eps)
        xd, rd = x.to(d), res.to(d)
        ops().fused_add_rms_norm_static_fp8_quant(out, xd, rd, w.to(d), scale.to(d), eps)
        assert_bit_exact(rd, ref_res, "residual")
        assert_bit_exact(xd, x, "input must stay untouched")
    else:
        ref_q = R.rms_norm_static_fp8_quant(x, w, scale, eps)
        ops().rms_norm_static_fp8_quant(out, x.to(d), w.to(d), scale.to(d), eps)
    assert_mostly_exact(out, ref_q, 1, 5e-3, "fp8 out")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("hidden", [64, 1026, 5120, 8192])   # 1026: not a multiple of 4/8
@pytest.mark.parametrize("tokens", [1, 7, 83])
@pytest.mark.parametrize("add_residual", [False, True])
@pytest.mark.parametrize("use_ub", [False, True])
def test_rms_norm_dynamic_per_token_quant(dtype, hidden, tokens, add_residual, use_ub):
    torch.manual_seed(0)
    eps = 1e-6
    w = (torch.randn(hidden) * 0.2 + 1.0).to(dtype)
    x = (torch.randn(tokens, hidden) * 3).to(dtype)
    res = torch.randn(tokens, hidden).to(dtype) if add_residual else None
    ub = torch.tensor([2.5], dtype=torch.float32) if use_ub else None
    ref_q, ref_s, ref_res = R.rms_norm_dynamic_per_token_quant(x, w, eps, ub, res)
    d = dev()
    out = torch.empty(tokens, hidden, dtype=torch.float8_e4m3fn, device=d)
    scales = torch.empty(tokens, 1, dtype=torch.float32, device=d)
    rd = res.to(d) if add_residual else None
    ops().rms_norm_dynamic_per_token_quant(out, x.to(d), w.to(d), scales, eps,
                                           ub.to(d) if use_ub else None, rd)
    assert_close_rel(scales, ref_s, 1e-6, "scales")
    if add_residual:
        assert_bit_exact(rd, ref_res, "residual")
    assert_mostly_exact(out, ref_q, 1, 5e-3, "fp8 out")


# -------------------------------------------------------------------------- fp8 quant
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hidden", [17, 1024, 5137, 8192])
@pytest.mark.parametrize("tokens", [1, 7, 4096 // 64])
def test_fp8_quant(dtype, hidden, tokens):
    torch.manual_seed(0)
    x = (torch.rand(tokens, hidden) * 200 - 100).to(dtype)
    x[0, 0] = 1e4 if dtype != torch.float16 else 6e4  # saturation path
    d = dev()
    xd = x.to(d)
    # static
    scale = torch.tensor([0.3], dtype=torch.float32)
    out = torch.empty(tokens, hidden, dtype=torch.float8_e4m3fn, device=d)
    ops().static_scaled_fp8_quant(out, xd, scale.to(d))
    assert_bit_exact(out, R.static_scaled_fp8_quant(x, scale), "static")
    # dynamic per tensor (scale zero-initialised by the caller)
    sd = torch.zeros(1, dtype=torch.float32, device=d)
    ops().dynamic_scaled_fp8_quant(out, xd, sd)
    ref_q, ref_s = R.dynamic_scaled_fp8_quant(x)
    assert_bit_exact(sd, ref_s, "dynamic scale")
    assert_bit_exact(out, ref_q, "dynamic")
    # dynamic per token, with and without an upper bound
    for ub in (None, torch.tensor([60.0], dtype=torch.float32)):
        st = torch.empty(tokens, 1, dtype=torch.float32, device=d)
        ops().dynamic_per_token_scaled_fp8_quant(out, xd, st, ub.to(d) if ub is not None else None)
        ref_q, ref_s = R.dynamic_per_token_scaled_fp8_quant(x, ub)
        assert_bit_exact(st, ref_s, "per-token scales")
        assert_mostly_exact(out, ref_q, 1, 1e-3, "per-token")  # x/s: division rounding only


def test_fp8_quant_strided_rows():
    torch.manual_seed(0)
    x = torch.randn(9, 2 * 512).to(torch.bfloat16)
    d = dev()
    xd = x.to(d)[:, :512]
    out = torch.empty(9, 512, dtype=torch.float8_e4m3fn, device=d)
    scale = torch.tensor([0.01], dtype=torch.float32)
    ops().static_scaled_fp8_quant(out, xd, scale.to(d))
    assert_bit_exact(out, R.static_scaled_fp8_quant(x[:, :512], scale), "strided static")


# ----------------------------------------------------------------------------- rotary
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("head_size,rot_dim", [(64, 64), (80, 80), (128, 128), (128, 64), (256, 32), (120, 120)])
@pytest.mark.parametrize("use_key", [True, False])
def test_rotary_embedding(dtype, is_neox, head_size, rot_dim, use_key):
    torch.manual_seed(0)
    T, H, KVH, max_pos = 11 * 3, 8, 2, 8192
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    t = torch.arange(max_pos).float()
    freqs = torch.outer(t, inv_freq)
    cache = torch.cat([freqs.cos(), freqs.sin()], dim=-1).to(dtype)
    positions = torch.randint(0, max_pos, (T,), dtype=torch.int64)
    qkv = torch.randn(T, (H + 2 * KVH) * head_size).to(dtype)
    q = qkv[:, :H * head_size]                                   # strided views of a fused qkv
    k = qkv[:, H * head_size:(H + KVH) * head_size] if use_key else None
    ref_q, ref_k = R.rotary_embedding(positions, q, k, head_size, cache, is_neox)
    d = dev()
    qkv_d = qkv.to(d)
    qd = qkv_d[:, :H * head_size]
    kd = qkv_d[:, H * head_size:(H + KVH) * head_size] if use_key else None
    ops().rotary_embedding(positions.to(d), qd, kd, head_size, cache.to(d), is_neox)
    if dtype == torch.float32:
        assert_close_rel(qd, ref_q, 1e-6, "q")
        if use_key:
            assert_close_rel(kd, ref_k, 1e-6, "k")
    else:
        assert_bit_exact(qd.contiguous(), ref_q.contiguous(), "q")
        if use_key:
            assert_bit_exact(kd.contiguous(), ref_k.contiguous(), "k")
    # the value slice of the fused tensor must be untouched
    assert_bit_exact(qkv_d[:, (H + KVH) * head_size:].contiguous(),
                     qkv[:, (H + KVH) * head_size:].contiguous(), "v untouched")


def test_rotary_embedding_3d_and_batched_positions():
    torch.manual_seed(0)
    B, L, H, hs = 2, 5, 4, 128
    cache = torch.randn(64, hs).to(torch.bfloat16)
    positions = torch.randint(0, 64, (B, L), dtype=torch.int64)
    q = torch.randn(B, L, H, hs).to(torch.bfloat16)
    k = torch.randn(B, L, H, hs).to(torch.bfloat16)
    ref_q, ref_k = R.rotary_embedding(positions, q, k, hs, cache, True)
    d = dev()
    qd, kd = q.to(d), k.to(d)
    ops().rotary_embedding(positions.to(d), qd, kd, hs, cache.to(d), True)
    assert_bit_exact(qd, ref_q, "q")
    assert_bit_exact(kd, ref_k, "k")


@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_batched_rotary_embedding(is_neox, dtype):
    """batched_rotary_embedding (multi-LoRA rope tables stacked in one cache; csrc/pos_encoding_kernels.cu:
    102-129, :219-306): cache row = position + per-token offset; via the ctypes op and torch.ops._C, both
    bit-exact against the oracle.  Grid as tests/kernels/core/test_pos_encoding.py:126-192 (scaled down)."""
    import vllm_metax_amd._C  # noqa: F401
    torch.manual_seed(1)
    T, H, KVH, hs, rot, max_pos, tables = 37, 8, 2, 128, 64, 512, 3
    cache = torch.randn(max_pos * tables, rot).to(dtype)
    positions = torch.randint(0, max_pos, (T,), dtype=torch.int64)
    offsets = torch.randint(0, tables, (T,), dtype=torch.int64) * max_pos
    q = torch.randn(T, H * hs).to(dtype)
    k = torch.randn(T, KVH * hs).to(dtype)
    ref_q, ref_k = R.batched_rotary_embedding(positions, q, k, hs, cache, is_neox, offsets)
    d = dev()
    for call in ("ctypes", "torch.ops"):
        qd, kd = q.to(d), k.to(d)
        if call == "ctypes":
            ops().batched_rotary_embedding(positions.to(d), qd, kd, hs, cache.to(d), is_neox, rot, offsets.to(d))
        else:
            torch.ops._C.batched_rotary_embedding(positions.to(d), qd, kd, hs, cache.to(d), is_neox, rot,
                                                  offsets.to(d))
        assert_bit_exact(qd, ref_q, f"q ({call})")
        assert_bit_exact(kd, ref_k, f"k ({call})")
    with pytest.raises(RuntimeError):
        ops().batched_rotary_embedding(positions.to(d), q.to(d), None, hs, cache.to(d), is_neox, rot,
                                       offsets[:5].to(d))


# ------------------------------------------------------------------------- activation
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d_", [13, 512, 2048, 14336])
@pytest.mark.parametrize("tokens", [1, 7, 83])
def test_silu_and_mul(dtype, d_, tokens):
    torch.manual_seed(0)
    x = torch.randn(tokens, 2 * d_).to(dtype)
    ref = R.silu_and_mul(x)
    d = dev()
    out = torch.empty(tokens, d_, dtype=dtype, device=d)
    ops().silu_and_mul(out, x.to(d))
    if dtype == torch.float32:
        assert_close_rel(out, ref, 1e-6, "silu_and_mul")
    else:
        # expf vs torch.exp can differ by an fp32 ulp -> rare 1-ulp flips of T(silu(x)),
        # which the following T*T product can turn into 2 ulp of the output
        assert_mostly_exact(out, ref, 2, 2e-3, "silu_and_mul")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sk", [0, 3])
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (8, 2, 64), (6, 3, 128), (5, 1, 256)])   # last two: head groups straddle q|k|v
def test_qkv_rope_cache_matches_the_three_ops(dtype, sk, H, KVH, D):
    """MI355X-side decode fusion: [slab sum +] rotary + reshape_and_cache in one launch must give
    exactly the bits of the reference-op sequence (oracle: R.rotary_embedding / R.reshape_and_cache)."""
    from tests.util import make_kv_cache_x
    torch.manual_seed(3)
    T, BS, NB, max_pos = 19, 16, 40, 4096
    width = (H + 2 * KVH) * D
    inv_freq = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    freqs = torch.outer(torch.arange(max_pos).float(), inv_freq)
    cache = torch.cat([freqs.cos(), freqs.sin()], dim=-1).to(dtype)
    positions = torch.randint(0, max_pos, (T,), dtype=torch.int64)
    slots = torch.randperm(NB * BS)[:T].to(torch.int64)
    slots[5] = -1                                            # padding token: no cache write
    if sk:
        slabs = torch.randn(sk, T, width)
        qkv = slabs[0].clone()
        for s in range(1, sk):
            qkv = qkv + slabs[s]                             # fp32, slab order
        qkv = qkv.to(dtype)
    else:
        slabs = None
        qkv = torch.randn(T, width).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KVH) * D], qkv[:, (H + KVH) * D:]
    ref_q, ref_k = R.rotary_embedding(positions, q, k, D, cache, True)
    kc, vc = make_kv_cache_x(NB, BS, KVH, D, dtype, seed=4)
    ref_kc, ref_vc = kc.clone(), vc.clone()
    R.reshape_and_cache(ref_k.reshape(T, KVH, D), v.reshape(T, KVH, D), ref_kc, ref_vc, slots)
    d = dev()
    qkv_d = (torch.full((T, width), float("nan")).to(dtype) if sk else qkv).to(d)
    kc_d, vc_d = kc.to(d), vc.to(d)
    ops().qkv_rope_cache(qkv_d, slabs.to(d) if sk else None, sk, positions.to(d), cache.to(d), kc_d, vc_d,
                         slots.to(d), H, KVH, D)
    assert_bit_exact(qkv_d[:, :H * D].contiguous(), ref_q.contiguous(), "q")
    assert_bit_exact(qkv_d[:, H * D:(H + KVH) * D].contiguous(), ref_k.contiguous(), "k")
    if sk:
        assert_bit_exact(qkv_d[:, (H + KVH) * D:].contiguous(), v.contiguous(), "v (from the slabs)")
    assert_bit_exact(kc_d, ref_kc, "key_cache")
    assert_bit_exact(vc_d, ref_vc, "value_cache")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d_", [13, 512, 14336])
@pytest.mark.parametrize("tokens", [1, 83, 70000])
def test_silu_and_mul_quant(dtype, d_, tokens):
    """fp8(silu_and_mul(x) / scale), ref csrc/quantization/activation_kernels.cu:21-127.  Parity
    unpinned beyond the oracle: the reference holds no vectors for this op."""
    if tokens == 70000 and d_ != 13:
        pytest.skip("the > 65535-token case is covered at the small width")
    torch.manual_seed(1)
    x = (torch.randn(tokens, 2 * d_) * 3).to(dtype)
    scale = torch.tensor([0.037])
    d = dev()
    xd = x.to(d)
    out = torch.empty(tokens, d_, dtype=torch.float8_e4m3fn, device=d)
    ops().silu_and_mul_quant(out, xd, scale.to(d))
    # the quantisation step itself is exact given the device's own T-rounded product (whose expf
    # may differ from torch.exp by an ulp, see test_silu_and_mul)
    mid = torch.empty(tokens, d_, dtype=dtype, device=d)
    ops().silu_and_mul(mid, xd)
    inv = np.float32(1.0) / np.float32(0.037)
    want = (mid.float().cpu() * inv).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(out.cpu().view(torch.uint8), want.view(torch.uint8))
    ref = R.silu_and_mul_quant(x, scale)
    diff = (out.cpu().view(torch.uint8) != ref.view(torch.uint8)).float().mean().item()
    assert diff < 2e-3, f"silu_and_mul_quant differs from the oracle in {diff:.2%} of the bytes"
    # saturation: a scale small enough to overflow clamps to +-448
    tiny = torch.tensor([1e-6], device=d)
    ops().silu_and_mul_quant(out, xd, tiny)
    assert out.float().abs().max().item() <= 448.0 and not torch.isnan(out.float()).any()
    # and through the registered op
    import vllm_metax_amd._C  # noqa: F401
    out2 = torch.empty_like(out)
    torch.ops._C.silu_and_mul_quant(out2, xd, tiny)
    assert torch.equal(out.view(torch.uint8), out2.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("n,vocab", [(64, 128256), (3, 1000), (1, 7), (5, 4099)])
def test_greedy_advance(dtype, n, vocab):
    """MI355X-side decode fusion (no reference op): argmax (lowest index among equal maxima) + positions /
    seq_lens += 1 + the next slot from the block table, against the torch ops it replaces."""
    g = torch.Generator().manual_seed(n * 131 + vocab)
    logits = torch.randn(n, vocab, generator=g).to(dtype)
    # ties: the same maximum at several places of a row (and across the 16-byte chunk / wave boundaries)
    for r in range(n):
        cols = torch.randint(0, vocab, (4,), generator=g)
        logits[r, cols] = 9.0
    bs, max_blocks = 16, 12
    bt = torch.randperm(n * max_blocks, generator=g).to(torch.int32).view(n, max_blocks)
    pos = torch.randint(0, bs * max_blocks - 1, (n,), generator=g)
    pos[0] = bs * max_blocks - 1            # a full sequence: the next slot does not exist (index clamped)
    if n > 1:
        pos[1] = bs - 1                     # the next token opens a new block
    seq = (pos + 1).to(torch.int32)
    d = dev()
    lg, btd = logits.to(d), bt.to(d)
    tok = torch.zeros(n, dtype=torch.int64, device=d)
    posd, seqd = pos.to(d), seq.to(d)
    slots = torch.zeros(n, dtype=torch.int64, device=d)
    ops().greedy_advance(lg, tok, posd, seqd, slots, btd, bs)
    ref_tok = torch.stack([(logits[r].float() == logits[r].float().max()).nonzero()[0, 0] for r in range(n)])
    assert torch.equal(tok.cpu(), ref_tok)
    assert torch.equal(posd.cpu(), pos + 1) and torch.equal(seqd.cpu(), seq + 1)
    npos = pos + 1
    bi = (npos // bs).clamp(max=max_blocks - 1)
    ref_slots = bt[torch.arange(n), bi].long() * bs + npos % bs
    assert torch.equal(slots.cpu()[1:], ref_slots[1:])            # (row 0: no next slot, value unspecified)
    # a strided logits view (row stride > vocab)
    wide = torch.zeros(n, vocab + 24, dtype=dtype, device=d)
    wide[:, :vocab] = lg
    tok2 = torch.zeros_like(tok)
    ops().greedy_advance(wide[:, :vocab], tok2, posd, seqd, slots, btd, bs)
    assert torch.equal(tok2.cpu(), ref_tok)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,hidden", [(1024, 4096), (1300, 4096), (300, 2048), (8192, 4096)])
@pytest.mark.parametrize("fused", [False, True])
def test_rms_norm_image_is_bit_identical(dtype, m, hidden, fused):
    """MI355X-side prefill fusion: (fused_add_)rms_norm written as the prefill GEMM's operand image == the
    row-major op followed by the re-tiling, bit for bit (image layout: [row tile 16][k tile 32][64 slots of
    8 elements], slot lr * 16 + (lc ^ g(lr)) = A[16 mt + lc][32 kt + 8 lr ..], rows past the end zero)."""
    g = torch.Generator().manual_seed(m + hidden)
    d = dev()
    x = (torch.randn(m, hidden + 8, generator=g) * 2).to(dtype).to(d)[:, :hidden]     # strided rows
    res = torch.randn(m, hidden, generator=g).to(dtype).to(d)
    w = torch.randn(hidden, generator=g).to(dtype).to(d)
    if fused:
        x_ref, r_ref = x.clone(), res.clone()
        ops().fused_add_rms_norm(x_ref, r_ref, w, 1e-5)
        r2 = res.clone()
        img = ops().fused_add_rms_norm_image(x, r2, w, 1e-5)
        assert_bit_exact(r2, r_ref, "residual")
        ref = x_ref
    else:
        ref = torch.empty(m, hidden, dtype=dtype, device=d)
        ops().rms_norm(ref, x, w, 1e-5)
        img = ops().rms_norm_image(x, w, 1e-5)
    assert img is not None and img.shape == (m, hidden)
    mt, kt = (m + 15) // 16, hidden // 32
    pad = torch.zeros(mt * 16, hidden, dtype=dtype, device=d)
    pad[:m] = ref
    # expected image: piece (mt, kt) slot s holds row 16 mt + lc, k 32 kt + 8 lr .. +7 with s = lr*16 + (lc ^ g(lr))
    t = pad.view(mt, 16, kt, 4, 8)                        # [mt, lc, kt, lr, 8]
    exp = torch.empty(mt, kt, 64, 8, dtype=dtype, device=d)
    for lr in range(4):
        gx = ((lr & 1) * 12) | (lr & 2)
        for lc in range(16):
            exp[:, :, lr * 16 + (lc ^ gx)] = t[:, lc, :, lr]
    assert_bit_exact(img.data.view(mt, kt, 64, 8), exp, "operand image")


def test_norm_image_not_applicable_returns_none():
    d = dev()
    x = torch.randn(64, 4096, device=d).to(torch.bfloat16)
    w = torch.ones(4096, device=d, dtype=torch.bfloat16)
    assert ops().rms_norm_image(x, w, 1e-5) is None                 # decode-sized batch
    x2 = torch.randn(1024, 1024, device=d).to(torch.bfloat16)
    assert ops().rms_norm_image(x2, torch.ones(1024, device=d, dtype=torch.bfloat16), 1e-5) is None
    assert ops().rms_norm_image(x.float().repeat(16, 1), w.float(), 1e-5) is None


def test_greedy_advance_nan_and_inf_rows():
    """torch.argmax order on non-finite rows: a NaN beats every number (first NaN wins), -inf rows give index 0 —
    the token must always be inside the vocabulary (the next step gathers an embedding row with it)."""
    d = dev()
    vocab, bs = 1000, 16
    logits = torch.randn(5, vocab)
    logits[0, :] = float("nan")
    logits[1, 777] = float("nan"); logits[1, 3] = float("inf")
    logits[2, :] = float("-inf")
    logits[3, 500] = float("inf"); logits[3, 900] = float("inf")
    logits[4, 10] = float("nan"); logits[4, 5] = float("nan")
    for dtype in (torch.bfloat16, torch.float32):
        lg = logits.to(dtype).to(d)
        tok = torch.full((5,), -1, dtype=torch.int64, device=d)
        pos = torch.zeros(5, dtype=torch.int64, device=d)
        seq = torch.ones(5, dtype=torch.int32, device=d)
        slots = torch.zeros(5, dtype=torch.int64, device=d)
        bt = torch.arange(5 * 4, dtype=torch.int32, device=d).view(5, 4)
        ops().greedy_advance(lg, tok, pos, seq, slots, bt, bs)
        assert tok.cpu().tolist() == [0, 777, 0, 500, 5]
        assert tok.cpu().tolist() == torch.argmax(lg.float().cpu(), dim=-1).tolist()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,KVH,D", [(100, 2, 128), (1024, 8, 128), (17, 1, 64)])
def test_rotary_reshape_and_cache_is_bit_identical(dtype, T, KVH, D):
    """MI355X-side prefill fusion: the key rows rotated on their way into the cache == rotary_embedding on the keys
    followed by reshape_and_cache (both caches, bit for bit; padded slots skipped; the key tensor itself untouched)."""
    g = torch.Generator().manual_seed(T + D)
    d = dev()
    BS, nb = 16, (T + 15) // 16 + 3
    width = (4 * KVH + 2 * KVH) * D
    qkv = (torch.randn(T, width, generator=g) * 0.7).to(dtype).to(d)
    k = qkv[:, 4 * KVH * D:5 * KVH * D].view(T, KVH, D)            # strided views of a qkv row, as in the model
    v = qkv[:, 5 * KVH * D:].view(T, KVH, D)
    pos = torch.randint(0, 2048, (T,), generator=g).to(d)
    cos_sin = torch.randn(2048, D, generator=g).to(dtype).to(d)
    slots = torch.randperm(nb * BS, generator=g)[:T].to(torch.int64)
    slots[T // 2] = -1
    slots = slots.to(d)
    kc = torch.zeros(nb, KVH, D // 8, BS, 8, dtype=dtype, device=d)
    vc = torch.zeros(nb, KVH, D, BS, dtype=dtype, device=d)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    k_rot = k.clone()
    ops().rotary_embedding(pos, k_rot.view(T, KVH * D), None, D, cos_sin, True)
    ops().reshape_and_cache(k_rot, v, kc_ref, vc_ref, slots)
    k_before = k.clone()
    assert ops().rotary_reshape_and_cache(pos, k, v, kc, vc, slots, cos_sin)
    assert_bit_exact(kc, kc_ref, "key cache")
    assert_bit_exact(vc, vc_ref, "value cache")
    assert_bit_exact(k, k_before, "key rows untouched")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d_", [8, 512, 3584, 14336, 16384])
@pytest.mark.parametrize("tokens", [1, 64, 1100])
def test_silu_and_mul_per_token_quant_equals_the_two_ops(dtype, d_, tokens):
    """MI355X-side fusion for the input of an fp8 down_proj: the bytes and scales of silu_and_mul followed by
    dynamic_per_token_scaled_fp8_quant, bit for bit; shapes it does not take report None."""
    torch.manual_seed(2)
    d = dev()
    x = (torch.randn(tokens, 2 * d_) * 3).to(dtype).to(d)
    x[0, :d_] = 0                       # an all-zero row: the minimum scale applies
    got = ops().silu_and_mul_per_token_quant(x)
    assert got is not None
    mid = torch.empty(tokens, d_, dtype=dtype, device=d)
    ops().silu_and_mul(mid, x)
    want = torch.empty(tokens, d_, dtype=torch.float8_e4m3fn, device=d)
    want_s = torch.empty(tokens, 1, dtype=torch.float32, device=d)
    ops().dynamic_per_token_scaled_fp8_quant(want, mid, want_s, None)
    assert torch.equal(got[1], want_s), "scales"
    assert torch.equal(got[0].view(torch.uint8), want.view(torch.uint8)), "fp8 bytes"
    assert ops().silu_and_mul_per_token_quant(torch.zeros(4, 2 * 20, dtype=dtype, device=d)) is None      # d % 8
    assert ops().silu_and_mul_per_token_quant(torch.zeros(2, 2 * 16392, dtype=dtype, device=d)) is None   # d > 16384
