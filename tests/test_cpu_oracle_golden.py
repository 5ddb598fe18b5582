"""CPU: the oracle (oracle/ref_ops.py) reproduces the committed golden fixtures bit for bit,
and the independently written C restatement (oracle/cpu_port.c) agrees with it.

There are no reference-held known-answer vectors for these kernels: the fixtures
freeze the oracle's restatement of the cited reference sources ("parity unpinned" against the
reference binary).  The C port is a second implementation written from the same sources, so
agreement of the two is the available cross-check.
"""
import numpy as np
import pytest
import torch

from tests import golden_io as G
from tests.util import assert_bit_exact

from oracle import ref_ops as R


def test_reshape_and_cache_golden():
    z = G.load("reshape_and_cache")
    key, value = G.bf16(z["key"]), G.bf16(z["value"])
    kc = torch.zeros(z["key_cache"].shape, dtype=torch.bfloat16)
    vc = torch.zeros(z["value_cache"].shape, dtype=torch.bfloat16)
    R.reshape_and_cache(key, value, kc, vc, G.i64(z["slots"]))
    assert_bit_exact(kc, G.bf16(z["key_cache"]), "key_cache")
    assert_bit_exact(vc, G.bf16(z["value_cache"]), "value_cache")
    # C port
    from oracle import cpu_port as C
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    C.reshape_and_cache(key, value, kc2, vc2, G.i64(z["slots"]))
    assert_bit_exact(kc2, kc, "C port key_cache")
    assert_bit_exact(vc2, vc, "C port value_cache")


def test_paged_attention_golden():
    z = G.load("paged_attention")
    q, kc, vc = G.bf16(z["q"]), G.bf16(z["key_cache"]), G.bf16(z["value_cache"])
    bt, sl = G.i32(z["block_tables"]), G.i32(z["seq_lens"])
    scale = q.shape[-1] ** -0.5
    o1 = R.paged_attention_v1(q, kc, vc, kc.shape[1], scale, bt, sl)
    assert_bit_exact(o1, G.bf16(z["out_v1"]), "v1")
    o2, es, ml, tmp = R.paged_attention_v2(q, kc, vc, kc.shape[1], scale, bt, sl, int(sl.max()))
    assert_bit_exact(o2, G.bf16(z["out_v2"]), "v2")
    assert_bit_exact(tmp, G.bf16(z["tmp_out"]), "tmp_out")
    assert np.array_equal(es.numpy(), z["exp_sums"]) and np.array_equal(ml.numpy(), z["max_logits"])
    # v1 and v2 are the same attention up to the partition-wise rounding of probabilities
    assert (o1.float() - o2.float()).abs().max() <= 2.0 ** -7 * o1.float().abs().max()
    # C port (its own loop order): within one output ulp of the oracle
    from oracle import cpu_port as C
    oc = C.paged_attention_v1(q, kc, vc, kc.shape[1], scale, bt, sl)
    err = (oc.float() - o1.float()).abs().max().item()
    assert err <= 2.0 ** -7 * o1.float().abs().max().item(), err


def test_paged_prefill_golden():
    z, za = G.load("paged_prefill"), G.load("paged_attention")
    kc, vc = G.bf16(za["key_cache"]), G.bf16(za["value_cache"])
    q = G.bf16(z["q"])
    o = R.paged_prefill_attention(q, kc, vc, kc.shape[1], q.shape[-1] ** -0.5, G.i32(z["block_tables"]),
                                  G.i32(z["seq_lens"]), G.i32(z["cu_seqlens_q"]))
    assert_bit_exact(o, G.bf16(z["out"]), "prefill")
    # the last query row of sequence 0 equals single-query attention over the same 100 keys
    o1 = R.paged_attention_v1(q[39:40], kc, vc, kc.shape[1], q.shape[-1] ** -0.5,
                              G.i32(z["block_tables"])[:1], torch.tensor([100], dtype=torch.int32))
    assert (o[39].float() - o1[0].float()).abs().max() <= 2.0 ** -6 * o1.float().abs().max()


def test_layernorm_and_fp8_quant_golden():
    z = G.load("layernorm_quant")
    x, res, w = G.bf16(z["x"]), G.bf16(z["residual"]), G.bf16(z["weight"])
    scale = G.f32(z["scale"])
    assert_bit_exact(R.rms_norm(x, w, 1e-5), G.bf16(z["rms_norm"]), "rms_norm")
    n2, r2 = R.fused_add_rms_norm(x, res, w, 1e-5)
    assert_bit_exact(n2, G.bf16(z["fused_out"]), "fused out")
    assert_bit_exact(r2, G.bf16(z["fused_residual"]), "fused residual")
    assert_bit_exact(R.rms_norm_static_fp8_quant(x, w, scale, 1e-5), G.fp8(z["static_q"]), "static q")
    q2, _ = R.fused_add_rms_norm_static_fp8_quant(x, res, w, scale, 1e-5)
    assert_bit_exact(q2, G.fp8(z["fused_static_q"]), "fused static q")
    q3, s3, r4 = R.rms_norm_dynamic_per_token_quant(x, w, 1e-5, None, res)
    assert_bit_exact(q3, G.fp8(z["dyn_q"]), "dyn q")
    assert np.array_equal(s3.numpy(), z["dyn_scales"])
    assert_bit_exact(r4, G.bf16(z["dyn_residual"]), "dyn residual")
    assert_bit_exact(R.static_scaled_fp8_quant(x, scale), G.fp8(z["fp8_static"]), "fp8 static")
    dq, ds = R.dynamic_scaled_fp8_quant(x)
    assert_bit_exact(dq, G.fp8(z["fp8_dynamic"]), "fp8 dynamic")
    pq, ps = R.dynamic_per_token_scaled_fp8_quant(x)
    assert_bit_exact(pq, G.fp8(z["fp8_token"]), "fp8 per token")
    # identities the reference's tests rely on (test_layernorm.py:104-180): fused == unfused
    assert_bit_exact(R.rms_norm_static_fp8_quant(x, w, scale, 1e-5),
                     R.static_scaled_fp8_quant(R.rms_norm(x, w, 1e-5), scale), "fused == unfused")
    # C port
    from oracle import cpu_port as C
    assert_bit_exact(C.rms_norm(x, w, 1e-5), G.bf16(z["rms_norm"]), "C rms_norm")
    xc, rc = x.clone(), res.clone()
    C.fused_add_rms_norm(xc, rc, w, 1e-5)
    assert_bit_exact(xc, G.bf16(z["fused_out"]), "C fused out")
    assert_bit_exact(rc, G.bf16(z["fused_residual"]), "C fused residual")


def test_rotary_silu_golden():
    z = G.load("rotary_silu")
    pos, q, k, cache = G.i64(z["positions"]), G.bf16(z["q"]), G.bf16(z["k"]), G.bf16(z["cache"])
    qn, kn = R.rotary_embedding(pos, q, k, 128, cache, True)
    assert_bit_exact(qn, G.bf16(z["q_neox"]), "q neox")
    assert_bit_exact(kn, G.bf16(z["k_neox"]), "k neox")
    qj, kj = R.rotary_embedding(pos, q, k, 128, cache, False)
    assert_bit_exact(qj, G.bf16(z["q_gptj"]), "q gptj")
    assert_bit_exact(kj, G.bf16(z["k_gptj"]), "k gptj")
    xs = G.bf16(z["silu_in"])
    assert_bit_exact(R.silu_and_mul(xs), G.bf16(z["silu_out"]), "silu")
    # rotation preserves the norm of each (x, y) pair up to bf16 rounding
    n0 = q.float().reshape(9, 4, 128).pow(2).sum(-1)
    n1 = qn.float().reshape(9, 4, 128).pow(2).sum(-1)
    assert ((n0 - n1).abs() / n0).max() < 2e-2
    from oracle import cpu_port as C
    qc, kcc = q.clone(), k.clone()
    C.rotary_neox(pos, qc, kcc, cache, 4, 2, 128)
    assert_bit_exact(qc, qn, "C rotary q")
    assert_bit_exact(kcc, kn, "C rotary k")
    sc = C.silu_and_mul(xs)
    d = (sc.view(torch.int16).int() - G.bf16(z["silu_out"]).view(torch.int16).int()).abs()
    assert d.max() <= 2 and (d > 0).float().mean() < 5e-3     # libm expf vs torch.exp last bit


def test_w4a16_golden():
    z = G.load("w4a16")
    qw, qz, sc, x = G.i32(z["awq_qweight"]), G.i32(z["awq_qzeros"]), G.bf16(z["scales"]), G.bf16(z["x"])
    q2 = R.awq_to_gptq_4bit(qw)
    assert np.array_equal(q2.numpy(), z["awq_repacked"])
    assert_bit_exact(R.awq_dequantize(qw, sc, qz), G.bf16(z["awq_dequant"]), "awq dequant")
    assert_bit_exact(R.awq_gemm(x, q2, sc, qz), G.bf16(z["awq_gemm"]), "awq gemm")
    gq, gz, perm = G.i32(z["gptq_qweight"]), G.i32(z["gptq_qzeros"]), G.i32(z["perm"])
    gs = R.gptq_shuffle(gq, None)
    assert np.array_equal(gs.numpy(), z["gptq_shuffled"])
    gsp = R.gptq_shuffle(gq, perm)
    assert np.array_equal(gsp.numpy(), z["gptq_shuffled_perm"])
    assert_bit_exact(R.gptq_gemm(x, gs, gz, sc, None, 128), G.bf16(z["gptq_gemm"]), "gptq gemm")
    assert_bit_exact(R.gptq_gemm(x, gsp, gz, sc, perm, 128), G.bf16(z["gptq_gemm_perm"]), "gptq perm")
    # both packings hold the same integer matrix: AWQ repack == GPTQ shuffle of the same W
    assert np.array_equal(q2.numpy().reshape(64, 128), gs.numpy())
    # the awq gemm equals X @ awq_dequantize(W) (test_awq_triton.py:160-172's definition)
    dense = (x.double() @ G.bf16(z["awq_dequant"]).double()).to(torch.bfloat16)
    assert_bit_exact(dense, G.bf16(z["awq_gemm"]), "gemm == x @ dequant")
    # C port
    from oracle import cpu_port as C
    got = C.w4a16_gemm(x, q2.reshape(64, 128), sc, qz, 0, 128)
    d = (got.view(torch.int16).int() - G.bf16(z["awq_gemm"]).view(torch.int16).int()).abs()
    assert d.max() <= 1 and (d > 0).float().mean() < 2e-2     # fp32 vs fp64 accumulation
    got2 = C.w4a16_gemm(x, gs, sc, gz, 1, 128)
    d2 = (got2.view(torch.int16).int() - G.bf16(z["gptq_gemm"]).view(torch.int16).int()).abs()
    assert d2.max() <= 1 and (d2 > 0).float().mean() < 2e-2


def test_scaled_mm_fp8_golden():
    z = G.load("scaled_mm_fp8")
    a, b = G.fp8(z["a"]), G.fp8(z["b_nk"]).t()
    out = R.scaled_mm_fp8(a, b, G.f32(z["a_scales"]), G.f32(z["b_scales"]), torch.bfloat16)
    assert_bit_exact(out, G.bf16(z["out"]), "scaled_mm")


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_pack_unpack_roundtrips(seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 16, size=(64, 32), dtype=np.uint8)
    assert np.array_equal(R.awq_unpack(R.awq_pack(w)), w)
    assert np.array_equal(R.gptq_unpack_rows(R.gptq_pack_rows(w)), w)
    assert np.array_equal(R.exllama_unpack(R.gptq_shuffle(R.gptq_pack_rows(w), None).numpy()), w)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(seed)).to(torch.int32)
    assert np.array_equal(R.exllama_unpack(R.gptq_shuffle(R.gptq_pack_rows(w), perm).numpy()),
                          w[perm.numpy()])


def test_merge_attn_states_oracle_matches_the_in_test_formula():
    """oracle.merge_attn_states vs the plain formula of the reference's own test
    (tests/kernels/attention/test_merge_attn_states.py:16-45): out = p*p_scale + s*s_scale with the
    +inf -> -inf convention; they may differ only by the fma's single rounding."""
    g = torch.Generator().manual_seed(1)
    n, h, d = 97, 8, 64
    p_out, s_out = torch.randn(n, h, d, generator=g), torch.randn(n, h, d, generator=g)
    p_lse, s_lse = torch.randn(h, n, generator=g) * 3, torch.randn(h, n, generator=g) * 3
    p_lse[0, :5] = float("inf")
    s_lse[1, 5:9] = float("inf")
    out, lse = R.merge_attn_states(p_out, p_lse, s_out, s_lse)
    pl, sl = p_lse.clone(), s_lse.clone()
    pl[pl == float("inf")] = -float("inf")
    sl[sl == float("inf")] = -float("inf")
    m = torch.maximum(pl, sl)
    pe, se = torch.exp(pl - m), torch.exp(sl - m)
    want = p_out * (pe / (pe + se)).t().unsqueeze(2) + s_out * (se / (pe + se)).t().unsqueeze(2)
    assert torch.allclose(out, want, rtol=1e-6, atol=1e-7)
    assert torch.allclose(lse, torch.log(pe + se) + m, rtol=1e-6, atol=1e-7)
    # a +inf lse removes that side entirely
    assert torch.equal(out[:5, 0], s_out[:5, 0]) and torch.equal(out[5:9, 1], p_out[5:9, 1])


def test_golden_merge_attn_states():
    """The frozen fixture pins the oracle (tests/golden/make_golden.py merge_attn_states)."""
    z = G.load("merge_attn_states")
    out, lse = R.merge_attn_states(G.bf16(z["prefix_output"]), G.f32(z["prefix_lse"]),
                                   G.bf16(z["suffix_output"]), G.f32(z["suffix_lse"]))
    assert torch.equal(out.view(torch.int16), G.bf16(z["output"]).view(torch.int16))
    assert torch.equal(lse, G.f32(z["output_lse"]))


def test_golden_int8_w8a8():
    """Frozen int8 fixtures pin the oracle (tests/golden/make_golden.py int8_w8a8)."""
    z = G.load("int8_w8a8")
    x = G.bf16(z["x"])
    q, s = R.scaled_int8_quant(x)
    assert np.array_equal(q.numpy(), z["q_dynamic"]) and np.array_equal(s.numpy(), z["scales_dynamic"])
    assert list(z["q_dynamic"][0, :4]) == [round(v) for v in (0.5 * 127 / float(x[0].float().abs().max()),
                                                               1.5 * 127 / float(x[0].float().abs().max()),
                                                               -2.5 * 127 / float(x[0].float().abs().max()), 0.0)]
    qs, _ = R.scaled_int8_quant(x, G.f32(z["scale_static"]))
    assert np.array_equal(qs.numpy(), z["q_static"])
    w = torch.from_numpy(z["w_nk"].copy())
    out = R.scaled_mm_int8(q, w.t(), s, G.f32(z["w_scales"]), torch.bfloat16, G.bf16(z["bias"]))
    assert torch.equal(out.view(torch.int16), G.bf16(z["out"]).view(torch.int16))
