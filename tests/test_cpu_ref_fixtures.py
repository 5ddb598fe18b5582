"""CPU: the oracle (oracle/ref_ops.py) against fixtures produced by the REFERENCE'S OWN torch test
references (tests/golden/ref_*.npz, generator tests/golden/make_ref_fixtures.py — it AST-extracts
the functions from /root/reference/tests/kernels/... in the build container; only the .npz travel).

This is what pins the oracle: each assertion below compares the restatement with outputs the
reference's code computed on the same inputs.  Tolerances: bit-exact where the reference function
is exact arithmetic (AWQ dequantize, fp8 / int8 quantisation); the reference's own test tolerance
for its 16-bit evaluation (cited per test); and north_star's bound — max|err| <= 1e-3 * max|ref|
plus one rounding of the 16-bit output — against the fp32 evaluation of the same function.
"""
import hashlib

import numpy as np
import pytest
import torch

from tests import ref_inputs as RI
from tests.ref_inputs import BF, F16, F32, FP8, DT
from tests.util import assert_bit_exact

from oracle import ref_ops as R

EPS = {BF: 2.0 ** -8, F16: 2.0 ** -11, F32: 2.0 ** -24}      # half an ulp, relative


def close_to_f32(got, ref32, what, rel=1e-3):
    """max|got - ref| <= rel * max|ref| + (one rounding of got's dtype at max|ref|)."""
    g, r = got.double(), ref32.double()
    assert torch.isfinite(g).all(), what
    bound = (rel + EPS[got.dtype]) * r.abs().max().item()
    err = (g - r).abs().max().item()
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"
    return err / max(r.abs().max().item(), 1e-30)


def sha(t):
    t = t.detach().contiguous()
    v = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[t.element_size()]
    return hashlib.sha256(t.view(v).numpy().tobytes()).hexdigest()


# ------------------------------------------------------------------------------ a1 / a2
@pytest.mark.parametrize("name", ["ref_paged_attention_grid_small_heads", "ref_paged_attention_grid_large_heads"])
def test_paged_attention_reference_grid(name):
    """Reference grid (test_attention.py:29-47) at the reference's tolerance atol 1e-3 / rtol 1e-5
    (:337-345), and at 1e-3 * max|ref| + one bf16 rounding."""
    z, meta = RI.load(name)
    for i, m in enumerate(meta):
        q, kc, vc, bt, sl, slopes = RI.decode_inputs(m)
        RI.check_crc(m, q=q, kc=kc, vc=vc)
        ref = RI.arr(z, i, "out", BF)
        o1 = R.paged_attention_v1(q, kc, vc, m["KVH"], m["scale"], bt, sl, slopes)
        torch.testing.assert_close(o1.float(), ref.float(), atol=1e-3, rtol=1e-5)
        close_to_f32(o1, ref.float(), f"{name}[{i}] v1", rel=1e-3 + 2.0 ** -8)    # ref itself is bf16-rounded
        if i % 4 == 0:
            o2 = R.paged_attention_v2(q, kc, vc, m["KVH"], m["scale"], bt, sl, int(sl.max()), slopes)[0]
            torch.testing.assert_close(o2.float(), ref.float(), atol=1e-3, rtol=1e-5)


def test_paged_attention_reference_sharp():
    """Unit-variance q/k (logits of order 1): oracle v1 and v2 vs the fp32 evaluation of the
    reference function; the reference's own 16-bit evaluation is itself only within ~1e-2 of that
    (it rounds q.k to 16 bit before the softmax), which the last assertion records."""
    z, meta = RI.load("ref_paged_attention_sharp")
    for i, m in enumerate(meta):
        q, kc, vc, bt, sl, slopes = RI.decode_inputs(m)
        RI.check_crc(m, q=q, kc=kc, vc=vc)
        ref32 = RI.arr(z, i, "out_f32")
        o1 = R.paged_attention_v1(q, kc, vc, m["KVH"], m["scale"], bt, sl, slopes)
        close_to_f32(o1, ref32, f"sharp[{i}] v1")
        o2 = R.paged_attention_v2(q, kc, vc, m["KVH"], m["scale"], bt, sl, int(sl.max()), slopes)[0]
        close_to_f32(o2, ref32, f"sharp[{i}] v2")
        ref16 = RI.arr(z, i, "out", DT[m["dtype"]])
        close_to_f32(ref16, ref32, f"sharp[{i}] reference 16-bit vs its fp32 evaluation", rel=3e-2)


# ------------------------------------------------------------------------------ a13
@pytest.mark.parametrize("name", ["ref_flash_paged_plain", "ref_flash_paged_opts"])
def test_paged_prefill_reference(name):
    """ref_paged_attn (test_flash_attn.py:27-80) on its own (query_len, kv_len) grid: the reference's
    tolerance atol 1.5e-2 / rtol 1e-2 (:183) against its bf16 evaluation; 1e-3 * max|ref| + one bf16
    rounding against its fp32 evaluation."""
    z, meta = RI.load(name)
    for i, m in enumerate(meta):
        q, kc_nhd, vc_nhd, bt = RI.prefill_inputs(m)
        RI.check_crc(m, q=q, kc=kc_nhd, vc=vc_nhd)
        kc, vc = RI.nhd_to_xsplit(kc_nhd, vc_nhd)
        cu = torch.tensor([0] + m["query_lens"], dtype=torch.int32).cumsum(0).to(torch.int32)
        sl = torch.tensor(m["kv_lens"], dtype=torch.int32)
        o = R.paged_prefill_attention(q, kc, vc, m["KVH"], m["scale"], bt, sl, cu, m["window"], m["softcap"])
        if f"c{i}_out" in z:
            torch.testing.assert_close(o.float(), RI.arr(z, i, "out", BF).float(), atol=1.5e-2, rtol=1e-2)
        if f"c{i}_out_f32" in z:
            close_to_f32(o, RI.arr(z, i, "out_f32"), f"{name}[{i}]")


# ------------------------------------------------------------------------------ a5
def test_awq_dequantize_reference_bit_exact():
    """awq_dequantize_torch (test_awq_triton.py:36-62): bit-exact (stored output or its SHA-256)."""
    z, meta = RI.load("ref_awq")
    for i, m in enumerate(meta):
        if m["kind"] != "dequantize":
            continue
        qw, qz, sc, _ = RI.awq_inputs(m)
        RI.check_crc(m, qw=qw, sc=sc)
        w = R.awq_dequantize(qw, sc, qz)
        assert sha(w) == m["out_sha"], f"dequantize case {i}: {m}"
        if f"c{i}_out" in z:
            assert_bit_exact(w, RI.arr(z, i, "out", DT[m["dtype"]]), f"dequantize[{i}]")


def test_awq_gemm_reference():
    """matmul(x, awq_dequantize_torch(...)) (test_awq_triton.py:160-172, fp32): the oracle's
    repack (awq_to_gptq_4bit) + GEMM within one output rounding + 2e-4 * max|ref| (the reference's
    own tolerance there is 1e-1)."""
    z, meta = RI.load("ref_awq")
    for i, m in enumerate(meta):
        if m["kind"] != "gemm":
            continue
        qw, qz, sc, x = RI.awq_inputs(m)
        RI.check_crc(m, qw=qw, sc=sc, x=x)
        assert sha(R.awq_dequantize(qw, sc, qz)) == m["w_sha"]
        out = R.awq_gemm(x, R.awq_to_gptq_4bit(qw), sc, qz)
        close_to_f32(out, RI.arr(z, i, "out_f32"), f"awq gemm[{i}] {m}", rel=2e-4)


# ------------------------------------------------------------------------------ f2
def test_merge_attn_states_reference():
    """merge_attn_states_torch (test_merge_attn_states.py:16-45); its test compares at 1e-3 (fp32) /
    1e-2 (16-bit) (:150-…); here: lse to 2e-6, output to one rounding + 1e-5 vs the fp32 evaluation."""
    z, meta = RI.load("ref_merge_attn_states")
    for i, m in enumerate(meta):
        p_out, p_lse, s_out, s_lse = RI.merge_inputs(m)
        RI.check_crc(m, p_out=p_out, p_lse=p_lse, s_lse=s_lse)
        out, lse = R.merge_attn_states(p_out, p_lse, s_out, s_lse)
        ref_lse = RI.arr(z, i, "out_lse")
        fin = torch.isfinite(ref_lse)
        assert torch.equal(torch.isfinite(lse), fin)
        assert (lse[fin] - ref_lse[fin]).abs().max() <= 2e-6 * ref_lse[fin].abs().max()
        close_to_f32(out, RI.arr(z, i, "out_f32"), f"merge[{i}]", rel=1e-5)


# ------------------------------------------------------------------------------ a11 / f4
def test_scaled_mm_reference():
    """baseline_scaled_mm (tests/kernels/utils.py:1231-1270) on operands built as
    test_cutlass_scaled_mm.py:68-100.  int8: exact accumulation, so the 16-bit output may differ from
    the reference's only by a rounding flip (fp32 product order); fp8 likewise."""
    z, meta = RI.load("ref_scaled_mm")
    for i, m in enumerate(meta):
        a, bt, a_s, b_s, bias = RI.scaled_mm_operands(m)
        RI.check_crc(m, a=a, bt=bt)
        odt = DT[m["out_dtype"]]
        fn = R.scaled_mm_fp8 if m["kind"] == "fp8" else R.scaled_mm_int8
        out = fn(a, bt.t(), a_s, b_s, odt, bias)
        ref = RI.arr(z, i, "out", odt)
        close_to_f32(out, ref.float(), f"scaled_mm[{i}] {m}", rel=2e-4 + EPS[odt])
        frac = (out.view(torch.int16) != ref.view(torch.int16)).double().mean().item()
        assert frac <= 0.05, f"scaled_mm[{i}]: {frac:.2%} of outputs differ from the reference's"
        if f"c{i}_out_f32" in z:
            close_to_f32(out, RI.arr(z, i, "out_f32"), f"scaled_mm[{i}] vs fp32", rel=2e-4)


# ------------------------------------------------------------------------------ a9 / f4
def test_dynamic_quant_reference_bit_exact():
    """ref_dynamic_per_token_quant / ref_dynamic_per_tensor_fp8_quant (tests/kernels/quant_utils.py:
    22-97): scales and quantised bytes bit for bit."""
    z, meta = RI.load("ref_dynamic_quant")
    for i, m in enumerate(meta):
        x = RI.quant_input(m)
        RI.check_crc(m, x=x)
        ref_s = RI.arr(z, i, "scales")
        if m["kind"] == "per_token_fp8":
            ub = torch.tensor([m["scale_ub"]], dtype=F32) if m["scale_ub"] is not None else None
            q, s = R.dynamic_per_token_scaled_fp8_quant(x, ub)
            assert_bit_exact(q, RI.arr(z, i, "q", FP8), f"quant[{i}] q")
        elif m["kind"] == "per_tensor_fp8":
            q, s = R.dynamic_scaled_fp8_quant(x)
            assert_bit_exact(q, RI.arr(z, i, "q", FP8), f"quant[{i}] q")
        else:
            q, s = R.scaled_int8_quant(x)
            # the reference clamps to the int8 range [-128, 127]; the kernel (and the oracle, following
            # int8_quant_kernels.cu:12-22) to [-127, 127] — identical unless x * (127/absmax) rounds
            # below -127, which it cannot
            assert_bit_exact(q, RI.arr(z, i, "q"), f"quant[{i}] q")
        assert np.array_equal(s.reshape(-1).numpy(), ref_s.reshape(-1).numpy()), f"quant[{i}] scales"


# ------------------------------------------------------------------------------ f3 (fp8 KV cache)
def test_paged_attention_fp8kv_reference():
    """fp8 KV cache, the reference's test procedure (test_attention.py:303-328): the cache is
    dequantised (scale 1.0, exact) and its torch attention reference evaluated; the oracle's fp8 read
    path (fp8_dequant + paged_attention_v1/v2) must agree with that."""
    z, meta = RI.load("ref_paged_attention_fp8kv")
    for i, m in enumerate(meta):
        q, kc8, vc8, bt, sl, slopes = RI.decode_inputs_fp8(m)
        RI.check_crc(m, q=q, kc=kc8, vc=vc8)
        kc, vc = R.fp8_dequant(kc8, 1.0, q.dtype), R.fp8_dequant(vc8, 1.0, q.dtype)
        ref32 = RI.arr(z, i, "out_f32")
        close_to_f32(R.paged_attention_v1(q, kc, vc, m["KVH"], m["scale"], bt, sl, slopes), ref32, f"fp8kv[{i}] v1")
        close_to_f32(R.paged_attention_v2(q, kc, vc, m["KVH"], m["scale"], bt, sl, int(sl.max()), slopes)[0],
                     ref32, f"fp8kv[{i}] v2")
