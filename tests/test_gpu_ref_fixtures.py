"""GPU: the HIP kernels (through the C-ABI) against fixtures produced by the REFERENCE'S OWN torch
test references (tests/golden/ref_*.npz; generator tests/golden/make_ref_fixtures.py).  Nothing under
oracle/ is used here: the expected values are the reference code's outputs.  Tolerances as in
tests/test_cpu_ref_fixtures.py (bit-exact for dequantize / quantisation; the reference's own test
tolerance vs its 16-bit evaluation; 1e-3 * max|ref| + one output rounding vs its fp32 evaluation).
"""
import numpy as np
import pytest
import torch

from tests import ref_inputs as RI
from tests.ref_inputs import BF, F16, F32, FP8, DT
from tests.test_cpu_ref_fixtures import EPS, close_to_f32, sha
from tests.util import assert_bit_exact, dev

pytestmark = pytest.mark.gpu


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _decode(m, q, kc, vc, bt, sl, slopes, version):
    d = dev()
    S, H, D = q.shape
    max_len = int(sl.max())
    out = torch.full_like(q, float("nan"), device=d)
    sl_d, bt_d = sl.to(d), bt.to(d)
    al = slopes.to(d) if slopes is not None else None
    if version == 1:
        ops().paged_attention_v1(out, q.to(d), kc.to(d), vc.to(d), m["KVH"], m["scale"], bt_d, sl_d,
                                 m["bs"], max_len, al, "auto")
    else:
        P = (max_len + 511) // 512
        tmp = torch.empty(S, H, P, D, dtype=q.dtype, device=d)
        es = torch.empty(S, H, P, dtype=torch.float32, device=d)
        ml = torch.empty_like(es)
        ops().paged_attention_v2(out, es, ml, tmp, q.to(d), kc.to(d), vc.to(d), m["KVH"], m["scale"], bt_d,
                                 sl_d, m["bs"], max_len, al, "auto")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("name", ["ref_paged_attention_grid_small_heads", "ref_paged_attention_grid_large_heads"])
def test_paged_attention_reference_grid(name):
    z, meta = RI.load(name)
    for i, m in enumerate(meta):
        q, kc, vc, bt, sl, slopes = RI.decode_inputs(m)
        RI.check_crc(m, q=q, kc=kc, vc=vc)
        ref = RI.arr(z, i, "out", BF)
        for version in (1, 2):
            o = _decode(m, q, kc, vc, bt, sl, slopes, version)
            torch.testing.assert_close(o.float(), ref.float(), atol=1e-3, rtol=1e-5)     # test_attention.py:337-345
            close_to_f32(o, ref.float(), f"{name}[{i}] v{version}", rel=1e-3 + 2.0 ** -8)


def test_paged_attention_reference_sharp():
    z, meta = RI.load("ref_paged_attention_sharp")
    for i, m in enumerate(meta):
        q, kc, vc, bt, sl, slopes = RI.decode_inputs(m)
        RI.check_crc(m, q=q, kc=kc, vc=vc)
        ref32 = RI.arr(z, i, "out_f32")
        for version in (1, 2):
            close_to_f32(_decode(m, q, kc, vc, bt, sl, slopes, version), ref32, f"sharp[{i}] v{version}")


@pytest.mark.parametrize("name", ["ref_flash_paged_plain", "ref_flash_paged_opts"])
def test_paged_prefill_reference(name):
    """ref_flash_paged_opts: sliding window 256 and / or soft-cap 50 (test_flash_attn.py:23-24 grid)."""
    z, meta = RI.load(name)
    d = dev()
    for i, m in enumerate(meta):
        q, kc_nhd, vc_nhd, bt = RI.prefill_inputs(m)
        RI.check_crc(m, q=q, kc=kc_nhd, vc=vc_nhd)
        kc, vc = RI.nhd_to_xsplit(kc_nhd, vc_nhd)
        cu = torch.tensor([0] + m["query_lens"], dtype=torch.int32).cumsum(0).to(torch.int32)
        sl = torch.tensor(m["kv_lens"], dtype=torch.int32)
        out = torch.full(q.shape, float("nan"), dtype=BF, device=d)
        ops().paged_prefill_attention(out, q.to(d), kc.to(d), vc.to(d), m["KVH"], m["scale"], bt.to(d), sl.to(d),
                                      cu.to(d), max(m["query_lens"]), m["bs"], "auto", None, None,
                                      m["window"], m["softcap"])
        torch.cuda.synchronize()
        o = out.cpu()
        if f"c{i}_out" in z:
            torch.testing.assert_close(o.float(), RI.arr(z, i, "out", BF).float(), atol=1.5e-2, rtol=1e-2)
        if f"c{i}_out_f32" in z:
            # kernel: unnormalised probabilities rounded to bf16 (flash style): 2e-3, as the oracle-based test
            close_to_f32(o, RI.arr(z, i, "out_f32"), f"prefill[{i}]", rel=2e-3)


def test_awq_dequantize_reference_bit_exact():
    z, meta = RI.load("ref_awq")
    d = dev()
    for i, m in enumerate(meta):
        if m["kind"] != "dequantize":
            continue
        qw, qz, sc, _ = RI.awq_inputs(m)
        RI.check_crc(m, qw=qw, sc=sc)
        w = ops().awq_dequantize(qw.to(d), sc.to(d), qz.to(d), 0, 0, 0).cpu()
        assert sha(w) == m["out_sha"], f"dequantize case {i}: {m}"


def test_awq_gemm_reference():
    z, meta = RI.load("ref_awq")
    d = dev()
    for i, m in enumerate(meta):
        if m["kind"] != "gemm":
            continue
        qw, qz, sc, x = RI.awq_inputs(m)
        RI.check_crc(m, qw=qw, sc=sc, x=x)
        qg = ops().awq_to_gptq_4bit(qw.to(d))
        # wrapper order (qzeros, scales): vllm_metax/_custom_ops.py:7-22
        out = ops().awq_gemm(x.to(d), qg, qz.to(d), sc.to(d), 0,
                             torch.zeros(m["M"], m["N"], dtype=torch.float32, device=d), m["dtype"] == "bfloat16")
        torch.cuda.synchronize()
        close_to_f32(out.cpu(), RI.arr(z, i, "out_f32"), f"awq gemm[{i}] {m}", rel=2e-4)


def test_merge_attn_states_reference():
    z, meta = RI.load("ref_merge_attn_states")
    d = dev()
    for i, m in enumerate(meta):
        p_out, p_lse, s_out, s_lse = RI.merge_inputs(m)
        RI.check_crc(m, p_out=p_out, p_lse=p_lse, s_lse=s_lse)
        out = torch.full(p_out.shape, float("nan")).to(p_out.dtype).to(d)
        lse = torch.full(p_lse.shape, float("nan"), dtype=F32, device=d)
        ops().merge_attn_states(out, p_out.to(d), p_lse.to(d), s_out.to(d), s_lse.to(d), lse)
        torch.cuda.synchronize()
        ref_lse = RI.arr(z, i, "out_lse")
        fin = torch.isfinite(ref_lse)
        lse = lse.cpu()
        assert (lse[fin] - ref_lse[fin]).abs().max() <= 2e-6 * ref_lse[fin].abs().max()
        close_to_f32(out.cpu(), RI.arr(z, i, "out_f32"), f"merge[{i}]", rel=1e-5)


def test_scaled_mm_reference():
    z, meta = RI.load("ref_scaled_mm")
    d = dev()
    for i, m in enumerate(meta):
        a, bt, a_s, b_s, bias = RI.scaled_mm_operands(m)
        RI.check_crc(m, a=a, bt=bt)
        odt = DT[m["out_dtype"]]
        out = torch.empty(m["m"], m["n"], dtype=odt, device=d)
        ops().cutlass_scaled_mm(out, a.to(d), bt.to(d).t(), a_s.to(d), b_s.to(d), bias.to(d) if bias is not None else None)
        torch.cuda.synchronize()
        o = out.cpu()
        ref = RI.arr(z, i, "out", odt)
        close_to_f32(o, ref.float(), f"scaled_mm[{i}] {m}", rel=2e-4 + EPS[odt])
        frac = (o.view(torch.int16) != ref.view(torch.int16)).double().mean().item()
        assert frac <= 0.05, f"scaled_mm[{i}]: {frac:.2%} of outputs differ from the reference's"
        if f"c{i}_out_f32" in z:
            close_to_f32(o, RI.arr(z, i, "out_f32"), f"scaled_mm[{i}] vs fp32", rel=2e-4)


def test_dynamic_quant_reference_bit_exact():
    z, meta = RI.load("ref_dynamic_quant")
    d = dev()
    for i, m in enumerate(meta):
        x = RI.quant_input(m)
        RI.check_crc(m, x=x)
        ref_s = RI.arr(z, i, "scales")
        xd = x.to(d)
        if m["kind"] == "per_token_fp8":
            q = torch.empty(x.shape, dtype=FP8, device=d)
            s = torch.empty(x.shape[0], 1, dtype=F32, device=d)
            ub = torch.tensor([m["scale_ub"]], dtype=F32, device=d) if m["scale_ub"] is not None else None
            ops().dynamic_per_token_scaled_fp8_quant(q, xd, s, ub)
            assert_bit_exact(q.cpu(), RI.arr(z, i, "q", FP8), f"quant[{i}] q")
        elif m["kind"] == "per_tensor_fp8":
            q = torch.empty(x.shape, dtype=FP8, device=d)
            s = torch.zeros(1, dtype=F32, device=d)
            ops().dynamic_scaled_fp8_quant(q, xd, s)
            assert_bit_exact(q.cpu(), RI.arr(z, i, "q", FP8), f"quant[{i}] q")
        else:
            q, s, _ = ops().scaled_int8_quant(xd)
            assert_bit_exact(q.cpu(), RI.arr(z, i, "q"), f"quant[{i}] q")
        assert np.array_equal(s.cpu().reshape(-1).numpy(), ref_s.reshape(-1).numpy()), f"quant[{i}] scales"
