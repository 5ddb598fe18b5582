"""int8 W8A8 path — HIP kernels vs the CPU oracle, through the C-ABI.
int32 accumulation is exact and the fp32 epilogue is one fixed sequence of IEEE operations, so the
bar is BIT-EXACT for the GEMM and for the quantisers.  Shapes follow the reference's own tests
(tests/kernels/quantization/test_cutlass_scaled_mm.py, test_int8_quant.py: per-tensor and
per-token / per-channel scales, bias, M from 1 to a few hundred).  Parity unpinned against the
reference binary (no known-answer vectors ship for these ops)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import assert_bit_exact, dev  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

pytestmark = pytest.mark.gpu


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("tokens,hidden", [(1, 64), (7, 4096), (33, 5120), (4, 1003)])
def test_scaled_int8_quant_dynamic_and_static(dtype, tokens, hidden):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(tokens, hidden, generator=g) * 3).to(dtype)
    x[0, :3] = torch.tensor([0.5, 1.5, 2.5]).to(dtype)        # exact ties after scaling by 1
    dv = dev()
    q_ref, s_ref = R.scaled_int8_quant(x)
    q, s, azp = ops().scaled_int8_quant(x.to(dv))
    assert azp is None
    assert torch.equal(s.cpu(), s_ref), "dynamic scales"
    assert torch.equal(q.cpu(), q_ref), "dynamic int8"
    scale = torch.tensor([0.037], dtype=torch.float32)
    q_ref, _ = R.scaled_int8_quant(x, scale)
    q, _, _ = ops().scaled_int8_quant(x.to(dv), scale.to(dv))
    assert torch.equal(q.cpu(), q_ref), "static int8"
    assert int(q.min()) >= -127                                 # this fork never emits -128


def test_scaled_int8_quant_zero_row_and_azp():
    dv = dev()
    x = torch.zeros(3, 128, dtype=torch.bfloat16, device=dv)
    x[1] = 2.0
    q, s, _ = ops().scaled_int8_quant(x)
    assert torch.equal(q[0].cpu(), torch.zeros(128, dtype=torch.int8)) and float(s[0]) == 0.0
    assert int(q[1, 0]) == 127
    with pytest.raises(NotImplementedError):
        ops().scaled_int8_quant(x, None, torch.zeros(3, dtype=torch.int32, device=dv), symmetric=False)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m", [1, 16, 33, 64, 100, 300, 1030])
@pytest.mark.parametrize("k,n", [(256, 128), (1024, 768), (4096, 1024), (320, 272)])
@pytest.mark.parametrize("per_token,per_channel,use_bias", [(False, False, False), (True, True, True),
                                                           (True, False, False), (False, True, True)])
def test_cutlass_scaled_mm_int8(out_dtype, m, k, n, per_token, per_channel, use_bias):
    g = torch.Generator().manual_seed(m * 7 + k)
    a = torch.randint(-127, 128, (m, k), generator=g, dtype=torch.int32).to(torch.int8)
    b_nk = torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int32).to(torch.int8)   # column-major [K, N]
    a_s = torch.rand(m if per_token else 1, 1, generator=g) * 1e-2 + 1e-3
    b_s = torch.rand(1, n if per_channel else 1, generator=g) * 1e-2 + 1e-3
    bias = (torch.randn(n, generator=g) * 0.5).to(out_dtype) if use_bias else None
    ref = R.scaled_mm_int8(a, b_nk.t(), a_s, b_s, out_dtype, bias)
    dv = dev()
    out = torch.full((m, n), float("nan"), dtype=out_dtype, device=dv)
    ops().cutlass_scaled_mm(out, a.to(dv), b_nk.to(dv).t(), a_s.to(dv), b_s.to(dv),
                            bias.to(dv) if use_bias else None)
    assert_bit_exact(out, ref, "cutlass_scaled_mm int8")


def test_int8_pipeline_and_torch_binding():
    """quantise activations per token -> int8 GEMM with per-channel weight scales, via torch.ops."""
    import vllm_metax_amd._C  # noqa: F401
    dv = dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(19, 512, generator=g).to(torch.bfloat16).to(dv)
    w = torch.randint(-127, 128, (256, 512), generator=g, dtype=torch.int32).to(torch.int8).to(dv)
    w_s = (torch.rand(1, 256, generator=g) * 1e-2 + 1e-3).to(dv)
    q = torch.empty(19, 512, dtype=torch.int8, device=dv)
    s = torch.empty(19, 1, dtype=torch.float32, device=dv)
    torch.ops._C.dynamic_scaled_int8_quant(q, x, s, None)
    q2, s2, _ = ops().scaled_int8_quant(x)
    assert torch.equal(q, q2) and torch.equal(s, s2)
    out1 = torch.empty(19, 256, dtype=torch.bfloat16, device=dv)
    out2 = torch.empty_like(out1)
    torch.ops._C.cutlass_scaled_mm(out1, q, w.t(), s, w_s, None)
    ops().cutlass_scaled_mm(out2, q, w.t(), s, w_s, None)
    assert_bit_exact(out1, out2, "binding")
    ref = R.scaled_mm_int8(q.cpu(), w.cpu().t(), s.cpu(), w_s.cpu(), torch.bfloat16)
    assert_bit_exact(out1, ref, "pipeline vs oracle")


def test_golden_int8_w8a8_gpu():
    from tests import golden_io as G
    z = G.load("int8_w8a8")
    dv = dev()
    x = G.bf16(z["x"]).to(dv)
    q, s, _ = ops().scaled_int8_quant(x)
    assert torch.equal(q.cpu(), torch.from_numpy(z["q_dynamic"].copy()))
    assert torch.equal(s.cpu(), G.f32(z["scales_dynamic"]))
    qs, _, _ = ops().scaled_int8_quant(x, G.f32(z["scale_static"]).to(dv))
    assert torch.equal(qs.cpu(), torch.from_numpy(z["q_static"].copy()))
    w = torch.from_numpy(z["w_nk"].copy()).to(dv)
    out = torch.empty(6, 64, dtype=torch.bfloat16, device=dv)
    ops().cutlass_scaled_mm(out, q, w.t(), s, G.f32(z["w_scales"]).to(dv), G.bf16(z["bias"]).to(dv))
    assert_bit_exact(out, G.bf16(z["out"]), "golden int8 GEMM")
