"""merge_attn_states — HIP kernel vs the CPU oracle, through the C-ABI.
Shapes / dtypes / the +inf convention follow the reference's own test
(tests/kernels/attention/test_merge_attn_states.py:48-51, :102-121).  Parity unpinned against the
reference binary (no known-answer vectors ship for this op): the oracle is the restated formula,
cross-checked against the reference's in-test torch implementation of the same formula."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import assert_close_rel, assert_mostly_exact, dev  # noqa: E402
from oracle import ref_ops as R  # noqa: E402

pytestmark = pytest.mark.gpu


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _case(n, h, d, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    p_out = torch.randn(n, h, d, generator=g).to(dtype)
    s_out = torch.randn(n, h, d, generator=g).to(dtype)
    p_lse = torch.randn(h, n, generator=g) * 3
    s_lse = torch.randn(h, n, generator=g) * 3
    # 10 % / 10 % of the entries are +inf in one of the two (never both), as in the reference's test
    mask_p = torch.rand(h, n, generator=g) < 0.1
    mask_s = (torch.rand(h, n, generator=g) < 0.1) & ~mask_p
    p_lse[mask_p] = float("inf")
    s_lse[mask_s] = float("inf")
    return p_out, p_lse, s_out, s_lse


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,h,d", [(256, 4, 32), (613, 8, 48), (1024, 32, 128), (37, 64, 256), (1, 16, 96)])
@pytest.mark.parametrize("with_lse", [True, False])
def test_merge_attn_states(dtype, n, h, d, with_lse):
    p_out, p_lse, s_out, s_lse = _case(n, h, d, dtype)
    ref_out, ref_lse = R.merge_attn_states(p_out, p_lse, s_out, s_lse)
    dv = dev()
    out = torch.full((n, h, d), float("nan")).to(dtype).to(dv)
    out_lse = torch.full((h, n), float("nan"), dtype=torch.float32, device=dv) if with_lse else None
    ops().merge_attn_states(out, p_out.to(dv), p_lse.to(dv), s_out.to(dv), s_lse.to(dv), out_lse)
    if dtype == torch.float32:
        assert_close_rel(out, ref_out, 2e-6, "output")
    else:
        # expf / division differ from torch-CPU by <= 1 fp32 ulp: a rounding flip of the 16-bit output
        assert_mostly_exact(out, ref_out, 1, 0.01, "output")
    if with_lse:
        assert torch.isfinite(out_lse).all()
        assert_close_rel(out_lse, ref_lse, 2e-6, "output_lse")


def test_merge_attn_states_checks():
    dv = dev()
    out = torch.empty(4, 2, 36, dtype=torch.bfloat16, device=dv)       # 36 % 8 != 0
    lse = torch.zeros(2, 4, device=dv)
    with pytest.raises(RuntimeError, match="pack_size"):
        ops().merge_attn_states(out, out.clone(), lse, out.clone(), lse)
    out = torch.empty(4, 2, 64, dtype=torch.bfloat16, device=dv)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops().merge_attn_states(out, out.clone().transpose(0, 1).contiguous().transpose(0, 1), lse,
                                out.clone(), lse)
    # empty batch is a no-op
    e = torch.empty(0, 2, 64, dtype=torch.bfloat16, device=dv)
    ops().merge_attn_states(e, e.clone(), torch.zeros(2, 0, device=dv), e.clone(), torch.zeros(2, 0, device=dv))


def test_golden_merge_attn_states_gpu():
    from tests import golden_io as G
    z = G.load("merge_attn_states")
    dv = dev()
    po, so = G.bf16(z["prefix_output"]).to(dv), G.bf16(z["suffix_output"]).to(dv)
    out = torch.empty_like(po)
    lse = torch.empty(po.shape[1], po.shape[0], device=dv)
    ops().merge_attn_states(out, po, G.f32(z["prefix_lse"]).to(dv), so, G.f32(z["suffix_lse"]).to(dv), lse)
    assert_mostly_exact(out, G.bf16(z["output"]), 1, 0.01, "golden output")
    assert_close_rel(lse, G.f32(z["output_lse"]), 2e-6, "golden lse")
