"""GPU: the vLLM-independent core of the plugin's AWQ / GPTQ linear methods (quant_config/linear.py, the
code `torch.ops.vllm._apply_awq` / `_apply_gptq` run; ref vllm_metax/quant_config/awq.py:69-80,118-159 and
gptq.py:49-75,180-229) against the CPU oracle, including the optional load-time prefill image."""
import numpy as np
import pytest
import torch

from tests.test_gpu_w4a16 import make_awq, make_gptq
from tests.util import assert_bit_exact, assert_gemm_close, dev

pytestmark = pytest.mark.gpu

from oracle import ref_ops as R  # noqa: E402


def lin():
    from vllm_metax_amd.quant_config import linear
    return linear


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 5, 256), (70, 256), (1, 256)])
def test_apply_awq(dtype, shape):
    k, n, g = 256, 384, 128
    qw, qz, sc, _, _ = make_awq(k, n, g, dtype, seed=5)
    x = (torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 0.5).to(dtype)
    bias = (torch.randn(n, generator=torch.Generator().manual_seed(2)) * 0.1).to(dtype)
    ref = R.awq_gemm(x.reshape(-1, k), R.awq_to_gptq_4bit(qw), sc, qz)
    ref_b = (ref.float() + bias.float()).to(dtype)
    d = dev()
    qd = lin().awq_process_weights(qw.to(d), g)
    out = lin().apply_awq(x.to(d), qd, sc.to(d), qz.to(d), None, 8, g)
    assert out.shape == x.shape[:-1] + (n,)
    assert_gemm_close(out.reshape(-1, n), ref, "apply_awq")
    out_b = lin().apply_awq(x.to(d), qd, sc.to(d), qz.to(d), bias.to(d), 8, g)
    assert_gemm_close(out_b.reshape(-1, n), ref_b, "apply_awq + bias", rel=4e-3)
    fake = lin().apply_awq_fake(x, qd, sc, qz, None, 8, g)
    assert fake.shape == out.shape and fake.dtype == out.dtype


def test_apply_awq_odd_group_uses_dequantize_matmul():
    """group_size % 32 != 0: the reference keeps the AWQ layout and serves the layer with awq_dequantize +
    matmul (awq.py:136-139); same here."""
    k, n, g = 240, 64, 48
    qw, qz, sc, _, _ = make_awq(k, n, g, torch.float16, seed=6)
    x = (torch.randn(9, k, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.float16)
    d = dev()
    qd = lin().awq_process_weights(qw.to(d), g)
    assert_bit_exact(qd, qw, "layout kept")
    out = lin().apply_awq(x.to(d), qd, sc.to(d), qz.to(d), None, 8, g)
    ref = (x.double() @ R.awq_dequantize(qw, sc, qz).double()).to(torch.float16)
    assert_gemm_close(out, ref, "dequantize + matmul", rel=2e-3, max_frac=0.2)


@pytest.mark.parametrize("desc_act", [False, True])
def test_apply_gptq(desc_act):
    dtype = torch.bfloat16
    k, n, g = 512, 256, 128
    qw, qz, sc = make_gptq(k, n, g, dtype, seed=7)
    rng = np.random.default_rng(3)
    g_idx = torch.from_numpy(np.repeat(np.arange(k // g), g)[rng.permutation(k)] if desc_act
                             else np.repeat(np.arange(k // g), g)).to(torch.int32)
    x = (torch.randn(33, k, generator=torch.Generator().manual_seed(4)) * 0.5).to(dtype)
    perm = torch.argsort(g_idx).to(torch.int32) if desc_act else None
    ref = R.gptq_gemm(x, R.gptq_shuffle(qw, perm), qz, sc, perm, g)
    d = dev()
    qd = qw.to(d)
    gi = lin().gptq_process_weights(qd, g_idx.to(d), desc_act, 4)
    out = lin().apply_gptq(x.to(d), qd, sc.to(d), qz.to(d), None, gi, True, 4, g, desc_act)
    assert_gemm_close(out, ref, f"apply_gptq desc_act={desc_act}")


def test_prefill_image_hook(monkeypatch):
    """MI355X_PREPACK_WEIGHTS=1: process_weights_after_loading attaches the weights' operand image to the LAYER
    (never a table keyed by a device address); a prefill-sized apply then runs on it and returns the bits of
    the per-call path; the image goes away with the layer."""
    monkeypatch.setenv("MI355X_PREPACK_WEIGHTS", "1")
    dtype = torch.bfloat16
    k, n, g = 256, 512, 128
    qw, qz, sc, _, _ = make_awq(k, n, g, dtype, seed=8)
    d = dev()

    class FakeLayer:
        pass
    layer = FakeLayer()
    layer.qweight, layer.qzeros, layer.scales = lin().awq_process_weights(qw.to(d), g), qz.to(d), sc.to(d)
    x = (torch.randn(1100, k, generator=torch.Generator().manual_seed(5)) * 0.5).to(dtype).to(d)
    plain = lin().apply_awq(x, layer.qweight, layer.scales, layer.qzeros, None, 8, g)
    assert lin().layer_image(layer, x) is None
    lin().attach_prefill_image(layer, False)
    img = lin().layer_image(layer, x)
    assert img is not None and img[1:] == (n, k)
    assert lin().layer_image(layer, x[:64]) is None            # decode keeps streaming the int4 words
    fast = lin().apply_awq(x, layer.qweight, layer.scales, layer.qzeros, None, 8, g, image=img)
    assert_bit_exact(fast, plain, "prefill image == per-call path")
    fast2 = lin().apply_w4a16_image(x.view(2, 550, k), img[0], img[1], img[2], None)
    assert_bit_exact(fast2.reshape(-1, n), plain, "apply_w4a16_image == per-call path")
    small = lin().apply_awq(x[:64], layer.qweight, layer.scales, layer.qzeros, None, 8, g, image=img)
    assert_bit_exact(small, lin().apply_awq(x[:64].clone(), layer.qweight, layer.scales, layer.qzeros, None, 8, g),
                     "decode unchanged")
    assert not hasattr(lin(), "_PREPACKED")
