"""GPU: the vLLM-independent core of register_patch()'s fused decoder-layer forwards (patch/fused_layers.py)
against the op sequence upstream vLLM runs on the plain op surface — BIT-IDENTICAL, at decode (M <= 64), mid
(64 < M < 1024: falls through to the plain ops) and prefill sizes (M >= 1024, with and without load-time weight
images), AWQ and GPTQ.  ref: the reference wires its model-level changes the same way
(vllm_metax/__init__.py:92-93, vllm_metax/patch/**)."""
import pytest
import torch

from tests.test_gpu_w4a16 import make_awq, make_gptq
from tests.util import assert_bit_exact, dev

pytestmark = pytest.mark.gpu


def F():
    from vllm_metax_amd.patch import fused_layers
    return fused_layers


def ops():
    from vllm_metax_amd import _custom_ops
    return _custom_ops


def _layer(kind, k, n, g, dtype, seed, image):
    from vllm_metax_amd.quant_config import linear
    d = dev()
    if kind == "awq":
        qw, qz, sc, _, _ = make_awq(k, n, g, dtype, seed=seed)
        qwd = linear.awq_process_weights(qw.to(d), g)
        g_idx = None
    else:
        qw, qz, sc = make_gptq(k, n, g, dtype, seed=seed)
        qwd = qw.to(d)
        g_idx = linear.gptq_process_weights(qwd, torch.arange(k, dtype=torch.int32, device=d) // g, False, 4)
    qzd, scd = qz.to(d), sc.to(d)
    img = None
    if image:
        img = (ops().w4a16_prepack(qwd, qzd, scd, kind == "gptq"), n, k)
    return F().W4Linear(kind, qwd, qzd, scd, g, g_idx, img, None)


def _plain_norm_mlp(x, residual, w, eps, gu, dn):
    """What vLLM runs unpatched: fused_add_rms_norm -> gate_up -> silu_and_mul -> down."""
    x, residual = x.clone(), residual.clone()
    ops().fused_add_rms_norm(x, residual, w, eps)
    h = gu.plain(x)
    act = torch.empty(h.shape[0], h.shape[1] // 2, dtype=h.dtype, device=h.device)
    ops().silu_and_mul(act, h)
    return dn.plain(act), residual


@pytest.mark.parametrize("kind,image", [("awq", False), ("awq", True), ("gptq", True), ("gptq", False)])
@pytest.mark.parametrize("m", [7, 64, 300, 1100])
def test_fused_norm_mlp_is_bit_identical_to_the_plain_ops(kind, image, m):
    dtype = torch.bfloat16
    hidden, ffn, g = 512, 1024, 128
    gu = _layer(kind, hidden, 2 * ffn, g, dtype, 11, image)
    dn = _layer(kind, ffn, hidden, g, dtype, 12, image)
    assert F().fusable(gu, dn)
    d = dev()
    gen = torch.Generator().manual_seed(m)
    x = (torch.randn(m, hidden, generator=gen) * 0.5).to(dtype).to(d)
    res = (torch.randn(m, hidden, generator=gen) * 0.5).to(dtype).to(d)
    w = (torch.rand(hidden, generator=gen) * 0.2 + 0.9).to(dtype).to(d)
    ref_out, ref_res = _plain_norm_mlp(x, res, w, 1e-5, gu, dn)
    xf, rf = x.clone(), res.clone()
    out, res_out = F().fused_norm_mlp(xf, rf, w, 1e-5, gu, dn)
    assert_bit_exact(out, ref_out, f"fused_norm_mlp {kind} image={image} m={m}")
    assert_bit_exact(res_out, ref_res, "residual")
    # the MLP alone (patched LlamaMLP.forward)
    h = x.clone()
    ops().fused_add_rms_norm(h, res.clone(), w, 1e-5)
    assert_bit_exact(F().fused_mlp(h, gu, dn), ref_out, "fused_mlp")


@pytest.mark.parametrize("m", [33, 1100])
@pytest.mark.parametrize("first", [True, False])
def test_fused_norm_linear_is_bit_identical(m, first):
    dtype = torch.bfloat16
    hidden, n, g = 512, 768, 128
    lin = _layer("awq", hidden, n, g, dtype, 13, True)
    d = dev()
    gen = torch.Generator().manual_seed(m + 1)
    x = (torch.randn(m, hidden, generator=gen) * 0.5).to(dtype).to(d)
    res = None if first else (torch.randn(m, hidden, generator=gen) * 0.5).to(dtype).to(d)
    w = (torch.rand(hidden, generator=gen) * 0.2 + 0.9).to(dtype).to(d)
    if first:
        h = torch.empty_like(x)
        ops().rms_norm(h, x, w, 1e-5)
        ref, ref_res = lin.plain(h), x
    else:
        h, ref_res = x.clone(), res.clone()
        ops().fused_add_rms_norm(h, ref_res, w, 1e-5)
        ref = lin.plain(h)
    out, res_out = F().fused_norm_linear(x.clone(), None if first else res.clone(), w, 1e-5, lin)
    assert_bit_exact(out, ref, "fused_norm_linear")
    assert_bit_exact(res_out, ref_res, "residual")


def test_unfusable_layers_are_reported():
    dtype = torch.bfloat16
    gu = _layer("awq", 256, 512, 128, dtype, 1, False)
    dn = _layer("awq", 256, 256, 128, dtype, 2, False)
    assert F().fusable(gu, dn)
    dn.bias = torch.zeros(256, dtype=dtype, device=dev())
    assert not F().fusable(gu, dn)
    assert not F().fusable(gu, None)


def test_register_patch_without_vllm_is_a_no_op():
    """In a vLLM-less environment the patch targets do not import: nothing is patched, nothing raises."""
    import vllm_metax_amd
    try:
        import vllm  # noqa: F401
        pytest.skip("vLLM is installed here")
    except ImportError:
        pass
    assert vllm_metax_amd.register_patch() == []


def test_plugin_surface_model_patched_equals_plain():
    """harness.PluginSurfaceModel(patched=True) — the decoder-layer forwards register_patch() installs — serves the
    same tokens and bit-identical logits as the plain op surface (prefill chunk with images + graph-replayed decode)."""
    from vllm_metax_amd import harness
    torch.manual_seed(0)
    cfg = harness.ModelConfig.llama_geometry("awq", layers=2, vocab=4096)
    n, plen, steps = 2, 1024, 3
    tokens = torch.randint(0, cfg.vocab, (n, plen), device="cuda:0")
    res = []
    for patched in (False, True):
        m = harness.PluginSurfaceModel(cfg, n, plen + 16, device="cuda:0", seed=0, prepack_weights=True, patched=patched)
        m.setup_decode(n, plen, plen + 16)
        first = m.prefill(tokens, list(range(n)), 0)
        logits = [m.last_prefill_logits.float().cpu()]
        m.d_tokens.copy_(first)
        m.set_decode_lengths(torch.full((n,), plen, device=m.device))
        toks = [first.cpu()]
        for _ in range(steps):
            m.decode_step(use_graph=True)
            torch.cuda.synchronize()
            toks.append(m.d_tokens.cpu().clone())
            logits.append(m.last_logits.float().cpu().clone())
        res.append((torch.stack(toks), logits))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
