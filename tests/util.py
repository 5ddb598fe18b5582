"""Shared helpers for the parity tests (oracle = oracle/ref_ops.py, CPU)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def dev():
    return torch.device("cuda:0")


def bits(t: torch.Tensor) -> torch.Tensor:
    """Reinterpret any tensor as raw integers for bit-exact comparison."""
    t = t.detach().cpu().contiguous()
    view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()]
    return t.view(view)


def assert_bit_exact(a: torch.Tensor, b: torch.Tensor, what=""):
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ba, bb = bits(a), bits(b)
    if not torch.equal(ba, bb):
        n = (ba != bb).sum().item()
        raise AssertionError(f"{what}: {n}/{ba.numel()} elements differ bitwise")


def ulp_bf16(x: torch.Tensor) -> torch.Tensor:
    """Size of one unit in the last place of the given dtype at |x|."""
    return torch.clamp(x.float().abs(), min=1e-30) * 2.0 ** -7


def assert_close_rel(got: torch.Tensor, ref: torch.Tensor, rel: float, what="", abs_floor=0.0):
    """max |got - ref| <= rel * max|ref| + abs_floor  (norm-wise relative bound)."""
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    assert g.shape == r.shape, f"{what}: shape {g.shape} vs {r.shape}"
    assert torch.isfinite(g).all(), f"{what}: non-finite values in result"
    err = (g - r).abs().max().item()
    bound = rel * r.abs().max().item() + abs_floor
    assert err <= bound, f"{what}: max err {err:.3e} > bound {bound:.3e}"


def assert_mostly_exact(got, ref, max_ulp: int, max_frac: float, what=""):
    """Rounded-type outputs: every element within `max_ulp` units in the last place of the
    reference and at most `max_frac` of the elements different at all."""
    g, r = got.detach().cpu(), ref.detach().cpu()
    assert g.shape == r.shape and g.dtype == r.dtype, f"{what}: {g.shape}{g.dtype} vs {r.shape}{r.dtype}"
    if g.dtype == torch.float8_e4m3fn:
        gi, ri = g.view(torch.uint8).int(), r.view(torch.uint8).int()
    elif g.element_size() == 2:
        gi, ri = g.view(torch.int16).int(), r.view(torch.int16).int()
    else:
        gi, ri = g.view(torch.int32).long(), r.view(torch.int32).long()
    # sign-magnitude -> monotonic integer line
    def mono(i, nbits):
        sign = 1 << (nbits - 1)
        neg = (i & sign) != 0
        mag = i & (sign - 1)
        return torch.where(neg, -mag, mag)
    nb = g.element_size() * 8
    d = (mono(gi, nb) - mono(ri, nb)).abs()
    frac = (d > 0).double().mean().item()
    assert d.max().item() <= max_ulp, f"{what}: max ulp distance {d.max().item()} > {max_ulp}"
    assert frac <= max_frac, f"{what}: {frac:.2e} of elements differ (> {max_frac:.1e})"


def make_kv_cache_x(num_blocks, block_size, num_kv_heads, head_size, dtype, seed=0):
    """Random paged KV cache in the x-split layout, U(-scale, scale), like
    vllm.utils.create_kv_caches_with_random (scale = head_size ** -0.5)."""
    g = torch.Generator().manual_seed(seed)
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    scale = head_size ** -0.5
    kc = (torch.rand(num_blocks, num_kv_heads, head_size // x, block_size, x, generator=g) * 2 - 1) * scale
    vc = (torch.rand(num_blocks, num_kv_heads, head_size, block_size, generator=g) * 2 - 1) * scale
    return kc.to(dtype), vc.to(dtype)


def assert_gemm_close(got, ref, what="", rel=2e-4, max_frac=0.05):
    """GEMM-type outputs rounded once to a 16-bit type.  The reference value and ours are
    roundings of two fp32 sums that differ only in accumulation order, so:
      * every element is within ONE ulp of the output type (a rounding flip) plus
        rel * max|ref| (accumulation-order noise; matters only where the sum cancels), which
        also implies the north_star bound max|err| <= 1e-3 * max|ref| before output rounding;
      * at most `max_frac` of the elements differ at all."""
    g, r = got.detach().cpu(), ref.detach().cpu()
    assert g.shape == r.shape and g.dtype == r.dtype, f"{what}: {g.shape}{g.dtype} vs {r.shape}{r.dtype}"
    gf, rf = g.double(), r.double()
    assert torch.isfinite(gf).all(), f"{what}: non-finite values"
    eps = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2.0 ** -23}[g.dtype]
    mag = torch.maximum(gf.abs(), rf.abs())
    ulp = torch.pow(2.0, torch.floor(torch.log2(torch.clamp(mag, min=1e-30)))) * eps
    bound = ulp + rel * rf.abs().max()
    err = (gf - rf).abs()
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())} elements beyond 1 ulp + {rel:g}*max|ref|; "
                           f"worst err {err.max().item():.3e} (max|ref| {rf.abs().max().item():.3e})")
    frac = (g.view(torch.int16 if g.element_size() == 2 else torch.int32)
            != r.view(torch.int16 if r.element_size() == 2 else torch.int32)).double().mean().item()
    assert frac <= max_frac, f"{what}: {frac:.3%} of elements differ (> {max_frac:.1%})"
