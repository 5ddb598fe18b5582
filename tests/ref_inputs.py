"""Input builders shared by tests/golden/make_ref_fixtures.py (which feeds them to the reference's
own torch test references, in the build container) and by the tests that replay the stored
reference outputs against the oracle (CPU) and the HIP kernels (GPU).  Inputs are pure integer
formulas (tests/det_inputs.py); every fixture case stores the CRC32 of its large inputs, which
`check_crc` verifies on replay, so both sides are known to have seen the same bytes.
"""
import json

import numpy as np
import torch

from tests import det_inputs as D
from tests import golden_io as G

BF, F16, F32 = torch.bfloat16, torch.float16, torch.float32
FP8 = torch.float8_e4m3fn
DT = {"bfloat16": BF, "float16": F16, "float32": F32}


def load(name):
    z = G.load(name)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def arr(z, i, key, dtype=None):
    """Array `key` of case i as a torch tensor (16-bit floats / fp8 are stored as raw bits)."""
    a = z[f"c{i}_{key}"]
    if dtype in (BF, F16):
        return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    if dtype == FP8:
        return torch.from_numpy(a.copy()).view(FP8)
    return torch.from_numpy(a.copy())


def check_crc(meta, **tensors):
    for k, t in tensors.items():
        assert D.crc(t) == meta[f"{k}_crc"], f"input {k} was not rebuilt bit-identically"


# ------------------------------------------------------------------ decode attention
def decode_inputs(m):
    """(q, key_cache [nb,kvh,d/x,bs,x], value_cache [nb,kvh,d,bs], block_tables, seq_lens, slopes)"""
    dt = DT[m["dtype"]]
    H, KVH, d, bs, nb, seed = m["H"], m["KVH"], m["d"], m["bs"], m["nb"], m["seed"]
    S = len(m["seq_lens"])
    x = 16 // torch.tensor([], dtype=dt).element_size()
    s = float(1.0 / (d ** 0.5))
    if m["regime"] == "ref":       # the reference's distribution: test_attention.py:166-167, kv_cache_factory
        q = D.uniform((S, H, d), seed + 2, -s, s)
        kc = D.uniform((nb, KVH, d // x, bs, x), seed, -s, s)
        vc = D.uniform((nb, KVH, d, bs), seed + 1, -s, s)
    else:                          # unit-variance q / k: logits of order 1
        q = D.normalish((S, H, d), seed + 2, 1.0)
        kc = D.normalish((nb, KVH, d // x, bs, x), seed, 1.0)
        vc = D.normalish((nb, KVH, d, bs), seed + 1, 1.0)
    mb = (max(m["seq_lens"]) + bs - 1) // bs
    bt = D.randint((S, mb), seed + 3, 0, nb).to(torch.int32)
    sl = torch.tensor(m["seq_lens"], dtype=torch.int32)
    slopes = D.normalish((H,), seed + 4, 1.0) if m["alibi"] else None
    return q.to(dt), kc.to(dt), vc.to(dt), bt, sl, slopes


def decode_inputs_fp8(m):
    """fp8 (e4m3fn) KV cache in the x = 16 layout: (q, key_cache u8 [nb,kvh,d/16,bs,16],
    value_cache u8 [nb,kvh,d,bs], block_tables, seq_lens, slopes).  Bytes = e4m3 of U(-1, 1)."""
    dt = DT[m["dtype"]]
    H, KVH, d, bs, nb, seed = m["H"], m["KVH"], m["d"], m["bs"], m["nb"], m["seed"]
    S = len(m["seq_lens"])
    q = D.normalish((S, H, d), seed + 2, 1.0).to(dt)
    kc = D.uniform((nb, KVH, d // 16, bs, 16), seed, -1.0, 1.0).to(FP8).view(torch.uint8)
    vc = D.uniform((nb, KVH, d, bs), seed + 1, -1.0, 1.0).to(FP8).view(torch.uint8)
    mb = (max(m["seq_lens"]) + bs - 1) // bs
    bt = D.randint((S, mb), seed + 3, 0, nb).to(torch.int32)
    sl = torch.tensor(m["seq_lens"], dtype=torch.int32)
    slopes = D.normalish((H,), seed + 4, 1.0) if m["alibi"] else None
    return q, kc, vc, bt, sl, slopes


def fp8_cache_to(cache_u8, scale, dtype):
    """e4m3 bytes -> dtype values float(byte) * scale (exact for scale 1: e4m3 is a subset)."""
    return (cache_u8.view(FP8).float() * scale).to(dtype)


# ------------------------------------------------------------------ prefill attention
def prefill_inputs(m):
    """(q [T,H,d], key_cache NHD [nb,bs,kvh,d], value_cache NHD, block_tables) — all bf16."""
    H, KVH, d, bs, nb, seed = m["H"], m["KVH"], m["d"], m["bs"], m["nb"], m["seed"]
    q = D.normalish((sum(m["query_lens"]), H, d), seed, 1.0).to(BF)
    kc = D.normalish((nb, bs, KVH, d), seed + 1, 1.0).to(BF)
    vc = D.normalish((nb, bs, KVH, d), seed + 2, 1.0).to(BF)
    mb = (max(m["kv_lens"]) + bs - 1) // bs
    bt = D.randint((len(m["query_lens"]), mb), seed + 3, 0, nb).to(torch.int32)
    return q, kc, vc, bt


def nhd_to_xsplit(kc_nhd, vc_nhd):
    """NHD paged cache -> the x-split layout of paged_attention (pure permutation)."""
    nb, bs, kvh, d = kc_nhd.shape
    x = 16 // kc_nhd.element_size()
    kc = kc_nhd.reshape(nb, bs, kvh, d // x, x).permute(0, 2, 3, 1, 4).contiguous()
    vc = vc_nhd.permute(0, 2, 3, 1).contiguous()
    return kc, vc


# ------------------------------------------------------------------ AWQ
def awq_inputs(m):
    dt = DT[m["dtype"]]
    seed = m["seed"]
    if m["kind"] == "dequantize":
        rows, cols8, g = m["rows"], m["cols8"], m["group"]
        qw = D.int32_words((rows, cols8), seed)
        qz = D.int32_words((rows // g, cols8), seed + 50)
        sc = D.uniform((rows // g, cols8 * 8), seed + 100, 0.0, 1.0).to(dt)
        return qw, qz, sc, None
    M, K, N, g = m["M"], m["K"], m["N"], m["group"]
    qw = D.int32_words((K, N // 8), seed)
    qz = D.int32_words((K // g, N // 8), seed + 50)
    sc = D.uniform((K // g, N), seed + 100, 1e-3, 1e-2).to(dt)
    x = D.normalish((M, K), seed + 150, 1.0).to(dt)
    return qw, qz, sc, x


# ------------------------------------------------------------------ merge_attn_states
def merge_inputs(m):
    n, h, d, seed = m["n"], m["h"], m["d"], m["seed"]
    dt = DT[m["dtype"]]
    p_out = D.normalish((n, h, d), seed, 1.0).to(dt)
    s_out = D.normalish((n, h, d), seed + 20, 1.0).to(dt)
    p_lse = D.normalish((h, n), seed + 40, 3.0)
    s_lse = D.normalish((h, n), seed + 60, 3.0)
    mp = D.uniform((h, n), seed + 80, 0, 1) < 0.1            # +inf in one of the two (never both), as
    ms = (D.uniform((h, n), seed + 100, 0, 1) < 0.1) & ~mp   # the reference's test :102-121
    p_lse[mp] = float("inf")
    s_lse[ms] = float("inf")
    return p_out, p_lse, s_out, s_lse


# ------------------------------------------------------------------ scaled_mm
def scaled_mm_operands(m):
    """a [m,k], b^T [n,k] quantised as the reference's to_fp8 / to_int8 helpers do (round to an
    integer, clamp to the type's range, cast: tests/kernels/utils.py:1221-1228 — the generator runs
    the reference's helpers and stores the CRC32 of what they returned), scales and bias."""
    af, btf, a_s, b_s, bias = scaled_mm_float_operands(m)
    if m["kind"] == "fp8":
        q = lambda t: torch.round(t.clamp(min=-448.0, max=448.0)).to(FP8)          # noqa: E731
    else:
        q = lambda t: torch.round(t.clamp(min=-128, max=127)).to(torch.int8)       # noqa: E731
    return q(af), q(btf), a_s, b_s, bias


def scaled_mm_float_operands(m):
    """float32 pre-images of a [m,k] and b^T [n,k], scales and bias."""
    seed = m["seed"]
    std = 1.0 if m["kind"] == "fp8" else 5.0
    a = D.normalish((m["m"], m["k"]), seed, std)
    bt = D.normalish((m["n"], m["k"]), seed + 1, std)
    a_s = D.uniform((m["m"] if m["per_token"] else 1, 1), seed + 2, 1e-3, 1e-2)
    b_s = D.uniform((1, m["n"] if m["per_channel"] else 1), seed + 3, 1e-3, 1e-2)
    bias = D.uniform((m["n"],), seed + 4, -1.0, 1.0).to(DT[m["out_dtype"]]) if m["bias"] else None
    return a, bt, a_s, b_s, bias


# ------------------------------------------------------------------ dynamic quant
def quant_input(m):
    return (D.uniform((m["T"], m["hidden"]), m["seed"], -300.0, 700.0) * 0.01).to(DT[m["dtype"]])
