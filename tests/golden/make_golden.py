"""Generate the golden input/output fixtures under tests/golden/ from the CPU oracle.

The reference holds NO stored known-answer vectors for the hot-path kernels and neither its
CUDA-dialect csrc nor its vLLM-importing Python can run here, so these
fixtures are the oracle's own outputs (oracle/ref_ops.py, a restatement of the cited reference
sources), frozen so that (a) the oracle cannot drift silently, (b) the independently written C
port (oracle/cpu_port.c) and (c) the HIP kernels are all checked against the same bytes.
Status: "parity unpinned" against the reference binary; pinned against its source semantics.

Run:  python tests/golden/make_golden.py      (deterministic: fixed seeds, CPU only)
16-bit float tensors are stored as their raw uint16 / fp8 as uint8 bit patterns.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import ref_ops as R  # noqa: E402

OUT = Path(__file__).resolve().parent


def raw(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float8_e4m3fn:
        return t.view(torch.uint8).numpy()
    return t.numpy()


ONLY = set(sys.argv[1:])   # optional: regenerate just the named fixtures


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    np.savez_compressed(OUT / f"{name}.npz", **{k: (raw(v) if isinstance(v, torch.Tensor) else v)
                                                for k, v in arrays.items()})
    print(f"{name}.npz", {k: tuple(np.shape(v)) for k, v in arrays.items()})


def kv_cache(nb, bs, kvh, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    s = d ** -0.5
    kc = ((torch.rand(nb, kvh, d // 8, bs, 8, generator=g) * 2 - 1) * s).to(dtype)
    vc = ((torch.rand(nb, kvh, d, bs, generator=g) * 2 - 1) * s).to(dtype)
    return kc, vc


def main():
    bf = torch.bfloat16
    # ---- cache ops -----------------------------------------------------------------
    g = torch.Generator().manual_seed(1)
    T, H, D, bs, nb = 19, 4, 64, 16, 8
    key = torch.randn(T, H, D, generator=g).to(bf)
    value = torch.randn(T, H, D, generator=g).to(bf)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    slots[3] = -1
    kc = torch.zeros(nb, H, D // 8, bs, 8, dtype=bf)
    vc = torch.zeros(nb, H, D, bs, dtype=bf)
    R.reshape_and_cache(key, value, kc, vc, slots)
    save("reshape_and_cache", key=key, value=value, slots=slots.numpy(), key_cache=kc, value_cache=vc)

    # ---- paged attention v1 / v2 -----------------------------------------------------
    S, H, KVH, D, bs = 3, 4, 1, 128, 16
    lens = [17, 513, 530]                      # block tail, partition boundary, 2 partitions
    nblk = 36                                  # the three sequences alias one small block pool
    kc, vc = kv_cache(nblk, bs, KVH, D, bf, 2)
    g = torch.Generator().manual_seed(2)
    q = ((torch.rand(S, H, D, generator=g) * 2 - 1) * D ** -0.5).to(bf)
    bt = torch.stack([torch.randperm(nblk, generator=g)[:34] for _ in range(S)]).to(torch.int32)
    sl = torch.tensor(lens, dtype=torch.int32)
    o1 = R.paged_attention_v1(q, kc, vc, KVH, D ** -0.5, bt, sl)
    o2, es, ml, tmp = R.paged_attention_v2(q, kc, vc, KVH, D ** -0.5, bt, sl, 530)
    save("paged_attention", q=q, key_cache=kc, value_cache=vc, block_tables=bt.numpy(),
         seq_lens=sl.numpy(), out_v1=o1, out_v2=o2, exp_sums=es.numpy(), max_logits=ml.numpy(), tmp_out=tmp)

    # ---- prefill -----------------------------------------------------------------------
    q_lens, seq_lens = [40, 7], [100, 7]
    cu = torch.tensor([0, 40, 47], dtype=torch.int32)
    g = torch.Generator().manual_seed(3)
    qp = (torch.randn(47, H, D, generator=g) * 0.5).to(bf)
    slp = torch.tensor(seq_lens, dtype=torch.int32)
    op = R.paged_prefill_attention(qp, kc, vc, KVH, D ** -0.5, bt[:2], slp, cu)
    save("paged_prefill", q=qp, block_tables=bt[:2].numpy(), seq_lens=slp.numpy(), cu_seqlens_q=cu.numpy(), out=op)

    # ---- layernorm family / fp8 quant ---------------------------------------------------
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 1024, generator=g).to(bf)
    res = torch.randn(5, 1024, generator=g).to(bf)
    w = (torch.randn(1024, generator=g) * 0.1 + 1).to(bf)
    scale = torch.tensor([0.05], dtype=torch.float32)
    n1 = R.rms_norm(x, w, 1e-5)
    n2, r2 = R.fused_add_rms_norm(x, res, w, 1e-5)
    q1 = R.rms_norm_static_fp8_quant(x, w, scale, 1e-5)
    q2, r3 = R.fused_add_rms_norm_static_fp8_quant(x, res, w, scale, 1e-5)
    q3, s3, r4 = R.rms_norm_dynamic_per_token_quant(x, w, 1e-5, None, res)
    sq = R.static_scaled_fp8_quant(x, scale)
    dq, ds = R.dynamic_scaled_fp8_quant(x)
    pq, ps = R.dynamic_per_token_scaled_fp8_quant(x)
    save("layernorm_quant", x=x, residual=res, weight=w, scale=scale.numpy(), rms_norm=n1,
         fused_out=n2, fused_residual=r2, static_q=q1, fused_static_q=q2, dyn_q=q3,
         dyn_scales=s3.numpy(), dyn_residual=r4, fp8_static=sq, fp8_dynamic=dq,
         fp8_dynamic_scale=ds.numpy(), fp8_token=pq, fp8_token_scales=ps.numpy())

    # ---- rotary / silu -------------------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    hs, rot, Hq, Hk, Tn = 128, 128, 4, 2, 9
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    fr = torch.outer(torch.arange(64).float(), inv)
    cache = torch.cat([fr.cos(), fr.sin()], -1).to(bf)
    pos = torch.randint(0, 64, (Tn,), generator=g, dtype=torch.int64)
    qr = torch.randn(Tn, Hq * hs, generator=g).to(bf)
    kr = torch.randn(Tn, Hk * hs, generator=g).to(bf)
    qn, kn = R.rotary_embedding(pos, qr, kr, hs, cache, True)
    qj, kj = R.rotary_embedding(pos, qr, kr, hs, cache, False)
    xs = torch.randn(6, 2 * 512, generator=g).to(bf)
    save("rotary_silu", positions=pos.numpy(), q=qr, k=kr, cache=cache, q_neox=qn, k_neox=kn,
         q_gptj=qj, k_gptj=kj, silu_in=xs, silu_out=R.silu_and_mul(xs))

    # ---- int4 weight-only -------------------------------------------------------------------
    rng = np.random.default_rng(6)
    k, n, grp, m = 512, 128, 128, 11
    wq = rng.integers(0, 16, size=(k, n), dtype=np.uint8)
    zq = rng.integers(0, 16, size=(k // grp, n), dtype=np.uint8)
    qw, qz = R.awq_pack(wq), R.awq_pack(zq)
    g = torch.Generator().manual_seed(6)
    sc = (torch.rand(k // grp, n, generator=g) * 9e-3 + 1e-3).to(bf)
    xg = torch.randn(m, k, generator=g).to(bf)
    q2 = R.awq_to_gptq_4bit(qw)
    gq = R.gptq_pack_rows(wq)
    gz = torch.from_numpy(R._pack_nibbles(np.minimum(zq, 14).reshape(k // grp, n // 8, 8)).view(np.int32))
    perm = torch.randperm(k, generator=g).to(torch.int32)
    gs = R.gptq_shuffle(gq, None)
    gsp = R.gptq_shuffle(gq, perm)
    save("w4a16", awq_qweight=qw.numpy(), awq_qzeros=qz.numpy(), scales=sc, x=xg,
         awq_repacked=q2.numpy(), awq_dequant=R.awq_dequantize(qw, sc, qz),
         awq_gemm=R.awq_gemm(xg, q2, sc, qz), gptq_qweight=gq.numpy(), gptq_qzeros=gz.numpy(),
         gptq_shuffled=gs.numpy(), perm=perm.numpy(), gptq_shuffled_perm=gsp.numpy(),
         gptq_gemm=R.gptq_gemm(xg, gs, gz, sc, None, grp),
         gptq_gemm_perm=R.gptq_gemm(xg, gsp, gz, sc, perm, grp))

    # ---- fp8 GEMM ------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    a = (torch.randn(9, 256, generator=g) * 2).to(torch.float8_e4m3fn)
    b = (torch.randn(64, 256, generator=g) * 2).to(torch.float8_e4m3fn)   # [N, K] = column-major [K, N]
    a_s = torch.rand(9, 1, generator=g) * 9e-3 + 1e-3
    b_s = torch.rand(1, 64, generator=g) * 9e-3 + 1e-3
    save("scaled_mm_fp8", a=a, b_nk=b, a_scales=a_s.numpy(), b_scales=b_s.numpy(),
         out=R.scaled_mm_fp8(a, b.t(), a_s, b_s, bf))

    # ---- merge_attn_states ------------------------------------------------------
    g = torch.Generator().manual_seed(8)
    n, h, d = 13, 4, 64
    p_out, s_out = torch.randn(n, h, d, generator=g).to(bf), torch.randn(n, h, d, generator=g).to(bf)
    p_lse, s_lse = torch.randn(h, n, generator=g) * 3, torch.randn(h, n, generator=g) * 3
    p_lse[0, 1] = float("inf")
    s_lse[2, 7] = float("inf")
    m_out, m_lse = R.merge_attn_states(p_out, p_lse, s_out, s_lse)
    save("merge_attn_states", prefix_output=p_out, prefix_lse=p_lse.numpy(), suffix_output=s_out,
         suffix_lse=s_lse.numpy(), output=m_out, output_lse=m_lse.numpy())

    # ---- int8 W8A8 ---------------------------------------------------------------
    g = torch.Generator().manual_seed(9)
    x8 = (torch.randn(6, 256, generator=g) * 2).to(bf)
    x8[0, :4] = torch.tensor([0.5, 1.5, -2.5, 0.0]).to(bf)
    q_dyn, s_dyn = R.scaled_int8_quant(x8)
    s_static = torch.tensor([0.043], dtype=torch.float32)
    q_st, _ = R.scaled_int8_quant(x8, s_static)
    w8 = torch.randint(-127, 128, (64, 256), generator=g, dtype=torch.int32).to(torch.int8)   # [N, K]
    w_s = torch.rand(1, 64, generator=g) * 9e-3 + 1e-3
    bias8 = (torch.randn(64, generator=g) * 0.5).to(bf)
    save("int8_w8a8", x=x8, q_dynamic=q_dyn.numpy(), scales_dynamic=s_dyn.numpy(), scale_static=s_static.numpy(),
         q_static=q_st.numpy(), w_nk=w8.numpy(), w_scales=w_s.numpy(), bias=bias8,
         out=R.scaled_mm_int8(q_dyn, w8.t(), s_dyn, w_s, bf, bias8))


if __name__ == "__main__":
    main()
