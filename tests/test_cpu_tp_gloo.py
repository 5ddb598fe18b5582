"""CPU, world_size 2 over gloo: the tensor-parallel decomposition used by bench.py --gpus N
(one process per GPU; heads / FFN sharded; all-reduce after the row-parallel o_proj and
down_proj; all-gather of the vocab-parallel logits) reproduces the unsharded layer.

The layer arithmetic here is the CPU oracle (the HIP ops need a GPU); what is under test is the
sharding / collective structure of vllm_metax_amd/harness.py: which slices each
rank owns and where the collectives sit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_ops as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_weights(seed, h, heads, kv_heads, d, ffn, group):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)

    def lin(k, n):
        w = rng.integers(0, 16, size=(k, n), dtype=np.uint8)
        z = rng.integers(0, 16, size=(k // group, n), dtype=np.uint8)
        s = (torch.rand(k // group, n, generator=g) * 4e-3 + 1e-3).to(torch.bfloat16)
        return w, z, s
    return {"q": lin(h, heads * d), "k": lin(h, kv_heads * d), "v": lin(h, kv_heads * d),
            "o": lin(heads * d, h), "gate": lin(h, ffn), "up": lin(h, ffn), "down": lin(ffn, h)}


def _gemm(x, wzs, cols=None, rows=None, group=32):
    """x @ dequant(w) with optional column slice (column-parallel) or row slice (row-parallel)."""
    w, z, s = wzs
    if cols is not None:
        w, z, s = w[:, cols], z[:, cols], s[:, cols]
    if rows is not None:
        w = w[rows]
        z = z[rows.start // group:rows.stop // group]
        s = s[rows.start // group:rows.stop // group]
    wd = R.w4_dequant(w, z, s, group)
    return (x.double() @ wd.double()).to(x.dtype)


def _layer(x, W, heads, kv_heads, d, ffn, rank, tp, group, reduce_fn):
    """One decoder layer's linear path (attention replaced by a per-head identity mix, which is
    head-local like the real attention): returns the layer output on every rank."""
    hq, hkv, f = heads // tp, kv_heads // tp, ffn // tp
    qs = slice(rank * hq * d, (rank + 1) * hq * d)
    ks = slice(rank * hkv * d, (rank + 1) * hkv * d)
    fs = slice(rank * f, (rank + 1) * f)
    q = _gemm(x, W["q"], cols=qs, group=group)
    k = _gemm(x, W["k"], cols=ks, group=group)
    v = _gemm(x, W["v"], cols=ks, group=group)
    G = heads // kv_heads
    attn = (q.view(-1, hq, d).float() * 0.5 + k.view(-1, hkv, d).repeat_interleave(G, 1).float() * 0.25
            + v.view(-1, hkv, d).repeat_interleave(G, 1).float() * 0.25).to(x.dtype).reshape(-1, hq * d)
    o = reduce_fn(_gemm(attn, W["o"], rows=qs, group=group).float()).to(x.dtype)   # row-parallel
    gate = _gemm(o, W["gate"], cols=fs, group=group)
    up = _gemm(o, W["up"], cols=fs, group=group)
    act = R.silu_and_mul(torch.cat([gate, up], dim=-1))
    return reduce_fn(_gemm(act, W["down"], rows=fs, group=group).float()).to(x.dtype)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, heads, kv_heads, d, ffn, group = 128, 4, 2, 32, 256, 32
    W = _make_weights(0, h, heads, kv_heads, d, ffn, group)
    x = (torch.randn(6, h, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16)

    def allreduce(t):
        dist.all_reduce(t)
        return t
    out = _layer(x, W, heads, kv_heads, d, ffn, rank, world, group, allreduce)
    # vocab-parallel logits + all-gather (harness._logits_argmax)
    vocab = 64
    lm = (torch.randn(h, vocab, generator=torch.Generator().manual_seed(2)) * 0.02).to(torch.bfloat16)
    shard = lm[:, rank * vocab // world:(rank + 1) * vocab // world]
    part = torch.matmul(out.float(), shard.float())
    parts = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(parts, part)
    tok = torch.cat(parts, dim=-1).argmax(-1)
    if rank == 0:
        q.put((out.float().numpy(), tok.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_tp2_matches_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out_tp, tok_tp = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    h, heads, kv_heads, d, ffn, group = 128, 4, 2, 32, 256, 32
    W = _make_weights(0, h, heads, kv_heads, d, ffn, group)
    x = (torch.randn(6, h, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16)
    ref = _layer(x, W, heads, kv_heads, d, ffn, 0, 1, group, lambda t: t)
    lm = (torch.randn(h, 64, generator=torch.Generator().manual_seed(2)) * 0.02).to(torch.bfloat16)
    tok_ref = torch.matmul(ref.float(), lm.float()).argmax(-1)
    # partial sums are rounded to bf16 once after the fp32 all-reduce: within 1 bf16 ulp
    err = np.abs(out_tp - ref.float().numpy()).max()
    assert err <= 2.0 ** -7 * np.abs(ref.float().numpy()).max() * 1.01, err
    assert np.array_equal(tok_tp, tok_ref.numpy())


def test_harness_shard_sizes():
    """Per-rank shapes of the harness layers match BASELINE.md §3 (Qwen2-72B / Llama-70B TP=8)."""
    from vllm_metax_amd import harness
    c = harness.ModelConfig.qwen2_72b("gptq", tp=8)
    hq, hkv, ffn = c.heads // c.tp, max(c.kv_heads // c.tp, 1), c.ffn // c.tp
    assert (c.hidden, (hq + 2 * hkv) * c.head_dim) == (8192, 1280)
    assert (hq * c.head_dim, c.hidden) == (1024, 8192)
    assert (c.hidden, 2 * ffn) == (8192, 7424) and (ffn, c.hidden) == (3712, 8192)
    c = harness.ModelConfig.llama3_70b("fp8", tp=8)
    assert (c.hidden, 2 * (c.ffn // 8)) == (8192, 7168) and (c.ffn // 8) == 3584
    for tp in (1, 2, 4, 8):            # the bench's TP=N shards of Llama-3-8B stay group-aligned
        c = harness.ModelConfig.llama3_8b("awq", tp=tp)
        assert (c.heads // tp) * c.head_dim % c.group_size == 0
        assert (c.ffn // tp) % c.group_size == 0


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util, pathlib
    spec = importlib.util.spec_from_file_location("bench", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    q.put((rank, bench.max_over_ranks(1.0 + rank, "cpu", world)))
    dist.destroy_process_group()


def test_dp_replica_aggregation_world2():
    """bench.py --gpus N default (dp): whole-job tokens = all ranks' tokens, clock = slowest rank."""
    import importlib.util, pathlib
    import torch.multiprocessing as mp
    spec = importlib.util.spec_from_file_location("bench", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.whole_job_tokens(2, 64, 128, 8, 1) == 2 * 64 * 128 * 8      # dp8: 8 replicas
    assert bench.whole_job_tokens(2, 64, 128, 8, 8) == 2 * 64 * 128          # tp8: one job
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    assert got == {0: 2.0, 1: 2.0}


# ---------------------------------------------------------------------------------------------
# The harness itself under tensor parallelism (the test above re-implements the layer;
# vllm_metax_amd/harness.py's own TP branch was never executed).  Here harness.HotPathModel runs on
# the CPU with its op surface replaced by the oracle (tests/oracle_ops_shim.py): world_size 2 over
# gloo, every rank builds ITS shard of the same model (QLinear cuts it out of the full layer), runs
# a prefill chunk and decode steps with the real all-reduce / all-gather placement, and the greedy
# tokens must equal those of the unsharded model.
@pytest.fixture(autouse=True)
def _restore_op_surface():
    """_install_shim() swaps the op surface of harness / attention.backend IN THIS PROCESS (the tp = 1 comparison
    run): put the real modules back after every test, so that a test that runs later in the same process sees them
    whatever the collection order is."""
    from vllm_metax_amd import harness
    from vllm_metax_amd.attention import backend
    saved = (harness.ops, backend.ops)
    yield
    harness.ops, backend.ops = saved
    backend.v1_max_seq_len.cache_clear()


def _install_shim():
    from tests import oracle_ops_shim as shim
    from vllm_metax_amd import harness
    from vllm_metax_amd.attention import backend
    harness.ops = shim
    backend.ops = shim
    backend.v1_max_seq_len.cache_clear()
    return harness


def _tiny_cfg(harness, tp, rank, quant="awq"):
    cfg = harness.ModelConfig.tiny(quant)
    cfg.group_size = 64
    cfg.tp, cfg.tp_rank = tp, rank
    return cfg


def _harness_tokens(harness, cfg, group, steps=3):
    torch.manual_seed(0)
    model = harness.HotPathModel(cfg, 3, 48, device="cpu", dtype=torch.bfloat16, seed=0, tp_group=group)
    model.setup_decode(3, 20, 48)
    tok = torch.randint(0, cfg.vocab, (3, 20), generator=torch.Generator().manual_seed(5))
    first = model.prefill(tok, [0, 1, 2], 0)
    model.d_tokens.copy_(first)
    model.set_decode_lengths(torch.full((3,), 20))
    out = [first.clone()]
    for _ in range(steps):
        model.decode_step(use_graph=False)
        out.append(model.d_tokens.clone())
    return torch.stack(out)


def _harness_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    harness = _install_shim()
    toks = _harness_tokens(harness, _tiny_cfg(harness, world, rank), dist.group.WORLD)
    if rank == 0:
        q.put(toks.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_harness_tp2_tokens_equal_tp1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    toks_tp = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    harness = _install_shim()
    toks_1 = _harness_tokens(harness, _tiny_cfg(harness, 1, 0), None).numpy()
    assert toks_tp.shape == toks_1.shape == (4, 3)
    assert (toks_tp == toks_1).all(), f"TP=2 tokens {toks_tp.tolist()} != TP=1 tokens {toks_1.tolist()}"


def test_qlinear_shards_are_slices_of_the_full_layer():
    """Column / row shards cut by QLinear reproduce x @ W_full restricted to the shard (oracle GEMM)."""
    harness = _install_shim()
    cfg = harness.ModelConfig.tiny("awq")
    cfg.group_size = 64
    k, n = 256, 384
    x = (torch.randn(5, k, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16)

    def mk(**kw):
        return harness.QLinear(k, n, cfg, torch.bfloat16, "cpu", torch.Generator().manual_seed(7), **kw)
    full = mk()(x)
    cols = [(64, 64), (256, 32)]
    part = mk(cols=cols)(x)
    assert torch.equal(part, torch.cat([full[:, a:a + l] for a, l in cols], dim=1))
    # row shards: partial sums add up to the full product (fp32 accumulate, then one rounding)
    lo = mk(rows=(0, 128))(x[:, :128].contiguous())
    hi = mk(rows=(128, 128))(x[:, 128:].contiguous())
    err = (lo.float() + hi.float() - full.float()).abs().max().item()
    assert err <= 2.0 ** -6 * full.float().abs().max().item()
