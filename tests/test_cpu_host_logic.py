"""CPU: host-side logic of the plugin that needs no GPU and no vLLM — attention metadata
(decode / prefill split), KV-cache views, platform capability flags and KV-pool sizing,
env registry, entry points, harness accounting."""
import pytest
import torch

import vllm_metax_amd
from vllm_metax_amd import envs
from vllm_metax_amd.attention import backend as B
from vllm_metax_amd.platform import Mi355xPlatform


def test_entry_points():
    assert vllm_metax_amd.register() == "vllm_metax_amd.platform.Mi355xPlatform"
    # register_patch() returns the list of upstream classes it patched: none without vLLM installed
    assert vllm_metax_amd.register_patch() == [] and vllm_metax_amd.register_model() is None


def test_split_decodes_and_prefills():
    assert B.split_decodes_and_prefills([1, 1, 1]) == (3, 0, 3, 0)
    assert B.split_decodes_and_prefills([1, 1, 7, 300]) == (2, 2, 2, 307)
    assert B.split_decodes_and_prefills([5, 9]) == (0, 2, 0, 14)
    assert B.split_decodes_and_prefills([]) == (0, 0, 0, 0)
    # a decode after the first prefill is treated as a (1-token) prefill, as upstream does
    assert B.split_decodes_and_prefills([1, 8, 1]) == (1, 2, 1, 9)


def test_build_metadata_mixed_batch():
    qsl = torch.tensor([0, 1, 2, 34, 234], dtype=torch.int32)
    seq_lens = torch.tensor([40, 1100, 32, 713], dtype=torch.int32)
    bt = torch.zeros(4, 70, dtype=torch.int32)
    slots = torch.arange(234, dtype=torch.int64)
    md = B.build_metadata(qsl, qsl.tolist(), seq_lens, seq_lens.tolist(), bt, slots, 234, 200, 1100,
                          num_heads=32, head_size=128, dtype=torch.bfloat16)
    assert (md.num_decodes, md.num_decode_tokens, md.num_prefills, md.num_prefill_tokens) == (2, 2, 2, 232)
    assert md.max_decode_seq_len == 1100 and md.max_prefill_query_len == 200
    assert md.prefill_query_start_loc.tolist() == [0, 32, 232]
    assert md.exp_sums.shape == (2, 32, 3) and md.tmp_out.shape == (2, 32, 3, 128)
    md2 = B.build_metadata(qsl[:3], [0, 1, 2], seq_lens[:2], [40, 1100], bt[:2], slots[:2], 2, 1, 1100,
                           32, 128, torch.bfloat16)
    assert md2.num_prefills == 0 and md2.prefill_query_start_loc is None


def test_kv_cache_views():
    shape = B.kv_cache_shape(10, 16, 8, 128)
    assert shape == (2, 10, 16 * 8 * 128)
    kv = torch.arange(2 * 10 * 16 * 8 * 128, dtype=torch.float32).to(torch.bfloat16).reshape(shape)
    kc, vc = B.split_kv_cache(kv, 8, 128)
    assert kc.shape == (10, 8, 16, 16, 8) and vc.shape == (10, 8, 128, 16)
    assert kc.data_ptr() == kv[0].data_ptr() and vc.data_ptr() == kv[1].data_ptr()
    assert kc.stride(0) == vc.stride(0) == 16 * 8 * 128


def test_platform_flags_and_kv_sizing():
    P = Mi355xPlatform
    assert P.supports_fp8() is True and P.use_custom_allreduce() is False
    assert P.dist_backend == "nccl" and "awq" in P.supported_quantization and "gptq" in P.supported_quantization
    assert P.get_attn_backend_cls(None, 128, torch.bfloat16, "auto", 16, True, False).endswith(
        "Mi355xPagedAttentionBackend")
    with pytest.raises(ValueError, match="kv cache"):
        P.get_attn_backend_cls(None, 128, torch.bfloat16, "fp8", 16, True, False)
    with pytest.raises(NotImplementedError):
        P.get_attn_backend_cls(None, 128, torch.bfloat16, "auto", 16, True, True)
    # Llama-3-70B at TP=8 -> 80 layers x 1 kv head x 128 x 2 (K,V) x 2 B = 40 960 B/token
    assert P.kv_bytes_per_token(80, 1, 128) == 40960
    toks = P.kv_pool_tokens(weight_bytes=int(70e9 / 8), num_layers=80, num_kv_heads=1, head_size=128)
    assert toks % 16 == 0 and 5_500_000 < toks < 6_300_000     # "~6 M tokens/rank"
    assert P.kv_pool_tokens(weight_bytes=400 * 10 ** 9, num_layers=80, num_kv_heads=1, head_size=128) == 0

    class Cfg:  # minimal stand-in for VllmConfig
        class parallel_config:
            worker_cls = "auto"

        class cache_config:
            block_size = None

        class model_config:
            disable_cascade_attn = False
    P.check_and_update_config(Cfg)
    assert Cfg.parallel_config.worker_cls == "vllm.v1.worker.gpu_worker.Worker"
    assert Cfg.cache_config.block_size == 16 and Cfg.model_config.disable_cascade_attn is True


def test_envs_registry(monkeypatch):
    assert envs.MI355X_PA_ALLOW_V1 is False
    monkeypatch.setenv("MI355X_MAX_BATCHED_TOKENS", "4096")
    assert envs.MI355X_MAX_BATCHED_TOKENS == 4096
    with pytest.raises(AttributeError):
        envs.NOT_A_KNOB


def test_harness_accounting_matches_baseline_md():
    """Llama-3-8B AWQ per-layer weights ~109 MB (+4.3 MB scales/zeros);
    attention decode 285 MB / layer at batch 64, mean context 1088."""
    from vllm_metax_amd import harness
    cfg = harness.ModelConfig.llama3_8b("awq")
    shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
    wbytes = sum(k * n // 2 for k, n in shapes)
    assert abs(wbytes - 109e6) / 109e6 < 0.01
    meta = sum((k // 128) * n * 2 + (k // 128) * n // 2 for k, n in shapes)
    assert abs(meta - 4.3e6) / 4.3e6 < 0.05
    kv = 64 * 1088 * cfg.kv_heads * cfg.head_dim * 2 * 2
    assert abs(kv - 285e6) / 285e6 < 0.01


def test_decode_kernel_choice_follows_the_upstream_rule():
    import torch
    from vllm_metax_amd.attention.backend import use_paged_attention_v1, v1_max_seq_len, v1_v2_rule
    bf, f32 = torch.bfloat16, torch.float32
    args = (8, 128, 16)                                   # kv heads, head size, block size (Llama-3-8B)
    assert use_paged_attention_v1(64, 32, 1152, *args, bf)          # bench shape: 2048 (seq, head) pairs
    assert use_paged_attention_v1(1, 32, 400, *args, bf)            # one partition
    assert not use_paged_attention_v1(4, 32, 1152, *args, bf)       # few pairs, 3 partitions -> v2
    assert not use_paged_attention_v1(256, 32, 9000, *args, bf)     # beyond the upstream rule's 8192
    # the launcher's LDS budget, not a fixed 8192, bounds v1 (round-1 defect: 6721..8192 chose v1 and
    # the launcher refused it): 32/8 heads keep 4 heads' logits per workgroup
    assert v1_max_seq_len(32, 8, 128, 16, bf) == 6656
    assert v1_max_seq_len(32, 8, 128, 16, f32) == 4928
    for L, want in ((6656, True), (6657, False), (7000, False), (8192, False)):
        assert use_paged_attention_v1(64, 32, L, *args, bf) is want, L
    assert use_paged_attention_v1(64, 32, 4928, *args, f32) and not use_paged_attention_v1(64, 32, 5100, *args, f32)
    # MHA (one head per workgroup) fits the whole upstream range
    assert use_paged_attention_v1(64, 32, 8192, 32, 128, 16, bf)
    # the pure rule
    assert v1_v2_rule(64, 32, 7000, 8192) and not v1_v2_rule(64, 32, 7000, 6656)


def test_decode_workspace_views_and_fixed_geometry():
    """Persistent split-KV workspaces (build() allocated them per step, unsafe under a captured
    graph) and the frozen decode geometry used for HIP-graph capture / replay."""
    import torch
    from vllm_metax_amd.attention import backend as B
    ws = B.DecodeWorkspace(8, 4, 16, 2000, torch.float32, "cpu")
    assert ws.max_parts == 4
    es, ml, to = ws.views(3, 2)
    assert es.shape == (3, 4, 2) and to.shape == (3, 4, 2, 16) and es.is_contiguous() and to.is_contiguous()
    assert es.data_ptr() == ws.exp_sums.data_ptr() and to.data_ptr() == ws.tmp_out.data_ptr()
    import pytest
    with pytest.raises(RuntimeError):
        ws.views(9, 1)
    qsl = torch.tensor([0, 1, 2, 3], dtype=torch.int32)
    sl = torch.tensor([40, 700, 90], dtype=torch.int32)
    bt = torch.zeros(3, 128, dtype=torch.int32)
    slots = torch.zeros(3, dtype=torch.int64)
    md = B.build_metadata(qsl, qsl.tolist(), sl, sl.tolist(), bt, slots, 3, 1, 700, 4, 16, torch.float32, ws,
                          num_kv_heads=4, block_size=16, fixed_decode_len=2000)
    # 3 sequences x 4 kv heads are 12 workgroups per partition: the split goes down to the 128-token floor
    assert md.max_decode_seq_len == 2000 and md.partition_size == 128 and md.exp_sums.shape == (3, 4, 16)
    assert md.exp_sums.data_ptr() == ws.exp_sums.data_ptr()
    assert md.use_v1 is False                              # 12 (seq, head) pairs, several partitions -> v2
    md2 = B.build_metadata(qsl, qsl.tolist(), sl, sl.tolist(), bt, slots, 3, 1, 700, 4, 16, torch.float32, ws,
                           num_kv_heads=4, block_size=16)
    assert md2.max_decode_seq_len == 700 and md2.exp_sums.shape == (3, 4, 6)
    md3 = B.build_metadata(qsl, qsl.tolist(), sl, sl.tolist(), bt, slots, 3, 1, 700, 4, 16, torch.float32, ws)
    assert md3.partition_size == 512 and md3.exp_sums.shape == (3, 4, 2)   # kv head count unknown: the reference's 512


def test_decode_partition_size_and_workspace():
    """Host rule behind mi355x_paged_attention_v2_ps: the reference's 512-token partitions unless they leave CUs
    without a workgroup; never below 128 tokens; a multiple of the block size and of 16."""
    # Llama-3-8B, batch 64: 64 x 8 kv heads already fill the chip
    assert B.decode_partition_size(64, 32, 8, 1152, 16) == 512
    # one TP = 8 rank of Llama-3-70B, batch 64 (8 q / 1 kv head, one workgroup per kv head): 4 partitions of 288
    assert B.decode_heads_per_workgroup(8, 1) == 8 and B.decode_heads_per_workgroup(32, 8) == 4
    assert B.decode_partition_size(64, 8, 1, 1152, 16) == 288
    assert B.decode_partition_size(8, 8, 1, 1152, 16) == 128          # small batch: the floor
    assert B.decode_partition_size(64, 8, 1, 100, 16) == 512          # short contexts: one partition anyway
    assert B.decode_partition_size(1, 8, 1, 131072, 32) == 512        # 256 partitions of 512 are enough
    for n in (1, 3, 64, 200):
        for L in (129, 1000, 5000, 40000):
            for bs in (8, 16, 32):
                ps = B.decode_partition_size(n, 8, 1, L, bs)
                assert 128 <= ps <= 512 and ps % bs == 0 and ps % 16 == 0
    ws = B.DecodeWorkspace(64, 8, 128, 4096, torch.bfloat16, "cpu")
    ps = B.decode_partition_size(64, 8, 1, 1152, 16)
    es, ml, to = ws.views(64, -(-1152 // ps))
    assert es.shape == (64, 8, 4) and to.shape == (64, 8, 4, 128) and es.is_contiguous() and to.is_contiguous()
    ps1 = B.decode_partition_size(1, 8, 1, 4096, 16)
    es, ml, to = ws.views(1, -(-4096 // ps1))                         # one sequence, 32 partitions of 128
    assert es.shape == (1, 8, 32)
    md = B.build_metadata(torch.arange(65, dtype=torch.int32), list(range(65)), torch.full((64,), 1100, dtype=torch.int32),
                          [1100] * 64, torch.zeros(64, 72, dtype=torch.int32), torch.arange(64), 64, 1, 1100,
                          num_heads=8, head_size=128, dtype=torch.bfloat16, workspace=ws, num_kv_heads=1,
                          fixed_decode_len=1152)
    assert md.partition_size == 288 and md.exp_sums.shape == (64, 8, 4) and md.use_v1 is False


def test_scaled_mm_operands_in_place_rule(monkeypatch):
    """Which operands of the m > 320 8-bit GEMM need no operand image (and therefore no scratch): fp8, k % 128 == 0,
    16-byte aligned rows — the rule of fp8_gemm.hip run_fp8, mirrored by the bindings for the workspace they pass."""
    from vllm_metax_amd import _custom_ops as ops
    monkeypatch.delenv("MI355X_F8_ROWMAJOR", raising=False)
    a = torch.zeros(512, 4096, dtype=torch.uint8).view(torch.float8_e4m3fn)
    assert ops._scaled_mm_in_place(a, 4096, 0) == (True, True)
    assert ops._scaled_mm_in_place(a, 4100, 0) == (True, False)            # weight rows not 16-byte aligned
    assert ops._scaled_mm_in_place(a[:, :4032], 4096, 0) == (False, False)  # k % 128 != 0: the 64-byte-stage kernel
    assert ops._scaled_mm_in_place(torch.zeros(512, 4096, dtype=torch.int8), 4096, 0) == (False, False)
    wide = torch.zeros(512, 4096 + 8, dtype=torch.uint8).view(torch.float8_e4m3fn)
    assert ops._scaled_mm_in_place(wide[:, :4096], 4096, 0) == (False, False)   # activation rows 8 bytes off
    monkeypatch.setenv("MI355X_F8_ROWMAJOR", "1")
    assert ops._scaled_mm_in_place(a, 4096, 0) == (True, False)
    monkeypatch.setenv("MI355X_F8_ROWMAJOR", "0")
    assert ops._scaled_mm_in_place(a, 4096, 0) == (False, False)


def test_scaled_mm_split_plan_is_a_host_function():
    """mi355x_scaled_mm_split_elems (host-only planner, no GPU call): sk * m * n workspace elements for shapes whose
    256 x 256 tiles leave the chip mostly idle, 0 where the prefill kernel is not split."""
    from vllm_metax_amd import _abi
    f = _abi.load().mi355x_scaled_mm_split_elems
    assert f(576, 4096, 4096) == 4 * 576 * 4096          # 48 tiles, 32 k-steps
    assert f(2048, 1280, 8192) == 5 * 2048 * 1280        # one TP = 8 rank's qkv at a 2048-token chunk
    assert f(8192, 28672, 4096) == 0 and f(8192, 1280, 8192) == 0     # >= 160 tiles
    assert f(2048, 8192, 1024) == 0                      # 256 tiles
    assert f(512, 512, 512) == 0                         # 4 k-steps: nothing to split
    assert f(64, 4096, 4096) == 0 and f(320, 4096, 4096) == 0        # decode-kernel territory
    assert f(576, 4096, 4000) == 0                       # k % 64 != 0: not the prefill kernel's shape
