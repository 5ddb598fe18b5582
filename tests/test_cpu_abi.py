"""CPU: the C-ABI library loads and exports every symbol include/mi355x_hotpath.h declares;
the torch bindings register the reference's op names.  No compute (no GPU here)."""
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mi355x_hotpath.h"


def header_symbols():
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355x_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from vllm_metax_amd import _abi
    return _abi


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for needed in ["mi355x_paged_attention_v1", "mi355x_paged_attention_v2",
                   "mi355x_paged_prefill_attention", "mi355x_reshape_and_cache",
                   "mi355x_reshape_and_cache_flash", "mi355x_copy_blocks", "mi355x_swap_blocks",
                   "mi355x_awq_gemm", "mi355x_gptq_gemm", "mi355x_awq_to_gptq_4bit",
                   "mi355x_gptq_shuffle", "mi355x_awq_dequantize", "mi355x_scaled_mm_fp8",
                   "mi355x_rms_norm", "mi355x_fused_add_rms_norm", "mi355x_rotary_embedding",
                   "mi355x_rms_norm_static_fp8_quant", "mi355x_rms_norm_dynamic_per_token_quant",
                   "mi355x_static_scaled_fp8_quant", "mi355x_silu_and_mul"]:
        assert needed in syms


def test_library_exports_every_declared_symbol(built):
    lib = built.load()
    for sym in header_symbols():
        assert hasattr(lib, sym), f"{sym} declared in the header but not exported"
    # and the ctypes prototype table covers exactly the header
    assert sorted(built.PROTOTYPES) == header_symbols()
    assert lib.mi355x_abi_version() == built.ABI_VERSION == 5


def test_argument_errors_surface_without_a_gpu(built):
    """Validation happens on the host before any launch, so these are safe on CPU."""
    lib = built.load()
    rc = lib.mi355x_paged_attention_v1(None, None, None, None, 4, 32, 8, 72, 16, 1.0, None, None,
                                       4, 64, None, 0, 0, 0, built.BF16, built.KV_AUTO, None, None, None)
    assert rc == -2 and "Unsupported head size" in built.last_error()
    rc = lib.mi355x_paged_attention_v1(None, None, None, None, 4, 32, 8, 128, 16, 1.0, None, None,
                                       4, 64, None, 0, 0, 0, built.BF16, 7, None, None, None)
    assert rc == -1 or "kv cache" in built.last_error()
    rc = lib.mi355x_awq_gemm(None, None, None, None, None, None, 0, None, 0, 4, 100, 512, 128, 512, built.BF16, None)
    assert rc == -2 and "multiple of 8" in built.last_error()
    # v1 keeps the logits of a whole sequence in LDS: the limit the backend's v1 / v2 choice uses
    assert lib.mi355x_paged_attention_v1_max_seq_len(64, 32, 8, 128, 16, built.BF16) == 6656
    assert lib.mi355x_paged_attention_v1_max_seq_len(64, 32, 8, 128, 16, built.F32) == 4928
    assert lib.mi355x_paged_attention_v1_max_seq_len(64, 32, 32, 128, 16, built.BF16) == 26048
    rc = lib.mi355x_gptq_gemm(None, None, None, None, None, None, None, None, 0, None, 0, 4, 128, 512, 3, 128,
                              built.BF16, None)
    assert rc == -2 and "4-bit" in built.last_error()
    rc = lib.mi355x_rms_norm(None, None, None, 1e-5, 0, 4096, 4096, built.BF16, None)
    assert rc == 0   # empty input is a no-op


def test_ops_raise_on_cpu_tensors(built):
    """No CPU fallback: the Python op surface refuses non-GPU tensors."""
    from vllm_metax_amd import _custom_ops as ops
    x = torch.zeros(2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.rms_norm(torch.empty_like(x), x, torch.ones(64, dtype=torch.bfloat16), 1e-5)
    with pytest.raises(RuntimeError):
        ops.reshape_and_cache(x, x, x, x, torch.zeros(2, dtype=torch.int64), "fp8")
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.reshape_and_cache(x, x, x, x, torch.zeros(2, dtype=torch.int64), "fp8_e3m4")


def test_torch_bindings_register_reference_op_names(built):
    import vllm_metax_amd._C  # noqa: F401
    for name in ["paged_attention_v1", "paged_attention_v2", "rms_norm", "fused_add_rms_norm",
                 "rms_norm_static_fp8_quant", "fused_add_rms_norm_static_fp8_quant",
                 "rms_norm_dynamic_per_token_quant", "rotary_embedding", "batched_rotary_embedding", "awq_gemm",
                 "awq_dequantize", "awq_to_gptq_4bit", "gptq_gemm", "gptq_shuffle",
                 "cutlass_scaled_mm", "cutlass_scaled_mm_supports_fp8", "static_scaled_fp8_quant",
                 "dynamic_scaled_fp8_quant", "dynamic_per_token_scaled_fp8_quant", "silu_and_mul",
                 "silu_and_mul_quant", "weak_ref_tensor"]:
        assert hasattr(torch.ops._C, name), name
    for name in ["reshape_and_cache", "reshape_and_cache_flash", "copy_blocks", "swap_blocks", "convert_fp8"]:
        assert hasattr(torch.ops._C_cache_ops, name), name
    assert hasattr(torch.ops._C_cuda_utils, "get_device_attribute")
    # schema strings are the reference's (torch_bindings.cpp:45-55, 233-236)
    s = str(torch.ops._C.paged_attention_v1.default._schema)
    assert "! -> ) out" in s and "str kv_cache_dtype" in s and "int blocksparse_head_sliding_step" in s
    s = str(torch.ops._C.awq_gemm.default._schema)
    assert "SymInt split_k_iters" in s and "bool dtype_bf16" in s
    assert torch.ops._C.cutlass_scaled_mm_supports_fp8(90) is True
