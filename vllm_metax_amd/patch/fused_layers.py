"""vLLM-independent core of the fused decoder-layer forwards (what register_patch() routes vLLM's Llama / Qwen2
layers through).  Every function is bit-identical to the op sequence vLLM would run on the plain op surface
(RMSNorm.forward_cuda -> torch.ops._C.fused_add_rms_norm, the linear methods' apply -> awq_gemm / gptq_gemm,
SiluAndMul.forward_cuda -> torch.ops._C.silu_and_mul); tests/test_gpu_patch_fused_layers.py checks that on fake
layer objects, without vLLM.

What is fused (all MI355X-side entry points of include/mi355x_hotpath.h):
  prefill (>= 1024 tokens, layers carrying a load-time weight image — MI355X_PREPACK_WEIGHTS=1):
      fused_add_rms_norm_image -> w4a16_gemm_prepacked(A image, SILU epilogue, output image) ->
      w4a16_gemm_prepacked(A image): the norm writes the gate_up GEMM's operand image, the GEMM applies
      silu_and_mul in its epilogue and writes the down GEMM's operand image: no re-tiling, no row-major
      intermediate, no silu launch;
  prefill without images (AWQ): awq_gemm_silu_mul_packed -> awq_gemm_packed_a;
  decode (<= 64 tokens, AWQ): awq_gemm_silu_mul (SILU epilogue of the stripe kernel), and — single GPU only,
      where no all-reduce sits between GEMM and norm — the down / o_proj GEMMs leave their split-K slabs to the
      next norm (awq_gemm_deferred + fused_add_rms_norm_slabs).
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Tuple

import torch

from .. import _custom_ops as ops
from ..quant_config import linear as qlinear


@dataclasses.dataclass
class W4Linear:
    """What the fused paths need to know about one AWQ / GPTQ linear layer (taken from a vLLM layer by
    `patch.llama.describe`, or built by hand in tests)."""
    kind: str                     # "awq" | "gptq"
    qweight: torch.Tensor         # exllama layout (after process_weights_after_loading)
    qzeros: torch.Tensor
    scales: torch.Tensor
    group_size: int
    g_idx: Optional[torch.Tensor] = None       # GPTQ: empty when not act-order
    image: Optional[tuple] = None              # (image, n, k): the layer's load-time prefill image, or None
    bias: Optional[torch.Tensor] = None

    @property
    def n(self) -> int:
        return self.scales.shape[1]

    def plain(self, x: torch.Tensor) -> torch.Tensor:
        """The unfused linear: exactly what torch.ops.vllm._apply_awq / _apply_gptq compute."""
        if self.kind == "awq":
            return qlinear.apply_awq(x, self.qweight, self.scales, self.qzeros, self.bias, 8, self.group_size,
                                     image=self.image)
        return qlinear.apply_gptq(x, self.qweight, self.scales, self.qzeros, self.bias, self.g_idx, True, 4,
                                  self.group_size, False, image=self.image)


def fusable(*layers: Optional[W4Linear]) -> bool:
    """The fused paths cover 4-bit layers without bias and without act-order, group sizes the kernels take."""
    for L in layers:
        if L is None or L.bias is not None or L.kind not in ("awq", "gptq"):
            return False
        if L.g_idx is not None and L.g_idx.numel() > 0:
            return False
        if L.group_size % 32 or L.scales.dtype == torch.float32:
            return False
    return True


def fused_mlp(x, gate_up: W4Linear, down: W4Linear) -> torch.Tensor:
    """down(silu_and_mul(gate_up(x))) — x [tokens, hidden] (a tensor, or the operand image a fused norm wrote).
    Bit-identical to the three ops on the plain surface."""
    packed_in = isinstance(x, ops.PackedOperand)
    m = x.m if packed_in else x.shape[0]
    if m >= 1024 and gate_up.image is not None and down.image is not None and gate_up.n % 256 == 0:
        act = ops.w4a16_gemm_prepacked(x, gate_up.image[0], gate_up.image[1], gate_up.image[2], silu=True,
                                       out_image=True)
        return ops.w4a16_gemm_prepacked(act, down.image[0], down.image[1], down.image[2])
    if packed_in:
        raise RuntimeError("fused_mlp: an operand image needs layers with a prefill image")
    if gate_up.kind == "awq" and down.kind == "awq":
        if m >= 1024:
            act = ops.awq_gemm_silu_mul_packed(x, gate_up.qweight, gate_up.qzeros, gate_up.scales)
            if act is not None:
                return ops.awq_gemm_packed_a(act, down.qweight, down.qzeros, down.scales)
        act = ops.awq_gemm_silu_mul(x, gate_up.qweight, gate_up.qzeros, gate_up.scales)
        if act is not None:
            return down.plain(act)
    gu = gate_up.plain(x)
    act = torch.empty(gu.shape[:-1] + (gu.shape[-1] // 2,), dtype=gu.dtype, device=gu.device)
    ops.silu_and_mul(act, gu)
    return down.plain(act)


def fused_norm_mlp(x: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float,
                   gate_up: W4Linear, down: W4Linear) -> Tuple[torch.Tensor, torch.Tensor]:
    """post_attention_layernorm (fused add) + MLP: (mlp_out, residual).  `residual` receives x + residual in
    place, as RMSNorm.forward_cuda(x, residual) does.  Prefill with images: the norm writes the gate_up GEMM's
    operand image itself; otherwise fused_add_rms_norm, then fused_mlp."""
    m = x.shape[0]
    if m >= 1024 and gate_up.image is not None and down.image is not None and gate_up.n % 256 == 0:
        h = ops.fused_add_rms_norm_image(x, residual, norm_weight, eps)
        if h is not None:
            return fused_mlp(h, gate_up, down), residual
    ops.fused_add_rms_norm(x, residual, norm_weight, eps)
    return fused_mlp(x, gate_up, down), residual


def fused_norm_linear(x: torch.Tensor, residual: Optional[torch.Tensor], norm_weight: torch.Tensor, eps: float,
                      lin: W4Linear) -> Tuple[torch.Tensor, torch.Tensor]:
    """input_layernorm + qkv_proj: (qkv, residual).  residual None: first layer (rms_norm; the residual becomes
    x).  Prefill with an image: the norm writes the GEMM's operand image."""
    m = x.shape[0]
    img_ok = m >= 1024 and lin.image is not None
    if residual is None:
        residual = x
        h = ops.rms_norm_image(x, norm_weight, eps) if img_ok else None
        if h is None:
            h = torch.empty_like(x)
            ops.rms_norm(h, x, norm_weight, eps)
            return lin.plain(h), residual
        return ops.w4a16_gemm_prepacked(h, lin.image[0], lin.image[1], lin.image[2]), residual
    h = ops.fused_add_rms_norm_image(x, residual, norm_weight, eps) if img_ok else None
    if h is None:
        ops.fused_add_rms_norm(x, residual, norm_weight, eps)
        return lin.plain(x), residual
    return ops.w4a16_gemm_prepacked(h, lin.image[0], lin.image[1], lin.image[2]), residual
