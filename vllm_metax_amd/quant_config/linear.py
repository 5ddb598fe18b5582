"""vLLM-independent core of the AWQ / GPTQ linear methods (the code the vLLM-facing classes in
awq.py / gptq.py delegate to), so that it can be exercised without upstream vLLM.

ref: vllm_metax/quant_config/awq.py:69-80 (process_weights_after_loading), :118-159
(_apply_awq); vllm_metax/quant_config/gptq.py:49-129, :180-229 (_apply_gptq)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _custom_ops as ops
from .. import envs

# Optional (MI355X_PREPACK_WEIGHTS=1): the prefill GEMM's weight operand image, dequantised once in
# process_weights_after_loading instead of inside every prefill-sized awq_gemm / gptq_gemm call
# (n * k * 2 bytes per layer; bit-identical results).  The image belongs to the LAYER: the linear methods
# keep it as `layer._mi355x_prefill_image` and hand it to apply_awq / apply_gptq (`image=`) or to
# apply_w4a16_image — never a table keyed by a device address (a freed-and-reused address would serve
# another layer's weights).
PREFILL_IMAGE_MIN_M = 384     # = _custom_ops.W4_PREPACKED_MIN_M (was 1024 before the image GEMM could split K)
IMAGE_ATTR = "_mi355x_prefill_image"


def make_prefill_image(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                       gptq_zeros: bool):
    """(image, n, k) for a repacked 4-bit layer, or None when the switch is off / the shape has no image."""
    if not envs.MI355X_PREPACK_WEIGHTS or scales.dtype == torch.float32:
        return None
    n = scales.shape[1]
    k = qweight.numel() * 8 // n
    if n % 64 or k % 32 or (k // scales.shape[0]) % 32:
        return None
    # on by default: never let the image be what exhausts the device (n * k * 2 bytes; keep a 10 % margin of the HBM
    # free for the KV pool upstream sizes afterwards) — without it the layer simply dequantises per call
    if qweight.is_cuda:
        free, total = torch.cuda.mem_get_info(qweight.device)
        if free - 2 * n * k < total // 10:
            return None
    return ops.w4a16_prepack(qweight, qzeros, scales, gptq_zeros), n, k


def attach_prefill_image(layer, gptq_zeros: bool) -> None:
    """process_weights_after_loading hook: keep the image on the layer object (dropped with it)."""
    setattr(layer, IMAGE_ATTR, make_prefill_image(layer.qweight.data, layer.qzeros.data, layer.scales.data,
                                                  gptq_zeros))


def layer_image(layer, x: torch.Tensor):
    """The layer's image when this call is prefill-sized and contiguous, else None."""
    img = getattr(layer, IMAGE_ATTR, None)
    if img is None or x.numel() // x.shape[-1] < PREFILL_IMAGE_MIN_M or not x.is_contiguous():
        return None
    return img


def apply_w4a16_image(x: torch.Tensor, image: torch.Tensor, n: int, k: int,
                      bias: Optional[torch.Tensor]) -> torch.Tensor:
    """The prefill-sized linear on a load-time weight image (registered as torch.ops.vllm._apply_w4a16_image)."""
    out = ops.w4a16_gemm_prepacked(x.reshape(-1, x.shape[-1]), image, n, k)
    if bias is not None:
        out.add_(bias)
    return out.reshape(x.shape[:-1] + (n,))


def apply_w4a16_image_fake(x, image, n, k, bias):
    return torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)


def awq_process_weights(qweight: torch.Tensor, group_size: int) -> torch.Tensor:
    """AWQ -> exllama repack, once after loading (awq.py:72-80).  group_size % 32 != 0 keeps
    the original layout (served by awq_dequantize + matmul, as in the reference)."""
    if group_size % 32:
        return qweight
    return ops.awq_to_gptq_4bit(qweight)


def apply_awq(x: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor,
              bias: Optional[torch.Tensor], pack_factor: int, group_size: int, image=None) -> torch.Tensor:
    """ref: awq.py:118-159.  `image`: the layer's load-time prefill image (make_prefill_image), optional."""
    reshaped_x = x.reshape(-1, x.shape[-1])
    if group_size % 32:
        out_shape = x.shape[:-1] + (qweight.shape[-1] * pack_factor,)
        w = ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0)
        out = torch.matmul(reshaped_x, w)
    else:
        n = qweight.shape[0]                      # declared [N, K/8]
        out_shape = x.shape[:-1] + (n,)
        temp_space = torch.empty(0, dtype=torch.float32, device=x.device)
        if reshaped_x.shape[0] <= 64:             # split-K workspace only matters for decode
            temp_space = torch.zeros(reshaped_x.shape[0], n, dtype=torch.float32, device=x.device)
        img = image if reshaped_x.shape[0] >= PREFILL_IMAGE_MIN_M else None
        if img is not None and reshaped_x.is_contiguous():
            out = ops.w4a16_gemm_prepacked(reshaped_x, img[0], img[1], img[2])
        else:
            out = ops.awq_gemm(reshaped_x, qweight, qzeros, scales, pack_factor, temp_space,
                               reshaped_x.dtype == torch.bfloat16)
    if bias is not None:
        out.add_(bias)
    return out.reshape(out_shape)


def apply_awq_fake(x, qweight, scales, qzeros, bias, pack_factor, group_size):
    """ref: awq.py:101-115 (fake impl for torch.compile)."""
    n = qweight.shape[-1] * pack_factor if group_size % 32 else qweight.shape[0]
    return torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)


def gptq_process_weights(qweight: torch.Tensor, g_idx: torch.Tensor, desc_act: bool,
                         weight_bits: int) -> torch.Tensor:
    """ref: gptq.py:49-75 — argsort g_idx when act-order, then exllama shuffle in place.
    Returns the g_idx to keep on the layer (argsort permutation or empty)."""
    if desc_act:
        g_idx = torch.argsort(g_idx).to(torch.int)
    else:
        g_idx = torch.empty((0,), dtype=torch.int, device=g_idx.device)
    ops.gptq_shuffle(qweight, g_idx, weight_bits)
    return g_idx


def apply_gptq(x: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor,
               bias: Optional[torch.Tensor], g_idx: torch.Tensor, use_exllama: bool,
               weight_bits: int, group_size: int, desc_act: bool, image=None) -> torch.Tensor:
    """ref: gptq.py:180-229.  `image`: the layer's load-time prefill image, optional."""
    reshaped_x = x.reshape(-1, x.shape[-1])
    out_shape = x.shape[:-1] + (qweight.shape[-1],)
    perm_space = torch.empty(0)
    temp_space = torch.empty(0)
    if desc_act:
        perm_space = torch.empty(reshaped_x.shape[0], reshaped_x.shape[1], dtype=torch.float16,
                                 device=x.device)
    if reshaped_x.shape[0] <= 64:
        temp_space = torch.zeros(reshaped_x.shape[0], qweight.shape[1], dtype=torch.float32,
                                 device=x.device)
    img = image if (weight_bits == 4 and not desc_act and reshaped_x.shape[0] >= PREFILL_IMAGE_MIN_M) else None
    if img is not None and reshaped_x.is_contiguous():
        out = ops.w4a16_gemm_prepacked(reshaped_x, img[0], img[1], img[2])
    else:
        out = ops.gptq_gemm(reshaped_x, qweight, qzeros, scales, g_idx, use_exllama, weight_bits,
                            group_size, perm_space, temp_space, reshaped_x.dtype == torch.bfloat16)
    if bias is not None:
        out.add_(bias)
    return out.reshape(out_shape)


def apply_gptq_fake(x, qweight, scales, qzeros, bias, g_idx, use_exllama, weight_bits, group_size,
                    desc_act):
    """ref: gptq.py:164-177."""
    return torch.empty(x.shape[:-1] + (qweight.shape[-1],), dtype=x.dtype, device=x.device)
