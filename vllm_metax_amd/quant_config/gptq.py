"""GPTQ config / linear method for vLLM (ref: vllm_metax/quant_config/gptq.py:23-238).
The reference's warm-up GEMM (gptq.py:76-129, it triggers MetaX's GEMV autotuner) has no
counterpart: the MI355X kernels have no run-time autotuning."""
from typing import Optional

import torch
from vllm.model_executor.layers.quantization.gptq import ExllamaState, GPTQConfig
from vllm.model_executor.layers.quantization.gptq import GPTQLinearMethod as _GPTQLinearMethod
from vllm.model_executor.layers.quantization.utils.gptq_utils import get_linear_quant_method
from vllm.utils.torch_utils import direct_register_custom_op

from . import linear
from .hook_register import register_quantization_config


@register_quantization_config("gptq")
class Mi355xGPTQConfig(GPTQConfig):
    def get_supported_act_dtypes(self):
        return [torch.half, torch.bfloat16]

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        return get_linear_quant_method(self, layer, prefix, GPTQLinearMethod)


class GPTQLinearMethod(_GPTQLinearMethod):
    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        layer.qzeros = torch.nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.qweight = torch.nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.scales = torch.nn.Parameter(layer.scales.data, requires_grad=False)
        g_idx = linear.gptq_process_weights(layer.qweight.data, layer.g_idx.data,
                                            self.quant_config.desc_act,
                                            self.quant_config.weight_bits)
        layer.g_idx = torch.nn.Parameter(g_idx, requires_grad=False)
        layer.exllama_state = ExllamaState.READY
        if self.quant_config.weight_bits == 4 and not self.quant_config.desc_act:
            linear.attach_prefill_image(layer, True)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        img = linear.layer_image(layer, x)
        if img is not None:
            return torch.ops.vllm._apply_w4a16_image(x, img[0], img[1], img[2], bias)
        return torch.ops.vllm._apply_gptq(
            x, layer.qweight, layer.scales, layer.qzeros, bias, layer.g_idx,
            layer.exllama_state == ExllamaState.READY, self.quant_config.weight_bits,
            self.quant_config.group_size, self.quant_config.desc_act)


direct_register_custom_op(op_name="_apply_gptq", op_func=linear.apply_gptq, mutates_args=[],
                          fake_impl=linear.apply_gptq_fake,
                          tags=(torch.Tag.needs_fixed_stride_order,))
try:   # shared with awq.py
    direct_register_custom_op(op_name="_apply_w4a16_image", op_func=linear.apply_w4a16_image, mutates_args=[],
                              fake_impl=linear.apply_w4a16_image_fake,
                              tags=(torch.Tag.needs_fixed_stride_order,))
except RuntimeError:
    pass
