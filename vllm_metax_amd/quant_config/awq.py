"""AWQ config / linear method for vLLM (ref: vllm_metax/quant_config/awq.py:25-168).
Imported only by register_quant_configs(), i.e. only when upstream vLLM is present."""
from typing import Optional

import torch
from vllm.model_executor.layers.linear import LinearBase, UnquantizedLinearMethod
from vllm.model_executor.layers.quantization.awq import AWQConfig
from vllm.model_executor.layers.quantization.awq import AWQLinearMethod as _AWQLinearMethod
from vllm.model_executor.layers.quantization.awq import is_layer_skipped
from vllm.utils.torch_utils import direct_register_custom_op

from . import linear
from .hook_register import register_quantization_config


@register_quantization_config("awq")
class Mi355xAWQConfig(AWQConfig):
    def get_supported_act_dtypes(self):
        return [torch.half, torch.bfloat16]

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if isinstance(layer, LinearBase):
            if is_layer_skipped(prefix, self.modules_to_not_convert, self.packed_modules_mapping,
                                skip_with_substr=True):
                return UnquantizedLinearMethod()
            return AWQLinearMethod(self)
        return None   # fused-MoE layers: out of scope


class AWQLinearMethod(_AWQLinearMethod):
    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        layer.qzeros = torch.nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = torch.nn.Parameter(layer.scales.data, requires_grad=False)
        layer.qweight = torch.nn.Parameter(
            linear.awq_process_weights(layer.qweight.data, self.quant_config.group_size),
            requires_grad=False)
        if self.quant_config.group_size % 32 == 0:
            linear.attach_prefill_image(layer, False)

    def apply(self, layer: torch.nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        img = linear.layer_image(layer, x)
        if img is not None:
            return torch.ops.vllm._apply_w4a16_image(x, img[0], img[1], img[2], bias)
        return torch.ops.vllm._apply_awq(x, layer.qweight, layer.scales, layer.qzeros, bias,
                                         self.quant_config.pack_factor,
                                         self.quant_config.group_size)


direct_register_custom_op(op_name="_apply_awq", op_func=linear.apply_awq, mutates_args=[],
                          fake_impl=linear.apply_awq_fake,
                          tags=(torch.Tag.needs_fixed_stride_order,))
try:   # shared with gptq.py: whichever module is imported first registers it
    direct_register_custom_op(op_name="_apply_w4a16_image", op_func=linear.apply_w4a16_image, mutates_args=[],
                              fake_impl=linear.apply_w4a16_image_fake,
                              tags=(torch.Tag.needs_fixed_stride_order,))
except RuntimeError:
    pass
