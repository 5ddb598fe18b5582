"""ref: vllm_metax/quant_config/awq_marlin.py:14-19 and gptq_marlin.py:14-19 — vLLM upgrades
"awq"/"gptq" checkpoints to the Marlin kernels when it thinks the GPU supports them;
`override_quantization_method -> None` keeps them on this plugin's w4a16 path."""
from vllm.model_executor.layers.quantization.awq_marlin import AWQMarlinConfig
from vllm.model_executor.layers.quantization.gptq_marlin import GPTQMarlinConfig

from .hook_register import register_quantization_config


@register_quantization_config("awq_marlin")
class Mi355xAWQMarlinConfig(AWQMarlinConfig):
    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant):
        return None


@register_quantization_config("gptq_marlin")
class Mi355xGPTQMarlinConfig(GPTQMarlinConfig):
    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant):
        return None
