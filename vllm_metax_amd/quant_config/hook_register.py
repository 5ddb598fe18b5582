"""ref: vllm_metax/patch/model_executor/hook_register.py:12-61 — vLLM refuses to re-register a
built-in quantisation name ("awq", "gptq"); the reference swaps in a registrar that allows it.
Same mechanism here."""
import logging

from vllm.model_executor.layers import quantization as _q
from vllm.model_executor.layers.quantization import (_CUSTOMIZED_METHOD_TO_QUANT_CONFIG,
                                                     QUANTIZATION_METHODS, QuantizationConfig)

logger = logging.getLogger(__name__)


def register_quantization_config(quantization: str):
    def _wrapper(cls):
        if not issubclass(cls, QuantizationConfig):
            raise ValueError("The quantization config must be a subclass of `QuantizationConfig`.")
        if quantization in QUANTIZATION_METHODS:
            logger.warning("quantization method %s is overridden by %s", quantization, cls)
        else:
            QUANTIZATION_METHODS.append(quantization)
        _CUSTOMIZED_METHOD_TO_QUANT_CONFIG[quantization] = cls
        return cls
    return _wrapper


_q.register_quantization_config = register_quantization_config
