"""Quantisation-config overrides (ref: vllm_metax/__init__.py:107-121 and
vllm_metax/quant_config/{awq,gptq,awq_marlin,gptq_marlin}.py)."""


def register() -> None:
    from . import hook_register  # noqa: F401  (lets built-in names be overridden)
    from . import awq, gptq, marlin_off  # noqa: F401
