"""Lazy environment-variable registry (ref: vllm_metax/envs.py:16-64, a module-level
__getattr__ over a dict of getters).  Only knobs that matter to the hot path are kept."""
import os
from typing import Any, Callable

environment_variables: dict[str, Callable[[], Any]] = {
    # decode attention: 0 = always split-KV (v2 semantics), 1 = allow the single-pass v1 kernel
    "MI355X_PA_ALLOW_V1": lambda: os.getenv("MI355X_PA_ALLOW_V1", "0") == "1",
    # maximum tokens per prefill chunk the backend assumes when sizing workspaces
    "MI355X_MAX_BATCHED_TOKENS": lambda: int(os.getenv("MI355X_MAX_BATCHED_TOKENS", "8192")),
    # keep a dequantised operand image of every int4 layer for the prefill GEMM (n*k*2 bytes per layer)
    # load-time bf16 operand image of the int4 weights for prefill-sized GEMMs (n * k * 2 bytes per layer: 14 GB for
    # Llama-3-8B of the 288 GB; allocated in process_weights_after_loading, i.e. before upstream vLLM profiles the
    # memory left for the KV pool).  On by default since round 3 ("0" keeps only the int4 words)
    "MI355X_PREPACK_WEIGHTS": lambda: os.getenv("MI355X_PREPACK_WEIGHTS", "1") == "1",
    # register_patch(): route vLLM's Llama / Qwen2 MLP and decoder-layer forwards through the fused entry points
    "MI355X_FUSED_LAYERS": lambda: os.getenv("MI355X_FUSED_LAYERS", "1") == "1",
    # fraction of the 288 GB HBM3E the KV pool may take (config sizing helper)
    "MI355X_KV_FRACTION": lambda: float(os.getenv("MI355X_KV_FRACTION", "0.9")),
}


def __getattr__(name: str):
    if name in environment_variables:
        return environment_variables[name]()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return list(environment_variables.keys())
