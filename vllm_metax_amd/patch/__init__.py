"""Monkey patches the plugin applies to upstream vLLM (ref: vllm_metax/patch/__init__.py:3-8, installed by
register_patch(), vllm_metax/__init__.py:92-93).  The reference patches MACA symbol names and model-executor
details; here the patches are what brings the MI355X cross-op fusions under vLLM's own Llama / Qwen2 layers:
`fused_layers` holds the vLLM-independent core (testable on a GPU box without vLLM), `llama` the thin
vLLM-facing forward replacements."""


def apply() -> list:
    """Install every patch whose target imports; returns the names of the patched classes."""
    from . import llama
    return llama.apply()
