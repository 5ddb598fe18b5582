"""MI355X-native vLLM hardware plugin (hot path only) — see DESIGN.md / INTEGRATION.md.

Entry points mirror the reference plugin (vllm_metax/__init__.py:86-121, registered in its
pyproject.toml:43-50):

    vllm.platform_plugins : register()                -> "vllm_metax_amd.platform.Mi355xPlatform"
    vllm.general_plugins  : register_ops(), register_quant_configs()

All of them import upstream vLLM lazily; the package itself (C-ABI loader, op wrappers,
measurement harness) imports without vLLM.
"""
from __future__ import annotations

__version__ = "0.1.0"


def register():
    """Platform plugin entry point (ref: vllm_metax/__init__.py:86-89)."""
    return "vllm_metax_amd.platform.Mi355xPlatform"


def register_patch():
    """ref: vllm_metax/__init__.py:92-93 (`import vllm_metax.patch`).  The reference's patches rename MACA
    symbols (mccl*, mc* runtime) — not needed on ROCm, upstream already binds librccl / libamdhip64.  This
    plugin's patches route upstream's Llama / Qwen2 MLP and decoder-layer forwards through the fused MI355X
    entry points (patch/fused_layers.py), falling through to the original forward for anything not covered.
    MI355X_FUSED_LAYERS=0 leaves vLLM unpatched (plain op surface)."""
    from . import envs
    if not envs.MI355X_FUSED_LAYERS:
        return []
    from . import patch
    return patch.apply()


def register_ops():
    """ref: vllm_metax/__init__.py:96-98 — route RMSNorm / RotaryEmbedding OOT dispatch."""
    from . import ops  # noqa: F401
    ops.register()


def register_quant_configs():
    """ref: vllm_metax/__init__.py:107-121 — override "awq" / "gptq" (+ disable the Marlin
    auto-upgrade) so that linear layers call this plugin's w4a16 kernels."""
    from . import quant_config  # noqa: F401
    quant_config.register()


def register_model():
    """ref: vllm_metax/__init__.py:101-104 — forked model definitions are out of scope
    (configs use upstream Llama / Qwen2 / OPT)."""
    return None
