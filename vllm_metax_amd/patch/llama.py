"""vLLM-facing side of the fused decoder-layer forwards: replaces the `forward` of upstream's Llama / Qwen2 MLP
and decoder-layer classes with versions that go through `fused_layers` whenever the layer's linears are served by
this plugin's AWQ / GPTQ methods (ref mechanism: vllm_metax/patch/**, module-attribute patches applied by
register_patch, vllm_metax/__init__.py:92-93).  Anything the fused paths do not cover (bias, act-order, other quant
methods, LoRA-wrapped layers) falls through to the original forward, so the patch is safe to install always.
Imports upstream vLLM: only reachable when vLLM is installed."""
from __future__ import annotations

from typing import Optional

from . import fused_layers as F
from ..quant_config import linear as qlinear


def describe(layer) -> Optional[F.W4Linear]:
    """A vLLM LinearBase whose quant method is this plugin's AWQ / GPTQ method -> W4Linear; else None."""
    qm = getattr(layer, "quant_method", None)
    name = type(qm).__name__ if qm is not None else ""
    mod = type(qm).__module__ if qm is not None else ""
    if not mod.startswith("vllm_metax_amd.quant_config") or getattr(layer, "skip_bias_add", False):
        return None
    cfg = qm.quant_config
    if name == "AWQLinearMethod":
        kind, g_idx = "awq", None
    elif name == "GPTQLinearMethod" and cfg.weight_bits == 4 and not cfg.desc_act:
        kind, g_idx = "gptq", layer.g_idx.data
    else:
        return None
    return F.W4Linear(kind, layer.qweight.data, layer.qzeros.data, layer.scales.data, cfg.group_size, g_idx,
                      getattr(layer, qlinear.IMAGE_ATTR, None), getattr(layer, "bias", None))


def _mlp_parts(mlp):
    gu, dn = describe(getattr(mlp, "gate_up_proj", None)), describe(getattr(mlp, "down_proj", None))
    return (gu, dn) if F.fusable(gu, dn) else None


def _all_reduce_if_needed(down_proj, x):
    if getattr(down_proj, "reduce_results", False) and getattr(down_proj, "tp_size", 1) > 1:
        from vllm.distributed import tensor_model_parallel_all_reduce
        return tensor_model_parallel_all_reduce(x)
    return x


def apply() -> list:
    patched = []
    targets = []
    try:
        from vllm.model_executor.models import llama
        targets.append((llama.LlamaMLP, llama.LlamaDecoderLayer))
    except Exception:  # noqa: BLE001 - model class not present in this vLLM
        pass
    try:
        from vllm.model_executor.models import qwen2
        targets.append((qwen2.Qwen2MLP, qwen2.Qwen2DecoderLayer))
    except Exception:  # noqa: BLE001
        pass
    for mlp_cls, layer_cls in targets:
        if getattr(mlp_cls, "_mi355x_patched", False):
            continue
        orig_mlp, orig_layer = mlp_cls.forward, layer_cls.forward

        def mlp_forward(self, x, _orig=orig_mlp):
            parts = _mlp_parts(self)
            if parts is None or x.dim() != 2:
                return _orig(self, x)
            return _all_reduce_if_needed(self.down_proj, F.fused_mlp(x, *parts))

        def layer_forward(self, positions, hidden_states, residual, _orig=orig_layer):
            parts = _mlp_parts(self.mlp)
            if parts is None or hidden_states.dim() != 2:
                return _orig(self, positions, hidden_states, residual)
            # attention block as upstream (its fusions live in the attention backend), then norm + MLP fused
            if residual is None:
                residual = hidden_states
                hidden_states = self.input_layernorm(hidden_states)
            else:
                hidden_states, residual = self.input_layernorm(hidden_states, residual)
            hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states)
            norm = self.post_attention_layernorm
            out, residual = F.fused_norm_mlp(hidden_states, residual, norm.weight.data, norm.variance_epsilon, *parts)
            return _all_reduce_if_needed(self.mlp.down_proj, out), residual

        mlp_cls.forward, layer_cls.forward = mlp_forward, layer_forward
        mlp_cls._mi355x_patched = True
        patched += [mlp_cls.__name__, layer_cls.__name__]
    return patched
