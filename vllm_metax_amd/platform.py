"""vLLM Platform for MI355X — mirrors the reference's MacaPlatformBase
(vllm_metax/platform.py:50-620) for the methods the hot path depends on:
import_kernels (:107-115), check_and_update_config (:117-221), get_attn_backend_cls
(:259-400), get_device_communicator_cls (:406-410), supported_quantization (:59-66),
supports_fp8 (:412-414, False there, True here), use_custom_allreduce (:416-418).

Upstream vLLM is imported lazily: without it the class still imports (base = object) so
that its pure logic is unit-testable in a vLLM-less container.
"""
from __future__ import annotations

import logging
from typing import Optional

import torch

logger = logging.getLogger(__name__)

try:  # pragma: no cover - only with upstream vLLM installed
    from vllm.platforms.interface import DeviceCapability, Platform, PlatformEnum
    _HAVE_VLLM = True
except Exception:  # noqa: BLE001
    Platform = object  # type: ignore[misc,assignment]
    PlatformEnum = None
    DeviceCapability = None
    _HAVE_VLLM = False

HBM_BYTES = 288 * 10 ** 9          # MI355X HBM3E capacity
BACKEND_CLS = "vllm_metax_amd.attention.backend.Mi355xPagedAttentionBackend"


class Mi355xPlatform(Platform):  # type: ignore[misc,valid-type]
    _enum = PlatformEnum.OOT if _HAVE_VLLM else None
    device_name: str = "cuda"        # torch-ROCm exposes HIP devices as "cuda"
    device_type: str = "cuda"
    dispatch_key: str = "CUDA"
    ray_device_key: str = "GPU"
    dist_backend: str = "nccl"       # == RCCL on ROCm (ref: platform.py:56)
    device_control_env_var: str = "HIP_VISIBLE_DEVICES"

    # ref: platform.py:59-66 — awq / gptq like the reference, plus fp8 (new capability)
    supported_quantization: list[str] = ["awq", "gptq", "fp8", "compressed-tensors"]

    # ------------------------------------------------------------------ kernels
    @classmethod
    def import_kernels(cls) -> None:
        """ref: platform.py:107-115 — importing the extension registers torch.ops._C*."""
        try:
            import vllm_metax_amd._C  # noqa: F401
        except ImportError as e:
            # unlike the reference this is fatal on a GPU box: there is no fallback path
            raise ImportError(
                "vllm_metax_amd._C is not built; run `python -m vllm_metax_amd.build`") from e

    # ------------------------------------------------------------------- config
    @classmethod
    def check_and_update_config(cls, vllm_config) -> None:
        """ref: platform.py:117-221 — worker class, default block size 16, cascade off.
        (The MLA / DeepEP branches of the reference are out of scope.)"""
        parallel_config = vllm_config.parallel_config
        if getattr(parallel_config, "worker_cls", None) == "auto":
            parallel_config.worker_cls = "vllm.v1.worker.gpu_worker.Worker"
        cache_config = vllm_config.cache_config
        if cache_config is not None and getattr(cache_config, "block_size", None) is None:
            cache_config.block_size = 16
        model_config = getattr(vllm_config, "model_config", None)
        if model_config is not None:
            model_config.disable_cascade_attn = True

    @classmethod
    def get_attn_backend_cls(cls, selected_backend=None, head_size: int = 128,
                             dtype: torch.dtype = torch.bfloat16, kv_cache_dtype: str = "auto",
                             block_size: int = 16, use_v1: bool = True, use_mla: bool = False,
                             *args, **kwargs) -> str:
        """ref: platform.py:259-400 — one backend here: paged attention over the x-split cache."""
        if use_mla:
            raise NotImplementedError("MLA backends are out of scope of the MI355X hot path")
        if kv_cache_dtype not in (None, "auto"):
            # ref: csrc/quantization/fp8/metax/quant_utils.cuh:29-42 — only "auto"
            raise ValueError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
        return BACKEND_CLS

    @classmethod
    def get_device_communicator_cls(cls) -> str:
        """ref: platform.py:406-410 — upstream CudaCommunicator; pynccl binds librccl on ROCm."""
        return "vllm.distributed.device_communicators.cuda_communicator.CudaCommunicator"

    @classmethod
    def supports_fp8(cls) -> bool:
        """ref: platform.py:412-414 returns False; gfx950 has OCP e4m3 MFMA."""
        return True

    @classmethod
    def use_custom_allreduce(cls) -> bool:
        """ref: platform.py:416-418 — off: RCCL all-reduce over xGMI only (north_star)."""
        return False

    @classmethod
    def is_cuda_alike(cls) -> bool:
        return True

    # ------------------------------------------------------------------- device
    @classmethod
    def get_device_name(cls, device_id: int = 0) -> str:
        return torch.cuda.get_device_name(device_id)

    @classmethod
    def get_device_total_memory(cls, device_id: int = 0) -> int:
        return torch.cuda.get_device_properties(device_id).total_memory

    @classmethod
    def get_device_capability(cls, device_id: int = 0):
        major, minor = torch.cuda.get_device_capability(device_id)
        return DeviceCapability(major=major, minor=minor) if _HAVE_VLLM else (major, minor)

    @classmethod
    def check_if_supports_dtype(cls, dtype: torch.dtype) -> None:
        """ref: platform.py:432-435 raises on fp8; every hot-path dtype is fine here."""
        return None

    # ------------------------------------------------------- KV sizing (288 GB HBM3E)
    @staticmethod
    def kv_bytes_per_token(num_layers: int, num_kv_heads: int, head_size: int,
                           dtype_bytes: int = 2) -> int:
        return 2 * num_layers * num_kv_heads * head_size * dtype_bytes

    @classmethod
    def kv_pool_tokens(cls, weight_bytes: int, num_layers: int, num_kv_heads: int, head_size: int,
                       utilization: float = 0.9, hbm_bytes: int = HBM_BYTES,
                       block_size: int = 16, dtype_bytes: int = 2) -> int:
        """Tokens that fit in the paged KV pool of one GPU: (HBM * utilization - weights) /
        bytes per token, rounded down to whole blocks (70B TP=8 -> 40 960 B/token)."""
        budget = int(hbm_bytes * utilization) - int(weight_bytes)
        if budget <= 0:
            return 0
        per_tok = cls.kv_bytes_per_token(num_layers, num_kv_heads, head_size, dtype_bytes)
        return (budget // per_tok) // block_size * block_size
