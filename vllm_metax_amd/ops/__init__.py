"""Out-of-tree CustomOp overrides (ref: vllm_metax/ops/layernorm.py:5-14,
ops/rotary_embedding.py:5-8): route RMSNorm / RotaryEmbedding `forward_oot` to vLLM's
`forward_cuda`, which calls torch.ops._C.rms_norm / fused_add_rms_norm / rotary_embedding —
the ops registered by this plugin's _C.so."""


def register() -> None:
    from vllm.model_executor.layers.layernorm import RMSNorm
    from vllm.model_executor.layers.rotary_embedding import RotaryEmbedding

    @RMSNorm.register_oot
    class Mi355xRMSNorm(RMSNorm):  # noqa: F811
        def forward_oot(self, *args, **kwargs):
            return self.forward_cuda(*args, **kwargs)

    @RotaryEmbedding.register_oot
    class Mi355xRotaryEmbedding(RotaryEmbedding):  # noqa: F811
        def forward_oot(self, *args, **kwargs):
            return self.forward_cuda(*args, **kwargs)

    try:  # SiluAndMul is §8f "next-1"; the kernel exists, so route it too when available
        from vllm.model_executor.layers.activation import SiluAndMul

        @SiluAndMul.register_oot
        class Mi355xSiluAndMul(SiluAndMul):  # noqa: F811
            def forward_oot(self, *args, **kwargs):
                return self.forward_cuda(*args, **kwargs)
    except Exception:  # noqa: BLE001
        pass
