"""Child process of tests/test_gpu_tp.py: one tensor-parallel rank of the tiny harness model on the GPU.
Rendezvous over 127.0.0.1 (env RANK / WORLD_SIZE / MASTER_PORT), backend from TP_BACKEND (gloo: several
ranks may share cuda:0; nccl: RCCL), decode with or without the HIP graph, tokens written to TP_OUT."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("TP_BACKEND", "gloo")
    use_graph = os.environ.get("TP_GRAPH", "0") == "1"
    torch.cuda.set_device(0)
    group = None
    if backend != "none":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    if os.environ.get("TP_SWITCHES_OFF", "0") == "1":
        # every cross-op fusion of the harness off: the plain op sequence (read at import / construction time)
        for k in ("MI355X_PACKED_SILU", "MI355X_PREPACK", "MI355X_FUSE_GREEDY", "MI355X_FUSE_ATTN_QKV",
                  "MI355X_NORM_IMAGE", "MI355X_FUSE_NORM_QUANT", "MI355X_FP8_DEFER"):
            os.environ[k] = "0"
    from vllm_metax_amd import harness
    quant = os.environ.get("TP_QUANT", "awq")
    if os.environ.get("TP_MODEL", "tiny") == "llama":
        # Llama-3-8B width (hidden 4096, 32 / 8 heads x 128, FFN 14336), 2 layers, >= 1024 prefill tokens: the
        # shapes at which the fused decode attention, the operand-image norms / attention and the rotary-in-cache
        # write are selected (never exercised under sharding or graph capture by the tiny model)
        cfg = harness.ModelConfig.llama_geometry(quant, layers=2, vocab=4096)
        n_seq, q_len, max_len = 2, 1024, 1056
    else:
        cfg = harness.ModelConfig.tiny(quant)
        cfg.group_size = 64
        n_seq, q_len, max_len = 4, 40, 96
    cfg.tp, cfg.tp_rank = world, rank
    model = harness.HotPathModel(cfg, n_seq, max_len, device="cuda:0", seed=0, tp_group=group)
    model.collectives_always = os.environ.get("TP_FORCE_COLLECTIVES", "0") == "1"
    model.setup_decode(n_seq, q_len, max_len)
    tok = torch.randint(0, cfg.vocab, (n_seq, q_len), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    first = model.prefill(tok, list(range(n_seq)), 0)
    model.d_tokens.copy_(first)
    model.set_decode_lengths(torch.full((n_seq,), q_len, device="cuda:0"))
    out = [first.cpu().tolist()]
    logits = [model.last_prefill_logits.float().cpu()]
    for _ in range(6):
        model.decode_step(use_graph=use_graph)
        out.append(model.d_tokens.cpu().tolist())
        logits.append(model.last_logits.float().cpu().clone())
    torch.cuda.synchronize()
    if rank == 0:
        with open(os.environ["TP_OUT"], "w") as f:
            json.dump({"tokens": out, "graph": model._graph not in (None, False), "graph_error": model.graph_error,
                       "finite": all(bool(torch.isfinite(l).all()) for l in logits),
                       "prefill_logits": logits[0].tolist()}, f)
    if group is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
