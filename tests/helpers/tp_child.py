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
    from vllm_metax_amd import harness
    cfg = harness.ModelConfig.tiny(os.environ.get("TP_QUANT", "awq"))
    cfg.group_size = 64
    cfg.tp, cfg.tp_rank = world, rank
    model = harness.HotPathModel(cfg, 4, 96, device="cuda:0", seed=0, tp_group=group)
    model.collectives_always = os.environ.get("TP_FORCE_COLLECTIVES", "0") == "1"
    model.setup_decode(4, 40, 96)
    tok = torch.randint(0, cfg.vocab, (4, 40), generator=torch.Generator().manual_seed(5)).to("cuda:0")
    first = model.prefill(tok, [0, 1, 2, 3], 0)
    model.d_tokens.copy_(first)
    model.set_decode_lengths(torch.full((4,), 40, device="cuda:0"))
    out = [first.cpu().tolist()]
    for _ in range(6):
        model.decode_step(use_graph=use_graph)
        out.append(model.d_tokens.cpu().tolist())
    torch.cuda.synchronize()
    if rank == 0:
        with open(os.environ["TP_OUT"], "w") as f:
            json.dump({"tokens": out, "graph": model._graph not in (None, False), "graph_error": model.graph_error}, f)
    if group is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
