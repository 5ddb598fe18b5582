"""Decode-only timing of the bench model (Llama-3-8B AWQ, batch 64, context 1024 + step): 127 graph-replayed
steps per round, several rounds; honours the harness switches (MI355X_FUSE_ATTN_QKV, MI355X_FUSE_GREEDY, ...).
usage: python scripts/bench_decode.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_metax_amd import harness
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, Lin, Lout = 64, 1024, 128
cfg = harness.ModelConfig.llama3_8b("awq")
m = harness.HotPathModel(cfg, B, Lin + Lout, device="cuda:0", seed=0)
m.setup_decode(B, Lin, Lin + Lout)
best = 1e9
for r in range(rounds + 1):
    m.d_tokens.random_(0, cfg.vocab)
    m.set_decode_lengths(torch.full((B,), Lin, device=m.device))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(Lout - 1):
        m.decode_step()
    b.record(); torch.cuda.synchronize()
    if r:   # round 0 captures the graph
        best = min(best, a.elapsed_time(b))
        print(f"round {r}: {a.elapsed_time(b):8.2f} ms for {Lout - 1} steps = {a.elapsed_time(b) / (Lout - 1) * 1e3:7.1f} us per step", flush=True)
print(f"best {best:.2f} ms  ({best / (Lout - 1) * 1e3:.1f} us per step)")
