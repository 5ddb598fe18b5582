"""Find the first op of an eager HotPathModel forward whose output is non-finite.

Wraps every function of vllm_metax_amd._custom_ops the harness calls; after each call all floating
tensors among the arguments / results are checked (fp8 tensors: NaN byte patterns 0x7f / 0xff).
Usage: python scripts/debug_fp8_nan.py [--quant fp8] [--model llama-3-8b] [--layers 2] [--tokens 8192]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vllm_metax_amd import _custom_ops as ops  # noqa: E402
from vllm_metax_amd import harness  # noqa: E402
from vllm_metax_amd.attention import backend  # noqa: E402

FIRST = []
CALLS = [0]


def bad(t: torch.Tensor):
    if not isinstance(t, torch.Tensor) or t.numel() == 0 or not t.is_cuda:
        return None
    if t.dtype == torch.float8_e4m3fn:
        b = t.view(torch.uint8)
        nb = int(((b & 0x7F) == 0x7F).sum())
        return f"{nb} NaN bytes" if nb else None
    if t.dtype in (torch.bfloat16, torch.float16, torch.float32):
        nf = int((~torch.isfinite(t)).sum())
        if nf:
            return f"{nf} non-finite of {t.numel()} (absmax finite part {t[torch.isfinite(t)].abs().max().item() if nf < t.numel() else float('nan')})"
    return None


def wrap(mod, name):
    fn = getattr(mod, name)

    def w(*a, **kw):
        pre = [bad(x) for x in a]
        r = fn(*a, **kw)
        torch.cuda.synchronize()
        CALLS[0] += 1
        outs = list(a) + (list(r) if isinstance(r, (tuple, list)) else [r])
        if isinstance(r, ops.PackedOperand):
            outs.append(r.data if hasattr(r, "data") else None)
        for i, x in enumerate(outs):
            if isinstance(x, ops.PackedOperand):
                continue
            msg = bad(x)
            if msg and (i >= len(pre) or pre[i] is None):
                rec = f"call #{CALLS[0]} {name}: tensor {i} shape {tuple(x.shape)} {x.dtype}: {msg}; " \
                      f"shapes {[tuple(y.shape) if isinstance(y, torch.Tensor) else y for y in a]}"
                if not FIRST:
                    print("FIRST NON-FINITE:", rec, flush=True)
                FIRST.append(rec)
        return r
    setattr(mod, name, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quant", default="fp8")
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--seqs", type=int, default=8)
    ap.add_argument("--qlen", type=int, default=1024)
    ap.add_argument("--decode", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--graph", action="store_true", help="no per-op checks; graph-replayed decode, logits checked per step")
    a = ap.parse_args()
    for name in ([] if a.graph else dir(ops)):
        f = getattr(ops, name)
        if callable(f) and not name.startswith("_") and getattr(f, "__module__", "") == ops.__name__ \
                and not isinstance(f, type):
            wrap(ops, name)
    cfg = harness.ModelConfig.tiny(a.quant) if a.model == "tiny" else harness.ModelConfig.llama3_8b(a.quant)
    cfg.layers = a.layers
    model = harness.HotPathModel(cfg, a.seqs, a.qlen + 64, device="cuda:0", seed=0)
    model.setup_decode(a.seqs, a.qlen, a.qlen + 64)
    tok = torch.randint(0, cfg.vocab, (a.seqs, a.qlen), device="cuda:0")
    first = torch.empty(a.seqs, dtype=torch.int64, device="cuda:0")
    for c0 in range(0, a.seqs, a.chunk):
        ids = list(range(c0, min(c0 + a.chunk, a.seqs)))
        first[ids[0]:ids[-1] + 1] = model.prefill(tok[ids[0]:ids[-1] + 1], ids, 0)
        lg = model.last_prefill_logits
        print("chunk", c0, "logits finite", bool(torch.isfinite(lg).all()), "absmax", lg.float().abs().max().item(),
              "records", len(FIRST), flush=True)
    print("prefill done, calls", CALLS[0], "first tokens", first[:4].tolist(), "non-finite records", len(FIRST), flush=True)
    model.d_tokens.copy_(first)
    model.set_decode_lengths(torch.full((a.seqs,), a.qlen, device="cuda:0"))
    for s in range(a.decode):
        model.decode_step(use_graph=a.graph)
        torch.cuda.synchronize()
        lg = model.last_logits
        print("decode step", s, "tokens", model.d_tokens[:4].tolist(), "records", len(FIRST),
              "logits finite", bool(torch.isfinite(lg).all()), "absmax", lg.float().abs().max().item(), flush=True)
    for r in FIRST[:12]:
        print(r)
    print("RESULT:", "finite" if not FIRST else f"{len(FIRST)} non-finite op results")


if __name__ == "__main__":
    main()
