import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vllm_metax_amd import harness, _custom_ops as ops
torch.manual_seed(0)
cfg = harness.ModelConfig.tiny("awq")
m = harness.HotPathModel(cfg, 3, 64, device="cuda:0", seed=0)
m.setup_decode(3, 40, 64)
tokens = torch.randint(0, cfg.vocab, (3, 40), device=m.device)
first = m.prefill(tokens, [0, 1, 2], 0)
print("first", first.tolist())
m.d_tokens.copy_(first)
m.set_decode_lengths(torch.full((3,), 40, device=m.device))
# manual decode body with checks
slots = m._slots(m.d_seq_ids, m.d_positions)
print("slots", slots.tolist(), "pos", m.d_positions.tolist(), "seq_lens", m.d_seq_lens.tolist())
x = m.embed[m.d_tokens]
residual = None
def attn_fn(i, q3):
    out = torch.empty_like(q3)
    print(" q3", q3.shape, q3.stride(), "out", out.stride(), "qnan", torch.isnan(q3.float()).any().item())
    ops.paged_attention_v2(out, m.d_es, m.d_ml, m.d_tmp, q3, m.k_cache[i], m.v_cache[i], m.layers[i].kv_heads,
                           m.scale, m.d_bt, m.d_seq_lens, m.BLOCK, m.d_max_seq_len, None, "auto")
    torch.cuda.synchronize()
    print(" attn out nan", torch.isnan(out.float()).any().item(), out.float().abs().max().item(),
          "es", m.d_es.flatten()[:4].tolist(), "ml", m.d_ml.flatten()[:4].tolist())
    # v1 for comparison
    o1 = torch.empty(q3.shape, dtype=q3.dtype, device=q3.device)
    ops.paged_attention_v1(o1, q3, m.k_cache[i], m.v_cache[i], m.layers[i].kv_heads, m.scale, m.d_bt,
                           m.d_seq_lens, m.BLOCK, m.d_max_seq_len, None, "auto")
    print(" v1 vs v2 maxdiff", (o1.float() - out.float()).abs().max().item())
    return out
for i in range(cfg.layers):
    x, residual = m._layer(i, x, residual, m.d_positions, slots, attn_fn)
    print("layer", i, "x nan", torch.isnan(x.float()).any().item(), x.float().abs().max().item())
nxt = m._logits_argmax(x, residual)
print("next", nxt.tolist())
