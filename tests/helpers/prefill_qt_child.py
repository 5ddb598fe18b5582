"""Child process of tests/test_gpu_prefill_attention.py::test_prefill_query_tile_shapes_agree: one paged prefill
attention call under the MI355X_PF_QT the parent set (the launcher reads it once per process), output saved to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.util import make_kv_cache_x  # noqa: E402
from vllm_metax_amd import _custom_ops as ops  # noqa: E402


def main():
    out_path, kind = sys.argv[1], sys.argv[2]
    q_lens, seq_lens = [300, 64, 129], [813, 64, 400]
    H, KVH, D, bs, dtype = 8, 2, 128, 16, torch.bfloat16
    torch.manual_seed(7)
    S = len(q_lens)
    max_blocks = (max(seq_lens) + bs - 1) // bs
    nb = S * max_blocks + 3
    kc, vc = make_kv_cache_x(nb, bs, KVH, D, dtype, 7)
    perm = torch.randperm(nb)
    bt = perm[:S * max_blocks].reshape(S, max_blocks).to(torch.int32)
    cu = torch.zeros(S + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(q_lens).cumsum(0)
    T = int(cu[-1])
    q = (torch.randn(T, H, D) * 0.5).to(dtype)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    d = torch.device("cuda:0")
    out = torch.full((T, H, D), float("nan"), dtype=dtype, device=d)
    window, softcap = (200, 30.0) if kind == "opts" else (None, None)
    ops.paged_prefill_attention(out, q.to(d), kc.to(d), vc.to(d), KVH, D ** -0.5, bt.to(d), sl.to(d), cu.to(d),
                                max(q_lens), bs, "auto", None, None, window, softcap, None)
    torch.cuda.synchronize()
    np.save(out_path, out.view(torch.int16).cpu().numpy())


if __name__ == "__main__":
    main()
