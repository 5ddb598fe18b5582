#!/bin/bash
# round-3 call: 8-heads-per-workgroup decode attention + caller-chosen partition size (TP = 8 rank shapes), norm
# launcher without the spilling 4-chunk variant: tests, micro-benchmark, the two TP = 8 rank jobs, the default job.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp3.txt
set -o pipefail
{
timeout -k 10 900 python -m pytest tests/test_gpu_paged_attention.py tests/test_gpu_cache_norm_rotary.py tests/test_gpu_ref_fixtures.py tests/test_gpu_fp8_kv.py tests/test_gpu_golden_backend.py tests/test_gpu_tp.py -x -q 2>&1 | tail -n 15 || exit 1
echo "== attention at the TP = 8 rank head shape"
python scripts/bench_attn_tp8.py 2>&1 | grep -v amdgpu
python scripts/bench_attn_tp8.py 1088 16 2>&1 | grep -v amdgpu
echo "== headline attention shape (unchanged path)"
python scripts/bench_attn.py 2>&1 | grep -v amdgpu
for m in "llama-3-70b" "qwen2-72b"; do
  timeout -k 10 300 python bench.py --model $m --tp-rank-of 8 --skip-cpu 2>/dev/null | tail -n 1 > gpurun_out/r03b_rank_of_8_$m.json || exit 2
  python - <<PY
import json
d=json.load(open("gpurun_out/r03b_rank_of_8_$m.json"))
print("$m", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
for e in [d["roofline"]]+d["roofline_other"][:8]:
    print("   ", e["kernel"], round(e["avg_launch_us"],2), round(e["frac"],3), e.get("job_share"))
PY
done
timeout -k 10 300 python bench.py --skip-cpu --no-plugin-surface 2>/dev/null | tail -n 1 > gpurun_out/r03b_bench_default.json || exit 3
python -c "
import json
d=json.load(open('gpurun_out/r03b_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['ttft_p50_ms'])"
} > $O 2>&1
tail -n 70 $O
