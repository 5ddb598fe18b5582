#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp3b.txt
{
for m in "llama-3-70b" "qwen2-72b"; do
  timeout -k 10 300 python bench.py --model $m --tp-rank-of 8 --skip-cpu 2> gpurun_out/r03b_$m.err | tail -n 1 > gpurun_out/r03b_rank_of_8_$m.json
  tail -n 5 gpurun_out/r03b_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03b_rank_of_8_$m.json"))
print("$m", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
for e in [d["roofline"]]+d["roofline_other"][:9]:
    print("   ", e["kernel"], round(e["avg_launch_us"],2), round(e["frac"],3), e.get("job_share"))
PY
done
timeout -k 10 300 python bench.py --skip-cpu --no-plugin-surface 2>/dev/null | tail -n 1 > gpurun_out/r03b_bench_default.json
python -c "
import json
d=json.load(open('gpurun_out/r03b_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['ttft_p50_ms'])"
timeout -k 10 300 python bench.py --quant fp8 --skip-cpu --no-plugin-surface 2>/dev/null | tail -n 1 > gpurun_out/r03b_bench_fp8.json
python -c "
import json
d=json.load(open('gpurun_out/r03b_bench_fp8.json')); print('fp8', d['value'], d['ms_per_step'], d['ttft_p50_ms'])"
} > $O 2>&1
tail -n 70 $O
