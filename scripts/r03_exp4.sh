#!/bin/bash
# round-3 call: fp8 decode with the K splits reduced by the consumers — tests, then the fp8 jobs with the switch on / off.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp4.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_paged_attention.py tests/test_gpu_golden_backend.py tests/test_gpu_cache_norm_rotary.py tests/test_gpu_tp.py -x -q 2>&1 | tail -n 25
for sw in 1 0; do
for args in "--model llama-3-70b --tp-rank-of 8" "--quant fp8"; do
  echo "== MI355X_FP8_DEFER=$sw bench.py $args"
  MI355X_FP8_DEFER=$sw timeout -k 10 300 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  tail -n 3 gpurun_out/r03c.err | cut -c1-300
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print(d["value"], d["ms_per_step"], d["ttft_p50_ms"])
for e in [d["roofline"]]+d["roofline_other"][:12]:
    print("   ", e["kernel"], round(e["avg_launch_us"],2), round(e["frac"],3), e.get("job_share"))
PY
  cp gpurun_out/r03c_tmp.json "gpurun_out/r03c_defer${sw}_$(echo $args | tr -d ' -')".json
done
done
} > $O 2>&1
tail -n 120 $O
