#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp9.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_cache_norm_rotary.py tests/test_gpu_ref_fixtures.py tests/test_gpu_fp8_gemm.py tests/test_gpu_paged_attention.py tests/test_gpu_golden_backend.py tests/test_gpu_fp8_kv.py -x -q 2>&1 | tail -n 6
for args in "--quant fp8" "--model llama-3-70b --tp-rank-of 8"; do
  timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print("bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
for e in [d["roofline"]]+d["roofline_other"][:12]:
    print("   ", e["kernel"], round(e["avg_launch_us"],2), round(e["frac"],3), e.get("job_share"))
PY
done
} > $O 2>&1
tail -n 50 $O
