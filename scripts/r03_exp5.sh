#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp5.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_golden_backend.py tests/test_gpu_fp8_gemm.py -x -q 2>&1 | tail -n 5
for sw in 1 0 1 0; do
for args in "--model llama-3-70b --tp-rank-of 8" "--quant fp8"; do
  MI355X_FP8_DEFER=$sw timeout -k 10 300 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print("MI355X_FP8_DEFER=$sw $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
PY
  cp gpurun_out/r03c_tmp.json "gpurun_out/r03d_defer${sw}_$(echo $args | tr -d ' -')".json
done
done
} > $O 2>&1
tail -n 30 $O
