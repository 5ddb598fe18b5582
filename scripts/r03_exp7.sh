#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp7.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py tests/test_gpu_golden_backend.py -x -q 2>&1 | tail -n 4
for t in 1 0 1 0; do
for args in "--quant fp8" "--model llama-3-70b --tp-rank-of 8" "--quant fp8 --chunk-tokens 512 --steps 2"; do
  MI355X_PREPACK=$t timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print("MI355X_PREPACK=$t $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],1), d["config"]["hbm_weights_gb"])
PY
done
done
} > $O 2>&1
tail -n 30 $O
