#!/bin/bash
# final form of the in-place operands (fp8, k % 128 == 0; no weight image): tests and job records
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp16.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py tests/test_gpu_golden_backend.py tests/test_gpu_tp.py tests/test_gpu_patch_fused_layers.py -x -q 2>&1 | tail -n 4
for args in "--quant fp8" "--quant fp8 --chunk-tokens 512" "--model llama-3-70b --tp-rank-of 8" "--quant int8"; do
  timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
ps=d.get("plugin_surface") or {}
print("bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], d["config"]["hbm_weights_gb"], {k:(round(e["avg_launch_us"],2), round(e["frac"],3)) for k,e in t.items() if "gemm_large" in k}, ps.get("value"), (ps.get("with_register_patch") or {}).get("value"))
PY
  cp gpurun_out/r03c_tmp.json "gpurun_out/r03k_$(echo bench$args | tr -d ' -')".json
done
} > $O 2>&1
tail -n 12 $O
