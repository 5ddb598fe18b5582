#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp11.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_cache_norm_rotary.py tests/test_gpu_fp8_gemm.py tests/test_gpu_golden_backend.py tests/test_gpu_tp.py tests/test_gpu_patch_fused_layers.py -x -q 2>&1 | tail -n 4
for args in "--model llama-3-70b --tp-rank-of 8" "--model qwen2-72b --tp-rank-of 8" "--quant fp8" ""; do
  timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
names=[k for k in t if "norm" in k and "decode" not in k]
ps=d.get("plugin_surface") or {}
print("bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:(round(t[k]["avg_launch_us"],1), round(t[k]["frac"],3)) for k in names}, ps.get("value"), (ps.get("with_register_patch") or {}).get("value"))
PY
  cp gpurun_out/r03c_tmp.json "gpurun_out/r03i_$(echo bench$args | tr -d ' -')".json
done
} > $O 2>&1
tail -n 12 $O
