#!/bin/bash
# int8 prefill GEMM: the 4-stage ping-pong ring on operand images (default) against the two-slice stage with both operands
# in place (variants/libi8wide.so, MI355X_PREPACK=0 so that no image is built)
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp20.txt
{
MI355X_HOTPATH_LIB=variants/libi8wide.so timeout -k 10 600 python -m pytest tests/test_gpu_int8.py -x -q 2>&1 | tail -n 2
for mode in default wide default wide; do
  for args in "--quant int8" "--quant int8 --chunk-tokens 512"; do
    if [ $mode = wide ]; then export MI355X_HOTPATH_LIB=variants/libi8wide.so MI355X_PREPACK=0; else unset MI355X_HOTPATH_LIB MI355X_PREPACK; fi
    timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("$mode bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], d["config"]["hbm_weights_gb"]["total"], {k:round(e["avg_launch_us"],2) for k,e in t.items() if "gemm_large" in k})
PY
  done
done
} > $O 2>&1
tail -n 12 $O
