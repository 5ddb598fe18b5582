#!/bin/bash
# prefill attention: 64 query rows per workgroup for grids below two workgroups per CU (MI355X_PF_QT=2 = always 128)
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp25.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_prefill_attention.py tests/test_gpu_fp8_kv.py tests/test_gpu_golden_backend.py -x -q 2>&1 | tail -n 3
MI355X_PF_QT=1 timeout -k 10 900 python -m pytest tests/test_gpu_prefill_attention.py -x -q 2>&1 | tail -n 2
MI355X_PF_QT=2 timeout -k 10 900 python -m pytest tests/test_gpu_prefill_attention.py -x -q 2>&1 | tail -n 2
for qt in 2 0 2 0; do
  for args in "--chunk-tokens 512" "--model llama-3-70b --tp-rank-of 8 --chunk-tokens 2048" "--quant fp8 --chunk-tokens 512"; do
    MI355X_PF_QT=$qt timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface --steps 2 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("qt=$qt bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:round(e["avg_launch_us"],2) for k,e in t.items() if "prefill_attention" in k})
PY
  done
done
} > $O 2>&1
tail -n 20 $O
