#!/bin/bash
# fp8 prefill kernel: persistent workgroups with the next item's first stage in flight (MI355X_F8_PERSIST=0: one workgroup per item)
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp28.txt
{
timeout -k 10 900 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -n 3
for ps in 0 1 0 1; do
  echo "== MI355X_F8_PERSIST=$ps"
  MI355X_F8_PERSIST=$ps timeout -k 10 300 python scripts/bench_scaled_mm.py --fp8 576 2048 8192 2>&1 | grep "^fp8"
  MI355X_F8_PERSIST=$ps timeout -k 10 300 python scripts/bench_scaled_mm.py --fp8 --70b-rank 8192 2>&1 | grep "^fp8"
done
for ps in 0 1 0 1; do
  for args in "--quant fp8" "--model llama-3-70b --tp-rank-of 8"; do
    MI355X_F8_PERSIST=$ps timeout -k 10 400 python bench.py $args --skip-cpu --no-plugin-surface 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("persist=$ps bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:(round(e["avg_launch_us"],2), round(e["frac"],3)) for k,e in t.items() if "gemm_large" in k})
PY
  done
done
} > $O 2>&1
grep -E "passed|failed|^==|total|^persist" $O
