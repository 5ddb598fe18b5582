#!/bin/bash
# persistent fp8 prefill kernel at chunked-prefill sizes: A/B on one box
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp29.txt
{
for ps in 0 1 0 1; do
  echo "== MI355X_F8_PERSIST=$ps"
  MI355X_F8_PERSIST=$ps timeout -k 10 300 python scripts/bench_scaled_mm.py --fp8 576 2>&1 | grep "^fp8"
  MI355X_F8_PERSIST=$ps timeout -k 10 400 python bench.py --quant fp8 --chunk-tokens 512 --skip-cpu --no-plugin-surface --steps 2 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("persist=$ps chunk512 fp8:", d["value"], d["ms_per_step"], {k:(round(e["avg_launch_us"],2), round(e["frac"],3)) for k,e in t.items() if "gemm_large" in k or "prefill_att" in k})
PY
done
} > $O 2>&1
cat $O | grep -v amdgpu
