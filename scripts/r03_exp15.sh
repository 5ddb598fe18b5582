#!/bin/bash
# in-place operands, whole-line form at P = 2 (8 rows x 128 bytes per copy): tests, micro, chunk-512 and plain jobs
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp15.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -n 3
for rm in 0 1; do
  echo "== prepacked weights, MI355X_F8_ROWMAJOR=$rm"
  MI355X_F8_ROWMAJOR=$rm timeout -k 10 300 python scripts/bench_scaled_mm.py --prepacked 576 1024 4096 8192 2>&1 | grep "total"
done
echo "== weights in place, MI355X_F8_ROWMAJOR=3"
MI355X_F8_ROWMAJOR=3 timeout -k 10 300 python scripts/bench_scaled_mm.py 576 1024 4096 8192 2>&1 | grep "total"
for rm in 0 1; do
  for args in "--quant fp8 --chunk-tokens 512" "--quant fp8" "--quant int8 --chunk-tokens 512"; do
  MI355X_F8_ROWMAJOR=$rm timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
t={e["kernel"]:e for e in [d["roofline"]]+d["roofline_other"]}
print("rowmajor=$rm $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"], {k:round(e["avg_launch_us"],2) for k,e in t.items() if "gemm_large" in k or k in ("rms_norm_dynamic_per_token_quant","dynamic_per_token_scaled_fp8_quant")})
PY
  done
done
} > $O 2>&1
tail -n 30 $O
