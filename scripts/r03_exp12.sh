#!/bin/bash
# 8-bit prefill GEMM reading its operands in place (MI355X_F8_ROWMAJOR bits: 1 activations, 2 weights) against the
# operand-image form (0): tests, per-projection times, job records.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp12.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_fp8_gemm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -n 4
for rm in 0 1 3; do
  echo "== MI355X_F8_ROWMAJOR=$rm"
  MI355X_F8_ROWMAJOR=$rm timeout -k 10 300 python scripts/bench_scaled_mm.py 512 2048 8192 2>&1 | grep -v "^int8 M=.*:  " 
done
for rm in 0 3; do
  for args in "--quant fp8" "--model llama-3-70b --tp-rank-of 8" "--quant fp8 --chunk-tokens 512"; do
    MI355X_F8_ROWMAJOR=$rm timeout -k 10 400 python bench.py $args --skip-cpu 2> gpurun_out/r03c.err | tail -n 1 > gpurun_out/r03c_tmp.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r03c_tmp.json"))
print("rowmajor=$rm bench $args:", d["value"], d["ms_per_step"], d["ttft_p50_ms"])
PY
  done
done
} > $O 2>&1
tail -n 40 $O
